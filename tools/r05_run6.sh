#!/bin/bash
# SQ counters of the sparse-design kernel: the default build against the pipelined-gather build
export TMPDIR=/tmp
O=gpurun_out/r05sq; mkdir -p $O
for v in base pipe4; do
	rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/${v}_a -o pmc -- python3 tools/with_lib.py tools/exp/nrm_de_sparse_$v.so tools/time_de_sparse.py > /dev/null 2> $O/${v}_a.err
	rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $O/${v}_b -o pmc -- python3 tools/with_lib.py tools/exp/nrm_de_sparse_$v.so tools/time_de_sparse.py > /dev/null 2> $O/${v}_b.err
	python3 tools/pmc_summary.py $O/${v}_a $O/${v}_b > $O/sq_$v.json
	rm -rf $O/${v}_a $O/${v}_b
done
python3 - <<'PY'
import json
for v in ('base','pipe4'):
    d=json.load(open('gpurun_out/r05sq/sq_%s.json'%v))
    for k,e in d.items():
        if 'k_de_sparse' in k:
            print(v,k)
            for c,x in sorted(e.items()): print('   %-24s %.4g'%(c,x))
PY
