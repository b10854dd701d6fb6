#!/bin/bash
# Round-4 profile of the sparse-design path (csrc/nrm_de_sparse.hip) on BASELINE configs[3]: norm.de and norm.de(single=4), per-kernel
# stats and HBM-side counters (separate passes), the kernel alone with parts of its work taken away, the dense path beside it.
export TMPDIR=/tmp
O=gpurun_out/r04prof
mkdir -p $O
B="python3 bench.py --cpu-seconds 0 --e2e 0 --no-extras"
for w in de_c4 de_c4_single4; do
	rocprofv3 --kernel-trace --stats --output-format csv -d $O/${w}_sp_stats -o $w -- $B --workload $w --steps 5 --warmup 2 > $O/${w}_sp_stats.json 2> $O/${w}_sp_stats.err
	f=$(find $O/${w}_sp_stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/r04_${w}_sparse_kernel_stats.csv
	for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
		rocprofv3 --pmc $c --output-format csv -d $O/${w}_sp_$c -o pmc -- $B --workload $w --steps 3 --warmup 1 > /dev/null 2> $O/${w}_sp_$c.err
	done
	python3 tools/pmc_summary.py $O/${w}_sp_FETCH_SIZE $O/${w}_sp_WRITE_SIZE $O/${w}_sp_GRBM_GUI_ACTIVE > $O/r04_pmc_${w}_sparse.json
done
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/de_c4_sp_SQ -o pmc -- $B --workload de_c4 --steps 3 --warmup 1 > /dev/null 2> $O/de_c4_sp_SQ.err
python3 tools/pmc_summary.py $O/de_c4_sp_SQ > $O/r04_pmc_de_c4_sparse_sq.json
python3 tools/time_de_sparse_parts.py > $O/r04_de_sparse_parts.txt 2>&1
python3 tools/time_de_sparse.py > $O/r04_de_sparse_vs_dense.txt 2>&1
ls $O/r04_*sparse*
