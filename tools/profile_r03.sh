#!/bin/bash
# Round-3 profiles (run on the GPU box from the repo root): per-kernel stats and HBM-side counters of ALL bench workloads.
# Counters are collected in their own passes (no trace options besides --kernel-trace), as gpurun requires.
export TMPDIR=/tmp
O=gpurun_out/r03prof
mkdir -p $O
B="python3 bench.py --cpu-seconds 0 --e2e 0 --no-extras"
declare -A W=( [c2]="--steps 20 --warmup 3" [de_c3]="--workload de_c3 --steps 20 --warmup 3" [de_c4]="--workload de_c4 --steps 5 --warmup 2" [coex_c5]="--workload coex_c5 --steps 3 --warmup 1" )
for w in c2 de_c3 de_c4 coex_c5; do
	rocprofv3 --kernel-trace --stats --output-format csv -d $O/${w}_stats -o $w -- $B ${W[$w]} > $O/${w}_stats.json 2> $O/${w}_stats.err
	for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
		rocprofv3 --pmc $c --output-format csv -d $O/${w}_$c -o pmc -- $B ${W[$w]} --steps 3 --warmup 1 > /dev/null 2> $O/${w}_$c.err
	done
	python3 tools/pmc_summary.py $O/${w}_FETCH_SIZE $O/${w}_WRITE_SIZE $O/${w}_GRBM_GUI_ACTIVE > $O/r03_pmc_$w.json
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/c2_SQ -o pmc -- $B --steps 3 --warmup 1 > /dev/null 2> $O/c2_SQ.err
python3 tools/pmc_summary.py $O/c2_FETCH_SIZE $O/c2_WRITE_SIZE $O/c2_GRBM_GUI_ACTIVE $O/c2_SQ > $O/r03_pmc_c2.json
for w in c2 de_c3 de_c4 coex_c5; do f=$(find $O/${w}_stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/r03_${w}_kernel_stats.csv; done
ls $O/*.json $O/*.csv
