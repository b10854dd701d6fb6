#!/bin/bash
# Round-6 profiles (run on the GPU box from the repo root): per-kernel stats and HBM-side counters of every bench workload, after the host left the
# resident single=1 / single=4 steps (HIP graphs), normvar's passes went to four cells per thread, and K2's edge tiles to four waves.
# Counters are collected in their own passes (no trace options besides --kernel-trace), as gpurun requires; the program itself follows `--`.
export TMPDIR=/tmp
O=gpurun_out/r06prof
mkdir -p $O
B="python3 bench.py --cpu-seconds 0 --e2e 0 --no-extras"
declare -A W=( [c2]="--steps 20 --warmup 3" [de_c3]="--workload de_c3 --steps 20 --warmup 3" [de_c4]="--workload de_c4 --steps 10 --warmup 2" [coex_c5]="--workload coex_c5 --steps 3 --warmup 1" \
	[de_c4_single4]="--workload de_c4_single4 --steps 10 --warmup 2" [de_c4_single1]="--workload de_c4_single1 --steps 10 --warmup 2" [binnet_c5]="--workload binnet_c5 --steps 5 --warmup 2" \
	[c2_f64]="--workload coex_c2_f64 --steps 10 --warmup 3" [c5_f64]="--workload coex_c5_f64 --steps 3 --warmup 1" [normvar_c2]="--workload normvar_c2 --steps 10 --warmup 2" [chain_c2]="--workload chain_c2 --steps 10 --warmup 2" )
ALL=${1:-"c2 de_c3 de_c4 coex_c5 de_c4_single4 de_c4_single1 binnet_c5 c2_f64 c5_f64 normvar_c2 chain_c2"}
PMC=${2:-"c2 de_c4 de_c4_single4 de_c4_single1 coex_c5 de_c3 binnet_c5 normvar_c2"}
for w in $ALL; do
	rocprofv3 --kernel-trace --stats --output-format csv -d $O/${w}_stats -o $w -- $B ${W[$w]} > $O/${w}_stats.json 2> $O/${w}_stats.err
	f=$(find $O/${w}_stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/r06_${w}_kernel_stats.csv
done
for w in $PMC; do
	for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
		rocprofv3 --pmc $c --output-format csv -d $O/${w}_$c -o pmc -- $B ${W[$w]} --steps 3 --warmup 1 > /dev/null 2> $O/${w}_$c.err
	done
	python3 tools/pmc_summary.py $O/${w}_FETCH_SIZE $O/${w}_WRITE_SIZE $O/${w}_GRBM_GUI_ACTIVE > $O/r06_pmc_$w.json
	rm -rf $O/${w}_FETCH_SIZE $O/${w}_WRITE_SIZE $O/${w}_GRBM_GUI_ACTIVE
done
rm -rf $O/*_stats
ls -la $O
