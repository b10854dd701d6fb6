#!/bin/bash
# Round-4 profile of single=1 at BASELINE configs[3] size after its rewrite (k_s1_stream + k_s1_cells): per-kernel stats and HBM-side
# counters (separate passes, no trace options besides --kernel-trace), the stream kernel alone with parts of its work taken away.
export TMPDIR=/tmp
O=gpurun_out/r04prof
mkdir -p $O
B="python3 bench.py --cpu-seconds 0 --e2e 0 --no-extras --workload de_c4_single1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s1_stats -o s1 -- $B --steps 5 --warmup 2 > $O/s1_stats.json 2> $O/s1_stats.err
f=$(find $O/s1_stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/r04_de_c4_single1_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
	rocprofv3 --pmc $c --output-format csv -d $O/s1_$c -o pmc -- $B --steps 3 --warmup 1 > /dev/null 2> $O/s1_$c.err
done
python3 tools/pmc_summary.py $O/s1_FETCH_SIZE $O/s1_WRITE_SIZE $O/s1_GRBM_GUI_ACTIVE > $O/r04_pmc_de_c4_single1.json
python3 tools/time_single1_stream.py > $O/r04_single1_stream_ablation.txt 2>&1
NRM_S1_TRACE=1 python3 tools/time_single1_host.py 2>&1 | grep -E "phases|per step" | tail -2 > $O/r04_single1_phases.txt
ls $O/r04_*single1*
