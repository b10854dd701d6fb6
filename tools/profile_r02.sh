#!/bin/bash
# Round-2 profiles (run on the GPU box from the repo root): per-kernel stats and HBM-side counters of the bench workloads.
# Counters are collected in their own passes (no trace options besides --kernel-trace), as gpurun requires.
export TMPDIR=/tmp
O=gpurun_out/r02prof
mkdir -p $O
B="python3 bench.py --cpu-seconds 0 --e2e 0 --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2_stats -o c2 -- $B --steps 20 --warmup 3 > $O/c2_stats.json 2> $O/c2_stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_stats -o c3 -- $B --workload de_c3 --steps 20 --warmup 3 > $O/c3_stats.json 2> $O/c3_stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_stats -o c4 -- $B --workload de_c4 --steps 5 --warmup 2 > $O/c4_stats.json 2> $O/c4_stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -o c5 -- $B --workload coex_c5 --steps 3 --warmup 1 > $O/c5_stats.json 2> $O/c5_stats.err
for c in FETCH_SIZE WRITE_SIZE; do
	rocprofv3 --pmc $c --output-format csv -d $O/c2_$c -o pmc -- $B --steps 3 --warmup 1 > /dev/null 2> $O/c2_$c.err
	rocprofv3 --pmc $c --output-format csv -d $O/c3_$c -o pmc -- $B --workload de_c3 --steps 3 --warmup 1 > /dev/null 2> $O/c3_$c.err
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/c2_SQ -o pmc -- $B --steps 3 --warmup 1 > /dev/null 2> $O/c2_SQ.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/c3_SQ -o pmc -- $B --workload de_c3 --steps 3 --warmup 1 > /dev/null 2> $O/c3_SQ.err
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/c2_CLK -o pmc -- $B --steps 10 --warmup 2 > /dev/null 2> $O/c2_CLK.err
python3 tools/pmc_summary.py $O/c2_FETCH_SIZE $O/c2_WRITE_SIZE $O/c2_SQ $O/c2_CLK > $O/r02_pmc_c2.json
python3 tools/pmc_summary.py $O/c3_FETCH_SIZE $O/c3_WRITE_SIZE $O/c3_SQ > $O/r02_pmc_de_c3.json
find $O -name "*kernel_stats.csv" | head; head -c 1500 $O/r02_pmc_c2.json
