export TMPDIR=/tmp
O=gpurun_out/r02prof
mkdir -p $O
B="python3 bench.py --cpu-seconds 0 --e2e 0 --no-extras"
rm -rf $O/c3_stats $O/c3_FETCH_SIZE $O/c3_WRITE_SIZE $O/c3_SQ
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_stats -o c3 -- $B --workload de_c3 --steps 20 --warmup 3 > $O/c3_stats.json 2> $O/c3_stats.err
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --output-format csv -d $O/c3_$c -o pmc -- $B --workload de_c3 --steps 3 --warmup 1 > /dev/null 2> $O/c3_$c.err; done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/c3_SQ -o pmc -- $B --workload de_c3 --steps 3 --warmup 1 > /dev/null 2> $O/c3_SQ.err
python3 tools/pmc_summary.py $O/c3_FETCH_SIZE $O/c3_WRITE_SIZE $O/c3_SQ > $O/r02_pmc_de_c3.json
