#!/bin/bash
# round 6, fifth GPU call: the whole GPU suite in the driver's form (-x); normvar after the four-cells-per-thread rewrite (kernel statistics + step)
export TMPDIR=/tmp
O=gpurun_out/r06f
mkdir -p $O
python -m pytest tests -x -q -m gpu --durations=10 > $O/gputests_x.log 2>&1; echo "rc=$?" >> $O/gputests_x.log
tail -n 20 $O/gputests_x.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/nv -o nv -- python3 bench.py --workload normvar_c2 --steps 10 --warmup 2 --no-extras --cpu-seconds 0 --e2e 0 > $O/nv.json 2> $O/nv.err
f=$(find $O/nv -name "*kernel_stats.csv" | head -1); cp "$f" $O/r06_normvar_c2_kernel_stats.csv; rm -rf $O/nv
python3 tools/kstats.py $O/r06_normvar_c2_kernel_stats.csv k_nv
python bench.py --workload normvar_c2 --steps 20 --warmup 3 --no-extras --cpu-seconds 0 --e2e 0 2>&1 | grep "^{\"metric" | cut -c1-400
python bench.py --workload chain_c2 --steps 10 --warmup 3 --no-extras --cpu-seconds 0 --e2e 0 2>&1 | grep "^{\"metric" | cut -c1-400
