#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06p
mkdir -p $O
python -m pytest tests -x -q -m gpu --durations=8 > $O/gputests_x.log 2>&1; echo "rc=$?" >> $O/gputests_x.log
tail -n 14 $O/gputests_x.log
for rep in 1 2; do python bench.py --workload de_c4_single1 --steps 50 --warmup 5 --no-extras --cpu-seconds 0 --e2e 0 2>&1 | grep "^{\"metric" | cut -c1-200; done
