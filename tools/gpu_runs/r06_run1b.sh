#!/bin/bash
# the whole GPU suite in the driver's form (-x) on a fresh box, then without -x for whatever is left
export TMPDIR=/tmp
O=gpurun_out/r06c
mkdir -p $O
python -m pytest tests -x -q -m gpu --durations=30 > $O/gputests_x.log 2>&1; echo "rc=$?" >> $O/gputests_x.log
tail -n 60 $O/gputests_x.log
