#!/bin/bash
# the GPU suite in the driver's form (fresh box), log kept under gpurun_out/r06s/
mkdir -p gpurun_out/r06s
python -m pytest tests/ -x -q -m gpu --durations=15 > gpurun_out/r06s/gputests_x.log 2>&1
echo "rc=$?" >> gpurun_out/r06s/gputests_x.log
tail -25 gpurun_out/r06s/gputests_x.log
