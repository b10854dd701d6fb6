#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06q
mkdir -p $O
python -m pytest tests -x -q -m gpu --durations=5 > $O/gputests_x.log 2>&1; echo "rc=$?" >> $O/gputests_x.log
tail -n 10 $O/gputests_x.log
python tools/s1_ye_offset_exp.py 2>&1 | grep -v amdgpu.ids > $O/ye_offsets.txt; cat $O/ye_offsets.txt
