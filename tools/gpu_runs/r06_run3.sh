#!/bin/bash
# the GPU suite from the round-6 tests on (what -x cut off), then everything of tools/gpu_runs/r06_run2.sh but the tests it repeats
export TMPDIR=/tmp
O=gpurun_out/r06d
mkdir -p $O
python -m pytest tests/test_gpu_round6.py -q > $O/t_round6.log 2>&1; echo "rc=$?" >> $O/t_round6.log; tail -n 25 $O/t_round6.log
python -m pytest tests -q -m gpu -k "sharded or bench_ or two_ranks or rccl_ or cli_ or zz_perf" -s > $O/t_tier3.log 2>&1; echo "rc=$?" >> $O/t_tier3.log; tail -n 60 $O/t_tier3.log
bash tools/gpu_runs/r06_run2.sh
