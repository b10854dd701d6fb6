#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06o
mkdir -p $O
NRM_TEST_SEEDS=1000 python -m pytest tests/test_gpu_random_shapes.py -q -k "not dense" -p no:cacheprovider > $O/sweep_a.log 2>&1; echo "rc=$?" >> $O/sweep_a.log; tail -n 6 $O/sweep_a.log | cut -c1-300
python -m pytest tests/test_gpu_round6.py -q -k "normvar_host" > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log; tail -n 5 $O/t.log
