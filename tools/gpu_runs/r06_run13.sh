#!/bin/bash
# round 6: the seeded random-shape sweeps of round 5 again, under the 1e-6 bound for fp32 inputs (round-5 verdict item 4)
export TMPDIR=/tmp
O=gpurun_out/r06n
mkdir -p $O
NRM_TEST_SEEDS=1000 python -m pytest tests/test_gpu_random_shapes.py -q -k "not dense" -p no:cacheprovider > $O/sweep_a.log 2>&1; echo "rc=$?" >> $O/sweep_a.log; tail -n 15 $O/sweep_a.log | cut -c1-300
NRM_TEST_SEEDS=1500 python -m pytest tests/test_gpu_random_shapes.py -q -k "dense" -p no:cacheprovider > $O/sweep_b.log 2>&1; echo "rc=$?" >> $O/sweep_b.log; tail -n 15 $O/sweep_b.log | cut -c1-300
