#!/bin/bash
# round 6, end: the seeded random shapes far beyond the earlier sweeps (2000 extra seeds per test: the first 750 are the cases of profiles/r06_random_shapes.txt),
# through the package's device paths, on the final library
mkdir -p gpurun_out/r06w
NRM_TEST_SEEDS=${1:-2000} timeout 1500 python -m pytest tests/test_gpu_random_shapes.py -q -m gpu -p no:cacheprovider -x > gpurun_out/r06w/sweep.log 2>&1
echo "rc=$?" >> gpurun_out/r06w/sweep.log
tail -12 gpurun_out/r06w/sweep.log | cut -c1-300
