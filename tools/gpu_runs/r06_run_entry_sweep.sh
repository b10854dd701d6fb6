#!/bin/bash
# round 6: the seeded random shapes of tests/test_gpu_random_shapes.py through the LIBRARY'S WHOLE-PROBLEM ENTRIES (NRM_HOST_ENTRY=1: association_tests single=0/1/4,
# normvar route to nrm_association_tests_host / _single1_host / _single4_host / nrm_normvar_host; what an entry answers NRM_E_UNSUPPORTED goes on to the package's paths)
mkdir -p gpurun_out/r06e
NRM_HOST_ENTRY=1 NRM_TEST_SEEDS=${1:-150} python -m pytest tests/test_gpu_random_shapes.py -q -m gpu -p no:cacheprovider > gpurun_out/r06e/entry_sweep.log 2>&1
echo "rc=$?" >> gpurun_out/r06e/entry_sweep.log
tail -40 gpurun_out/r06e/entry_sweep.log | cut -c1-300
