#!/bin/bash
# round 6: the torch-free single=4 slow path (nrm_gram_host / nrm_pvalues_host), then the whole GPU suite in the driver's form
mkdir -p gpurun_out/r06h
python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py -x -q -m gpu -k "without_a_closed_form or de_methods_and_binnet" > gpurun_out/r06h/new.log 2>&1
echo "rc=$?" >> gpurun_out/r06h/new.log
tail -30 gpurun_out/r06h/new.log
python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/r06h/gputests_x.log 2>&1
echo "rc=$?" >> gpurun_out/r06h/gputests_x.log
tail -25 gpurun_out/r06h/gputests_x.log
