#!/bin/bash
# round 6, end: the default bench line and the profiles of the workloads whose kernels changed after the first profile run (normvar: two genes per workgroup)
mkdir -p gpurun_out/r06f
python3 bench.py > gpurun_out/r06f/bench_default.json 2> gpurun_out/r06f/bench_default.err
echo "bench rc=$?"
bash tools/profile_r06.sh "c2 normvar_c2 chain_c2" "c2 normvar_c2" > gpurun_out/r06f/profile.log 2>&1
tail -c 1500 gpurun_out/r06f/bench_default.json
