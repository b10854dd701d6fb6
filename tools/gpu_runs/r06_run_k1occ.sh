#!/bin/bash
# round 6: K1's register allocation held to 3 / 4 waves per SIMD (launch bounds) against the default 2 -- configs[1] rows, configs[3] genes, the configs[4] slice
mkdir -p gpurun_out/r06k
for rep in 1 2; do
for v in 1 3 4; do
	echo "== K1_MINW=$v"
	python3 tools/with_lib.py tools/exp/nrm_residualize_K1_MINW_$v.so tools/k1_time.py 5000 10000 f32 3 2>&1 | grep -v amdgpu.ids
	python3 tools/with_lib.py tools/exp/nrm_residualize_K1_MINW_$v.so tools/k1_time.py 15000 50000 f32 5 2>&1 | grep -v amdgpu.ids
	python3 tools/with_lib.py tools/exp/nrm_residualize_K1_MINW_$v.so tools/k1_time.py 3840 500000 f64 3 2>&1 | grep -v amdgpu.ids
done
done > gpurun_out/r06k/k1_occupancy.txt 2>&1
cat gpurun_out/r06k/k1_occupancy.txt
