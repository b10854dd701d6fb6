#!/bin/bash
# round 6: K2's ragged edge as a strip kernel -- exactness and the Gram engines' parity tests, then A/B timing of K2 on configs[1] (NRM_DEBUG=k2_strip=0: one launch)
mkdir -p gpurun_out/r06t
python -m pytest tests -x -q -m gpu -k "strip or integer_gram or gram_engines or g14 or c1_de_coex or full_size_c2 or pipelined_coex or banded" > gpurun_out/r06t/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06t/tests.log; tail -15 gpurun_out/r06t/tests.log
for rep in 1 2 3; do
	echo "== strip"; python3 tools/k2i8_time.py - 5000 10000 6 2>&1 | grep -v amdgpu.ids
	echo "== one launch (NRM_DEBUG=k2_strip=0)"; NRM_DEBUG=k2_strip=0 python3 tools/k2i8_time.py - 5000 10000 6 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r06t/time.txt 2>&1
cat gpurun_out/r06t/time.txt
