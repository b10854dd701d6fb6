#!/bin/bash
# round 6: binnet's search wave by wave and the next row requested ahead -- parity (new test + the binnet tests), then A/B timing on one box.
# (The kernel variants this timed are not in the tree: profiles/r06_binnet_variants.txt, DESIGN section 9 item 7.  The NRM_DEBUG switches below belonged to them.)
mkdir -p gpurun_out/r06b
python -m pytest tests -x -q -m gpu -k "binnet" > gpurun_out/r06b/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06b/tests.log; tail -5 gpurun_out/r06b/tests.log
for rep in 1 2; do
	echo "== default"; python3 tools/time_binnet.py 2>&1 | grep -v amdgpu.ids
	echo "== no row ahead (NRM_DEBUG=binnet_pipe=0)"; NRM_DEBUG=binnet_pipe=0 python3 tools/time_binnet.py 2>&1 | grep -v amdgpu.ids
	echo "== block-wide search (NRM_DEBUG=binnet_list=0)"; NRM_DEBUG=binnet_list=0 python3 tools/time_binnet.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r06b/time.txt 2>&1
cat gpurun_out/r06b/time.txt
