#!/bin/bash
# flakiness check: the GPU suite in the driver's form twice on one fresh box
mkdir -p gpurun_out/r06s2
for i in 1 2; do
	python -m pytest tests/ -x -q -m gpu > gpurun_out/r06s2/run$i.log 2>&1
	echo "rc=$?" >> gpurun_out/r06s2/run$i.log
	tail -4 gpurun_out/r06s2/run$i.log
done
