#!/bin/bash
# round 6: NormvarPlan -- its test, then the two bench workloads it serves
mkdir -p gpurun_out/r06n
python -m pytest tests -x -q -m gpu -k "normvar" > gpurun_out/r06n/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06n/tests.log; tail -15 gpurun_out/r06n/tests.log
for w in normvar_c2 chain_c2; do
	python3 bench.py --workload $w --steps 20 --warmup 3 --cpu-seconds 0 --e2e 0 --no-extras > gpurun_out/r06n/$w.json 2> gpurun_out/r06n/$w.err; echo "$w rc=$?"; tail -c 1200 gpurun_out/r06n/$w.json; echo; tail -3 gpurun_out/r06n/$w.err
done
