#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06r
mkdir -p $O
for rep in 1 2 3; do
	for v in 0 1; do
		echo "S1_PREFETCH=$v (rep $rep): $(python tools/with_lib.py tools/exp/nrm_single1_S1_PREFETCH_$v.so bench.py --workload de_c4_single1 --steps 50 --warmup 5 --no-extras --cpu-seconds 0 --e2e 0 2>&1 | grep '^{"workload_detail' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],4), d['kernels_ms'])")" >> $O/prefetch.txt
	done
done
cat $O/prefetch.txt
python tools/with_lib.py tools/exp/nrm_single1_S1_PREFETCH_1.so -m pytest tests -q -x -m gpu -k "single1" > $O/t.log 2>&1; tail -n 3 $O/t.log
