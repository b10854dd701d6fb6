#!/bin/bash
# round 5, first GPU trip: the new list kernels against numpy / the torch builder, the paths that use them, and the three CRISPR workloads
export TMPDIR=/tmp
O=gpurun_out/r05a
mkdir -p $O
python -m pytest tests/test_gpu_round5.py -x -q > $O/t5.log 2>&1; echo "t5 rc=$?" >> $O/t5.log
python -m pytest tests/test_gpu_round4.py -x -q -k "sparse or single1 or single4 or config3" > $O/t4.log 2>&1; echo "t4 rc=$?" >> $O/t4.log
B="python3 bench.py --cpu-seconds 0 --e2e 0 --no-extras"
for w in de_c4 de_c4_single1 de_c4_single4; do
	$B --workload $w --steps 10 --warmup 3 > $O/bench_$w.json 2> $O/bench_$w.err
done
tail -3 $O/t5.log $O/t4.log
for w in de_c4 de_c4_single1 de_c4_single4; do tail -c 1500 $O/bench_$w.json; echo; tail -3 $O/bench_$w.err; done
