#!/bin/bash
# round 5, second GPU trip: full GPU tests of the touched paths, kernel traces of the three CRISPR workloads (no at::native / rocprim inside a step?)
export TMPDIR=/tmp
O=gpurun_out/r05b
mkdir -p $O
python -m pytest tests/test_gpu_round5.py -x -q > $O/t5.log 2>&1; echo "t5 rc=$?" >> $O/t5.log
python -m pytest tests/test_gpu_round4.py tests/test_gpu_round2.py -x -q -k "single4 or single1 or sparse or config3" > $O/t4.log 2>&1; echo "t4 rc=$?" >> $O/t4.log
B="python3 bench.py --cpu-seconds 0 --e2e 0 --no-extras"
for w in de_c4 de_c4_single1 de_c4_single4; do
	rocprofv3 --kernel-trace --stats --output-format csv -d $O/${w}_stats -o $w -- $B --workload $w --steps 10 --warmup 3 > $O/${w}.json 2> $O/${w}.err
	f=$(find $O/${w}_stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/r05_${w}_kernel_stats.csv
	rm -rf $O/${w}_stats
done
tail -n 3 $O/t5.log $O/t4.log
for w in de_c4 de_c4_single1 de_c4_single4; do tail -c 1200 $O/$w.json; echo; done
