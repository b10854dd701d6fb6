#!/bin/bash
# round 6, final validation: the whole GPU suite in the driver's form, smoke(), the default bench line, kernel statistics of the two plans as committed
export TMPDIR=/tmp
O=gpurun_out/r06k
mkdir -p $O
python -m pytest tests -x -q -m gpu --durations=10 > $O/gputests_x.log 2>&1; echo "rc=$?" >> $O/gputests_x.log
tail -n 6 $O/gputests_x.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 2 $O/smoke.log
python bench.py > $O/r06_bench_default.json 2> $O/bench_default.err; tail -n 1 $O/r06_bench_default.json | cut -c1-400
for w in de_c4_single1 de_c4_single4; do
	rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$w -o $w -- python3 bench.py --workload $w --steps 10 --warmup 2 --no-extras --cpu-seconds 0 --e2e 0 > /dev/null 2> $O/st_$w.err
	f=$(find $O/st_$w -name "*kernel_stats.csv" | head -1); cp "$f" $O/r06_${w}_kernel_stats.csv; rm -rf $O/st_$w
done
ls $O
