#!/bin/bash
# round 6, fourth GPU call: the whole GPU suite in the driver's form (-x), per-step kernel timelines of the resident single=1 / single=4 steps (HIP graphs),
# normvar's kernel statistics, single=1 under three BLAS thread settings twice over
export TMPDIR=/tmp
O=gpurun_out/r06e
mkdir -p $O
python -m pytest tests -x -q -m gpu --durations=15 > $O/gputests_x.log 2>&1; echo "rc=$?" >> $O/gputests_x.log
tail -n 30 $O/gputests_x.log
for w in de_c4_single1 de_c4_single4; do
	rocprofv3 --kernel-trace --output-format csv -d $O/tl_$w -o tl -- python3 bench.py --workload $w --steps 6 --warmup 1 --no-extras --cpu-seconds 0 --e2e 0 > /dev/null 2> $O/tl_$w.err
	python3 tools/step_gaps.py $O/tl_$w $([ $w = de_c4_single1 ] && echo k_s1_cells || echo k_s4_sweep) > $O/timeline_$w.txt 2>&1
	rm -rf $O/tl_$w
	tail -n 30 $O/timeline_$w.txt
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/nv -o nv -- python3 bench.py --workload normvar_c2 --steps 10 --warmup 2 --no-extras --cpu-seconds 0 --e2e 0 > $O/nv.json 2> $O/nv.err
f=$(find $O/nv -name "*kernel_stats.csv" | head -1); cp "$f" $O/r06_normvar_c2_kernel_stats.csv; rm -rf $O/nv
python3 tools/kstats.py $O/r06_normvar_c2_kernel_stats.csv k_nv
for rep in 1 2; do
	for th in unset 1 256; do
		echo "== de_c4_single1 OPENBLAS_NUM_THREADS=$th (rep $rep)" >> $O/steps.txt
		if [ $th = unset ]; then
			python bench.py --workload de_c4_single1 --steps 50 --warmup 5 --no-extras --cpu-seconds 0 --e2e 0 2>&1 | grep "^{\"metric" | cut -c1-260 >> $O/steps.txt
		else
			OPENBLAS_NUM_THREADS=$th OMP_NUM_THREADS=$th python bench.py --workload de_c4_single1 --steps 50 --warmup 5 --no-extras --cpu-seconds 0 --e2e 0 2>&1 | grep "^{\"metric" | cut -c1-260 >> $O/steps.txt
		fi
	done
done
cat $O/steps.txt
