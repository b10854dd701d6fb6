#!/bin/bash
# round 6: the resident single=1 / single=4 steps with their design-side work on a second stream (fork / join inside the captured graph)
export TMPDIR=/tmp
O=gpurun_out/r06j
mkdir -p $O
python -m pytest tests/test_gpu_round6.py -q -x -k "plan or raises" > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log; tail -n 6 $O/t.log
for w in de_c4_single1 de_c4_single4; do
	for rep in 1 2; do
		python bench.py --workload $w --steps 50 --warmup 5 --no-extras --cpu-seconds 0 --e2e 0 2>&1 | grep "^{\"metric" | cut -c1-240 >> $O/steps.txt
	done
	rocprofv3 --kernel-trace --output-format csv -d $O/tl_$w -o tl -- python3 bench.py --workload $w --steps 6 --warmup 1 --no-extras --cpu-seconds 0 --e2e 0 > /dev/null 2> $O/tl_$w.err
	python3 tools/step_gaps.py $O/tl_$w $([ $w = de_c4_single1 ] && echo k_s1_cells || echo k_s4_sweep) > $O/timeline_$w.txt 2>&1
	rm -rf $O/tl_$w
	head -n 14 $O/timeline_$w.txt
done
cat $O/steps.txt
