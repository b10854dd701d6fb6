#!/bin/bash
# round 6, second GPU call: the new tests (resident single=1 / single=4 plans, device selection, exp probe, per-gene dimreduce), the random-shape suite
# under the 1e-6 bar, the bench tests; then the single=1 / single=4 resident steps under three BLAS thread settings and their kernel timelines
export TMPDIR=/tmp
O=gpurun_out/r06d
mkdir -p $O
for w in de_c4_single1 de_c4_single4 normvar_c2; do
	for th in unset 1 256; do
		echo "== $w OPENBLAS_NUM_THREADS=$th" >> $O/steps.txt
		if [ $th = unset ]; then
			python bench.py --workload $w --steps 20 --warmup 3 --no-extras --cpu-seconds 0 --e2e 0 2>&1 | grep -v "^{\"workload_detail" | cut -c1-700 >> $O/steps.txt
		else
			OPENBLAS_NUM_THREADS=$th OMP_NUM_THREADS=$th python bench.py --workload $w --steps 20 --warmup 3 --no-extras --cpu-seconds 0 --e2e 0 2>&1 | grep -v "^{\"workload_detail" | cut -c1-700 >> $O/steps.txt
		fi
	done
done
cat $O/steps.txt
for w in de_c4_single1 de_c4_single4; do
	rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tl_$w -o tl -- python3 bench.py --workload $w --steps 6 --warmup 1 --no-extras --cpu-seconds 0 --e2e 0 > /dev/null 2> $O/tl_$w.err
	python3 tools/step_gaps.py $O/tl_$w $([ $w = de_c4_single1 ] && echo k_s1_cells || echo k_s4_sweep) > $O/timeline_$w.txt 2>&1
	rm -rf $O/tl_$w
	tail -n 30 $O/timeline_$w.txt
done
python tools/k2_edge_exp.py > $O/k2_edge.txt 2>&1; cat $O/k2_edge.txt
