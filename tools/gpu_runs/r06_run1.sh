#!/bin/bash
# round 6, first GPU call: the whole GPU suite WITHOUT -x on a fresh box (what the driver's box would have shown behind the failing test),
# then the single=1 resident step under three BLAS thread settings (root cause of the 18.4 ms the driver's box measured)
export TMPDIR=/tmp
O=gpurun_out/r06a
mkdir -p $O
nproc > $O/host.txt; python -c "import numpy; numpy.show_config()" >> $O/host.txt 2>&1
python -m pytest tests -q -m gpu --durations=25 > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log
tail -n 40 $O/gputests.log
for th in unset 1 256; do
	echo "== OPENBLAS_NUM_THREADS=$th" >> $O/single1_threads.txt
	if [ $th = unset ]; then
		NRM_DEBUG=s1_trace=1 python bench.py --workload de_c4_single1 --steps 10 --warmup 2 --no-extras --cpu-seconds 0 --e2e 0 2>&1 | grep -v "^{\"workload_detail" | cut -c1-600 >> $O/single1_threads.txt
	else
		OPENBLAS_NUM_THREADS=$th OMP_NUM_THREADS=$th NRM_DEBUG=s1_trace=1 python bench.py --workload de_c4_single1 --steps 10 --warmup 2 --no-extras --cpu-seconds 0 --e2e 0 2>&1 | grep -v "^{\"workload_detail" | cut -c1-600 >> $O/single1_threads.txt
	fi
done
for th in unset 1 256; do
	echo "== single4 OPENBLAS_NUM_THREADS=$th" >> $O/single1_threads.txt
	if [ $th = unset ]; then
		python bench.py --workload de_c4_single4 --steps 10 --warmup 2 --no-extras --cpu-seconds 0 --e2e 0 2>&1 | grep -v "^{\"workload_detail" | cut -c1-400 >> $O/single1_threads.txt
	else
		OPENBLAS_NUM_THREADS=$th OMP_NUM_THREADS=$th python bench.py --workload de_c4_single4 --steps 10 --warmup 2 --no-extras --cpu-seconds 0 --e2e 0 2>&1 | grep -v "^{\"workload_detail" | cut -c1-400 >> $O/single1_threads.txt
	fi
done
cat $O/single1_threads.txt | tail -60
