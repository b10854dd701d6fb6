#!/bin/bash
# round 6: the default bench line (as the driver runs it), the single=1 step on both statistics routes under four BLAS thread settings, then the profiles of every workload
export TMPDIR=/tmp
O=gpurun_out/r06i
mkdir -p $O
python bench.py > $O/r06_bench_default.json 2> $O/bench_default.err; tail -n 1 $O/r06_bench_default.json | cut -c1-1500
for th in unset 1 8 256; do
	if [ $th = unset ]; then python tools/time_single1_routes.py 10 >> $O/single1_routes.txt 2>&1
	else OPENBLAS_NUM_THREADS=$th OMP_NUM_THREADS=$th MKL_NUM_THREADS=$th python tools/time_single1_routes.py 10 >> $O/single1_routes.txt 2>&1; fi
done
grep -v "amdgpu.ids" $O/single1_routes.txt
bash tools/profile_r06.sh > $O/profile.log 2>&1
cp gpurun_out/r06prof/r06_* $O/ 2>/dev/null
ls $O
