#!/bin/bash
# round 6: K2's edge-tile wave map A/B on ONE box (tools/exp/nrm_gram_i8_QI_EDGE_{0,1}.so), alternating
export TMPDIR=/tmp
O=gpurun_out/r06h
mkdir -p $O
for rep in 1 2 3; do
	for v in 0 1; do
		echo "QI_EDGE=$v rep $rep: $(python tools/k2i8_time.py tools/exp/nrm_gram_i8_QI_EDGE_$v.so 5000 10000 6 2>&1 | grep 'slices=6' | sed 's/.*quantise/quantise/')" >> $O/k2_edge_ab.txt
	done
done
echo "4992 genes (no edge): $(python tools/k2i8_time.py tools/exp/nrm_gram_i8_QI_EDGE_0.so 4992 10000 6 2>&1 | grep 'slices=6' | sed 's/.*quantise/quantise/')" >> $O/k2_edge_ab.txt
cat $O/k2_edge_ab.txt
for th in unset 1 8 256; do
	if [ $th = unset ]; then python tools/time_single1_routes.py 10 >> $O/single1_routes.txt 2>&1
	else OPENBLAS_NUM_THREADS=$th OMP_NUM_THREADS=$th MKL_NUM_THREADS=$th python tools/time_single1_routes.py 10 >> $O/single1_routes.txt 2>&1; fi
done
grep -v "amdgpu.ids" $O/single1_routes.txt
