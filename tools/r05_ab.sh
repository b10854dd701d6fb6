#!/bin/bash
# A/B of experiment builds of one source: tools/r05_ab.sh <stem> <script.py> [rounds]
export TMPDIR=/tmp
O=gpurun_out/r05ab; mkdir -p $O; : > $O/ab.txt
for r in $(seq 1 ${3:-2}); do
for lib in tools/exp/$1_*.so; do
	echo "== $lib (round $r)" >> $O/ab.txt
	python tools/with_lib.py $lib $2 2>&1 | grep "DE_SPARSE=1\|ms" | grep -v "DE_SPARSE=0" | tail -n 2 >> $O/ab.txt
done
done
cat $O/ab.txt
