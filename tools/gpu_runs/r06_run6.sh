#!/bin/bash
# round 6, sixth GPU call: K2 with the edge-tile wave map (QI_EDGE) -- parity tests of the Gram engines, then timing on configs[1] and on shapes with other edges
export TMPDIR=/tmp
O=gpurun_out/r06g
mkdir -p $O
python -m pytest tests -x -q -m gpu -k "gram or c2 or coex or schedule or g11 or g13 or g14 or config4 or bitwise or beyond or banded or large_cell" > $O/t_gram.log 2>&1; echo "rc=$?" >> $O/t_gram.log; tail -n 8 $O/t_gram.log
for shape in "5000 10000" "5001 10000" "5033 10000" "3750 100000" "4992 10000"; do
	python tools/k2i8_time.py - $shape 6 2>&1 | grep "slices=6" >> $O/k2_time.txt
done
cat $O/k2_time.txt
python bench.py --steps 20 --warmup 3 --no-extras --cpu-seconds 0 --e2e 0 2>&1 | grep "^{\"metric" | cut -c1-600
