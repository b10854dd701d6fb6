#!/bin/bash
# fused (one-pass) sparse-design kernel: parity tests, then A/B of the covariate-operand batch size against the two-pass form
export TMPDIR=/tmp
O=gpurun_out/r05d
mkdir -p $O
python -m pytest tests/test_gpu_round4.py tests/test_gpu_round5.py -x -q -k "sparse or single4 or config3" > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log
tail -n 5 $O/t.log
for ub in 4 8 16; do
	echo "== DS_UB=$ub" >> $O/ab.txt
	python tools/with_lib.py tools/exp/nrm_de_sparse_DS_UB_$ub.so tools/time_de_sparse.py 2>&1 | grep -v "DE_SPARSE=0\|largest" >> $O/ab.txt
done
echo "== two passes (NRM_DE_SPARSE_SUMS=stream)" >> $O/ab.txt
NRM_DE_SPARSE_SUMS=stream python tools/time_de_sparse.py 2>&1 | grep -v "DE_SPARSE=0" >> $O/ab.txt
cat $O/ab.txt
