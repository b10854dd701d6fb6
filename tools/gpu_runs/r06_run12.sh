#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06m
mkdir -p $O
python -m pytest tests -q -x -m gpu -k "bench_self or normvar or chain" > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log; tail -n 6 $O/t.log
