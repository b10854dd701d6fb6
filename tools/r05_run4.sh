#!/bin/bash
# A/B of the sparse-design kernel variants (tools/build_variants.sh nrm_de_sparse ...) + the test that failed
export TMPDIR=/tmp
python -m pytest tests/test_gpu_round5.py -q -k "without_torch" 2>&1 | tail -3
bash tools/r05_ab.sh nrm_de_sparse tools/time_de_sparse.py 2 > /dev/null
cat gpurun_out/r05ab/ab.txt
