#!/bin/bash
# tools/k2s_exp.sh "<variants>" "<nz list>": time experiment builds of the streaming Gram kernel (GPU box)
cd "$(dirname "$0")/.."
for nz in $2; do for v in $1; do
	python tools/k2s_time.py tools/exp/nrm_gram_skinny_SK_EXP_$v.so $nz 2>&1 | grep -v amdgpu.ids | sed "s/^/exp=$v /"
done; done
