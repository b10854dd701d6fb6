#!/bin/bash
# Time the streaming Gram kernel on the C3 shape (run on the GPU box): tools/k2s_sweep.sh [nz ...]
cd "$(dirname "$0")/.."
for nz in ${@:-21 16 32}; do python tools/k2s_time.py - $nz 2>&1 | grep -v amdgpu.ids; done
