#!/bin/bash
# Build experiment variants of the library: tools/build_exp.sh <source.hip> <MACRO> <v1> [v2 ...]  ->  tools/exp/<stem>_<MACRO>_<v>.so
# Every other source is compiled once into tools/exp/obj/ and reused; only <source.hip> is recompiled per variant (-D<MACRO>=<v>).
set -e
cd "$(dirname "$0")/.."
SRC=$1; MACRO=$2; shift 2
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -mllvm -amdgpu-mfma-vgpr-form=1 $EXTRA"
mkdir -p tools/exp/obj tools/exp/var
stem=$(basename $SRC .hip)
for f in normalisr_amd/csrc/*.hip; do
	o=tools/exp/obj/$(basename $f .hip).o
	if [ "$(basename $f .hip)" != "$stem" ] && { [ ! -f $o ] || [ $f -nt $o ] || [ normalisr_amd/csrc/nrm_common.h -nt $o ] || [ normalisr_amd/csrc/nrm_gram_sched.h -nt $o ] || [ include/normalisr_hip.h -nt $o ]; }; then
		/opt/rocm/bin/hipcc $FLAGS -c $f -o $o &
	fi
done
wait
objs=""
for f in normalisr_amd/csrc/*.hip; do
	b=$(basename $f .hip)
	[ "$b" != "$stem" ] && objs="$objs tools/exp/obj/$b.o"
done
for v in "$@"; do
	( /opt/rocm/bin/hipcc $FLAGS -D$MACRO=$v -c normalisr_amd/csrc/$stem.hip -o tools/exp/var/${stem}_${MACRO}_$v.o
	  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o tools/exp/${stem}_${MACRO}_$v.so $objs tools/exp/var/${stem}_${MACRO}_$v.o ) &
done
wait
ls -la tools/exp/${stem}_${MACRO}_*.so
