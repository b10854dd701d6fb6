#!/bin/bash
# tools/build_variants.sh <stem> <name>=<"-DA=1 -DB=2"> ...  ->  tools/exp/<stem>_<name>.so (the other objects from tools/exp/obj, as build_exp.sh leaves them)
set -e
cd "$(dirname "$0")/.."
stem=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -mllvm -amdgpu-mfma-vgpr-form=1"
mkdir -p tools/exp/obj tools/exp/var
for f in normalisr_amd/csrc/*.hip; do
	o=tools/exp/obj/$(basename $f .hip).o
	if [ "$(basename $f .hip)" != "$stem" ] && { [ ! -f $o ] || [ $f -nt $o ] || [ include/normalisr_hip.h -nt $o ]; }; then
		/opt/rocm/bin/hipcc $FLAGS -c $f -o $o &
	fi
done
wait
objs=""
for f in normalisr_amd/csrc/*.hip; do b=$(basename $f .hip); [ "$b" != "$stem" ] && objs="$objs tools/exp/obj/$b.o"; done
for spec in "$@"; do
	name=${spec%%=*}; defs=${spec#*=}
	( /opt/rocm/bin/hipcc $FLAGS $defs -c normalisr_amd/csrc/$stem.hip -o tools/exp/var/${stem}_$name.o
	  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o tools/exp/${stem}_$name.so $objs tools/exp/var/${stem}_$name.o ) &
done
wait
ls tools/exp/${stem}_*.so
