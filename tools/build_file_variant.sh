#!/bin/bash
# Measurement builds of ONE source file: libvirtex_amd_<tag>.so = the product objects with <file>.o recompiled under extra flags.
#   tools/build_file_variant.sh gemm.hip all8 -DVTX_EPI_ALL_MAX=8
set -e
cd "$(dirname "$0")/.."
python -m virtex_amd.build > /dev/null
L=virtex_amd/lib
src=$1; tag=$2; shift 2
base=$(basename $src .hip)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=fast -Wno-unused-result"
mkdir -p $L/obj_$tag
/opt/rocm/bin/hipcc $FLAGS "$@" -c virtex_amd/csrc/$src -o $L/obj_$tag/$base.o
objs=$(ls $L/obj/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libvirtex_amd_$tag.so $objs $L/obj_$tag/$base.o
echo built $tag
