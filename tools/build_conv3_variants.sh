#!/bin/bash
# Measurement builds of ONE file: libvirtex_amd_<tag>.so = the product objects with conv3_bwd.o recompiled under extra flags.
#   tools/build_conv3_variants.sh abl1 -DVTX_CB_ABL=1   abl2 -DVTX_CB_ABL=2 ...   (pairs: tag flag)
set -e
cd "$(dirname "$0")/.."
python -m virtex_amd.build > /dev/null
L=virtex_amd/lib
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=fast -Wno-unused-result"
while [ $# -ge 2 ]; do
  tag=$1; flag=$2; shift 2
  mkdir -p $L/obj_$tag
  /opt/rocm/bin/hipcc $FLAGS $flag -c virtex_amd/csrc/conv3_bwd.hip -o $L/obj_$tag/conv3_bwd.o
  objs=$(ls $L/obj/*.o | grep -v conv3_bwd.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libvirtex_amd_$tag.so $objs $L/obj_$tag/conv3_bwd.o
  echo built $tag
done
