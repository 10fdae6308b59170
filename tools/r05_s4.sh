#!/bin/bash
# Round 5, GPU session 4: conv3_bwd variants -- default | VTX_CB_LEAN (two-FMA transform, mask from y2) | -fno-slp-vectorize | both
set -x
cd $GRAFT_REPO_ROOT
for v in "" lean noslp leannoslp; do
  L=virtex_amd/lib/libvirtex_amd${v:+_$v}.so
  echo "=== ${v:-default}" >> gpurun_out/r05_s4_variants.txt
  VIRTEX_AMD_LIB=$PWD/$L timeout 200 python tools/bench_conv3_bwd.py >> gpurun_out/r05_s4_variants.txt 2>&1
done
cat gpurun_out/r05_s4_variants.txt
