#!/bin/bash
# Round 5, GPU session 5: where the fused conv3 backward spends its time -- ablation builds (VTX_CB_ABL bits: 1 no weight
# gradient, 2 no input-gradient MFMAs, 4 no epilogue, 8 transform = copy, 16 no dz / x3 loads)
cd $GRAFT_REPO_ROOT
for v in "" abl1 abl2 abl4 abl8 abl16 abl24 abl7 abl15; do
  L=virtex_amd/lib/libvirtex_amd${v:+_$v}.so
  echo -n "${v:-default}: " >> gpurun_out/r05_s5_ablation.txt
  VIRTEX_AMD_LIB=$PWD/$L timeout 200 python tools/bench_conv3_bwd.py 2>&1 | grep "^fused" >> gpurun_out/r05_s5_ablation.txt
done
cat gpurun_out/r05_s5_ablation.txt
