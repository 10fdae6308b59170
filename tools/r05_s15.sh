#!/bin/bash
# Round 5, GPU session 15: the join class with the epilogue operands of ALL four 16-row steps requested up front (VTX_EPI_ALL_MAX=8)
cd $GRAFT_REPO_ROOT
echo "== default" > gpurun_out/r05_s15_join_all8.txt
timeout 200 python tools/join_shapes_probe.py >> gpurun_out/r05_s15_join_all8.txt 2>&1
echo "== VTX_EPI_ALL_MAX=8" >> gpurun_out/r05_s15_join_all8.txt
VIRTEX_AMD_LIB=$PWD/virtex_amd/lib/libvirtex_amd_all8.so timeout 200 python tools/join_shapes_probe.py >> gpurun_out/r05_s15_join_all8.txt 2>&1
timeout 300 python tools/ab_step.py --steps 20 --rounds 3 base all8:lib=libvirtex_amd_all8.so >> gpurun_out/r05_s15_join_all8.txt 2>&1
grep -v amdgpu gpurun_out/r05_s15_join_all8.txt
