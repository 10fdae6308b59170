#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels.py -x -q -m gpu -k "feed_forward_epilogue or tied or generation3 or gemm" 2>&1 | tail -2
timeout 1200 python tools/ab_step.py --steps 20 --rounds 4 old:lib=libvirtex_amd_pre_pd.so new > gpurun_out/r04_s48_ab_pd.txt 2>&1
timeout 600 python tools/ab_step.py --steps 20 --rounds 2 old_serial:serial=1,lib=libvirtex_amd_pre_pd.so new_serial:serial=1 >> gpurun_out/r04_s48_ab_pd.txt 2>&1
grep -v amdgpu gpurun_out/r04_s48_ab_pd.txt
