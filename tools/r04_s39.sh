#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -k "generation3" 2>&1 | tail -3
VIRTEX_AMD_GEN3_PERS=0 timeout 600 python tools/bench_gen3.py > gpurun_out/r04_s39_gen3_nopers.txt 2>&1
timeout 600 python tools/bench_gen3.py > gpurun_out/r04_s39_gen3_pers.txt 2>&1
paste <(grep "^gemm" gpurun_out/r04_s39_gen3_nopers.txt | cut -c1-45,72-96) <(grep "^gemm" gpurun_out/r04_s39_gen3_pers.txt | cut -c72-96)
timeout 1200 python tools/ab_step.py --steps 20 --rounds 4 nopers:sw.gen3_pers=0 pers:sw.gen3_pers=256 > gpurun_out/r04_s39_ab_pers.txt 2>&1
grep -v amdgpu gpurun_out/r04_s39_ab_pers.txt
