#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -k "generation3" 2>&1 | tail -3
timeout 600 python tools/bench_gen3.py > gpurun_out/r04_s38_gen3_regs.txt 2>&1
VIRTEX_AMD_KFLAGS=256 timeout 600 python tools/bench_gen3.py > gpurun_out/r04_s38_gen3_strips.txt 2>&1
timeout 1200 python tools/ab_step.py --steps 20 --rounds 4 strips:sw.epi_strips=1 regs:sw.epi_strips=0 > gpurun_out/r04_s38_ab_epi.txt 2>&1
grep -v amdgpu gpurun_out/r04_s38_ab_epi.txt
paste <(grep "^gemm" gpurun_out/r04_s38_gen3_strips.txt | cut -c1-45,72-123) <(grep "^gemm" gpurun_out/r04_s38_gen3_regs.txt | cut -c72-123)
