#!/bin/bash
cd $GRAFT_REPO_ROOT
export VIRTEX_AMD_FORCE_DIST=nccl WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 VIRTEX_AMD_DP_PAYLOAD=bf16 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python bench.py --gpus 1 --steps 3 --warmup 2 --batch 8 --image-size 64 --vocab-size 1000 --textual transdec_postnorm::L1_H128_A2_F256 --no-cpu-baseline --roofline-steps 1 --dropout 0.0 --launch replay > gpurun_out/r05_s7_out.txt 2> gpurun_out/r05_s7_err.txt
echo rc=$?
grep -v "Warning\|warn" gpurun_out/r05_s7_err.txt | tail -30
