#!/bin/bash
# Round 5, GPU session 2: conv3_bwd with the register refill inside the transform -- timing, PMC passes (HBM bytes, LDS
# conflicts, wave-time split) on the kernel alone, step A/B.
set -x
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 200 python tools/bench_conv3_bwd.py > gpurun_out/r05_s2_conv3_bwd.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
  T=$(echo $C | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $C -d $R/gpurun_out/pmc_$T -- python $R/tools/bench_conv3_bwd.py > $R/gpurun_out/r05_s2_pmc_$T.log 2>&1
  python $R/tools/pmc_dump.py $(find $R/gpurun_out/pmc_$T -name "*.db" | head -1) conv3_bwd >> $R/gpurun_out/r05_s2_pmc.txt 2>&1
  python $R/tools/pmc_dump.py $(find $R/gpurun_out/pmc_$T -name "*.db" | head -1) bn_bwd_apply_fused >> $R/gpurun_out/r05_s2_pmc.txt 2>&1
  rm -rf $R/gpurun_out/pmc_$T
done
cd $R
timeout 300 python tools/ab_step.py --steps 20 --rounds 3 fused off:FUSE_CONV3_BWD=0 > gpurun_out/r05_s2_ab.txt 2>&1
cat gpurun_out/r05_s2_conv3_bwd.txt gpurun_out/r05_s2_pmc.txt gpurun_out/r05_s2_ab.txt
