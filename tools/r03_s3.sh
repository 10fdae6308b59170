#!/bin/bash
# Round 3, GPU session 3: full GPU suite (complete output), in-process interleaved A/B of the transposing-read variants and
# the streaming 3x3 weight gradient, per-layer numbers of the implicit-GEMM 3x3 on the fixed pipeline, counter passes
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
export VIRTEX_AMD_FUSE_STEM_FWD=1
python -m pytest tests -q -m gpu 2>&1 | grep -v "^E  " | tail -60 > gpurun_out/s3_gpu_tests.txt
python tools/ab_step.py --rounds 3 --steps 20 builtin:lib=libvirtex_amd_trbuiltin.so,sw.wgrad3x3=0 noladder:lib=libvirtex_amd_noladder.so,sw.wgrad3x3=0 \
   asm:sw.wgrad3x3=0 asm_w3:sw.wgrad3x3=1 asm_w3all:sw.wgrad3x3=2 asm_w3_sk256:sw.wgrad3x3=1,sw.splitk_blocks=256 asm_w3_sk768:sw.wgrad3x3=1,sw.splitk_blocks=768 \
   > gpurun_out/s3_ab.txt 2> gpurun_out/s3_ab.err
VIRTEX_AMD_WGRAD3X3=0 python tools/bench_layers.py > gpurun_out/s3_layers_asm_implicit3x3.txt 2>&1
python tools/bench_layers.py > gpurun_out/s3_layers_asm.txt 2>&1
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/s3_pmc; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set -d $O/p$i -- python $R/tools/probe_wgrad.py > $O/p$i.log 2>&1
  db=$(find $O/p$i -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/pmc_dump.py $db "" > $O/p$i.txt 2>&1
done
find $O -name "*.db" -delete; rm -rf $O/p?/
cd $R
python bench.py --no-cpu-baseline --no-fidelity --steps 30 --warmup 10 > gpurun_out/s3_bench.json 2> gpurun_out/s3_bench.err
cat gpurun_out/s3_gpu_tests.txt | tail -15; cat gpurun_out/s3_ab.txt
