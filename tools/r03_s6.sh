#!/bin/bash
# Round 3, GPU session 6: tile choice of the k-major kernels on the repaired pipeline (step-level A/B), configs 4 / 5 of BASELINE.json
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python tools/ab_step.py --rounds 3 --steps 20 eff84:sw.mc_eff128=84 eff100:sw.mc_eff128=100 eff115:sw.mc_eff128=115 eff200:sw.mc_eff128=200 \
   eff115_sk768:sw.mc_eff128=115,sw.splitk_blocks=768 eff115_sk1024:sw.mc_eff128=115,sw.splitk_blocks=1024 > gpurun_out/s6_ab.txt 2> gpurun_out/s6_ab.err
python bench.py --no-cpu-baseline --no-fidelity --textual transdec_postnorm::L4_H1024_A16_F4096 --batch 128 --steps 30 --warmup 10 > gpurun_out/s6_bench_L4_bs128.json 2> gpurun_out/s6_bench_L4.err
python bench.py --no-cpu-baseline --no-fidelity --visual torchvision::resnet101 --textual transdec_postnorm::L1_H2048_A32_F8192 --batch 64 --steps 30 --warmup 10 > gpurun_out/s6_bench_R101_H2048_bs64.json 2> gpurun_out/s6_bench_R101.err
python tools/host_trace.py > gpurun_out/s6_host_trace.txt 2>&1
cat gpurun_out/s6_ab.txt; head -c 250 gpurun_out/s6_bench_L4_bs128.json; echo; head -c 250 gpurun_out/s6_bench_R101_H2048_bs64.json
