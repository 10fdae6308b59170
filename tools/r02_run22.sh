#!/bin/bash
# the other BASELINE.json configurations on one GPU + fp32 parity mode + eval-mode throughput, final build
R=$GRAFT_REPO_ROOT; cd $R
o=gpurun_out/other_configs.txt; rm -f $o
run() { echo "## $*" >> $o; timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 30 --warmup 10 "$@" 2>> gpurun_out/other_configs.err | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], 'ms/step', r['value'], 'img/s', r['config'])" >> $o; }
run --textual transdec_postnorm::L4_H1024_A16_F4096 --batch 128
run --visual torchvision::resnet101 --textual transdec_postnorm::L1_H2048_A32_F8192 --batch 64
run --dtype fp32 --batch 64
timeout 300 python tools/bench_infer.py >> $o 2>> gpurun_out/other_configs.err
cat $o
