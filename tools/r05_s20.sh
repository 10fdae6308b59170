#!/bin/bash
# Round 5, GPU session 20: the bench step on shapes nobody tuned for (odd batches, other image sizes): does every picker / fused-path
# predicate fall back cleanly?  loss must be finite and the launch mode replay (the recording validates against the eager step)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
: > gpurun_out/r05_s20_shapes.txt
for ARGS in "--batch 33" "--batch 96" "--batch 200" "--batch 7" "--batch 64 --image-size 160" "--batch 48 --image-size 256" "--batch 31 --image-size 192" "--batch 40 --textual transdec_prenorm::L2_H1024_A16_F4096" "--batch 24 --visual torchvision::resnet101"; do
  OUT=$(timeout 200 python bench.py --no-cpu-baseline --no-fidelity --no-roofline --steps 4 --warmup 3 $ARGS 2>gpurun_out/r05_s20.err | tail -1)
  echo "$ARGS :: $(echo "$OUT" | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], 'img/s', d['ms_per_step'], 'ms', c['launch'], 'loss', c['final_loss'], c['launch_fallback_reason'])" 2>&1 | tail -1)" >> gpurun_out/r05_s20_shapes.txt
  grep -v "amdgpu.ids\|Warn\|key_padding" gpurun_out/r05_s20.err | tail -2 >> gpurun_out/r05_s20_shapes.txt
done
cat gpurun_out/r05_s20_shapes.txt
