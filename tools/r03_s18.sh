#!/bin/bash
# Round 3, GPU session 18: what the side streams are worth now, and a low-priority weight-gradient stream (separate processes on one box, alternating)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-fidelity --no-roofline --steps 30 --warmup 10"
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')" > gpurun_out/s18.txt 2>&1
for rep in 1 2; do
  for v in "default" "VIRTEX_AMD_WGRAD_PRIORITY=1" "VIRTEX_AMD_WGRAD_PRIORITY=-1" "VIRTEX_AMD_WGRAD_STREAM=0" "VIRTEX_AMD_BRANCH_STREAM=0" "VIRTEX_AMD_HEAD_STREAMS=0"; do
    if [ "$v" = default ]; then r=$($B 2>/dev/null | tail -1); else r=$(env $v $B 2>/dev/null | tail -1); fi
    echo "$v $(echo $r | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')" >> gpurun_out/s18.txt
  done
done
cat gpurun_out/s18.txt
