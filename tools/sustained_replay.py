"""Sustained launch replay: N replays of the recorded step (what bench.py times), ms/step per 500 replays, the loss trajectory and
the allocator / host memory at the start and at the end -- a recording that leaked (autograd nodes, tensors, events) or drifted
would show here (round 4's replay grew the autograd graph by 4 nodes per replay: fixed in round 5; this is the standing check).
    python tools/sustained_replay.py [replays = 6000]"""
import os
import resource
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import virtex_amd.factories as vf
from virtex_amd import distributed as vd, synthetic
from virtex_amd.optim import FusedPretrainOptimizer
from virtex_amd.replay import StepReplay

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = vf.build_bicaptioning_model(compute_dtype=torch.bfloat16).to(dev).train()
buckets = vd.GradientBuckets(model)
opt = FusedPretrainOptimizer(model, buckets, start_step=100)
batches = [synthetic.synthetic_batch(256, dev, seed=i) for i in range(4)]
step = StepReplay(model, buckets, opt, batches[0], warmup=2, validate=True)
torch.cuda.synchronize()
mem0, rss0 = torch.cuda.memory_allocated(dev), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
times, losses = [], []
for blk in range(n // 500):
    t0 = time.perf_counter()
    for i in range(500):
        loss = step(batches[i % 4])
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 2)
    losses.append(round(loss.item(), 3))
step.sync()
print("ms/step per 500 replays:", " ".join(f"{t:.2f}" for t in times))
print("loss after each block  :", losses)
print(f"device memory allocated: {mem0 / 2**30:.2f} -> {torch.cuda.memory_allocated(dev) / 2**30:.2f} GiB; host max RSS {rss0 / 2**20:.2f} -> "
      f"{resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20:.2f} GiB; optimizer step index {opt.step_idx}; replays {step.replays}")
assert all(torch.isfinite(p).all() for p in model.parameters())
print("all parameters finite")
