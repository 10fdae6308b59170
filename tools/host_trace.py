"""Is the host ahead of the GPU during the step, and where do the small copies come from?

1. host lead: at four milestones of every step (forward enqueued, backward enqueued, exchange finished, optimizer
   enqueued) the host clock and a GPU event on the compute stream are recorded; lead = (GPU time of the milestone) -
   (host time of the milestone), both relative to one common start.  A lead near zero means the host only just
   enqueued what the GPU is executing - gaps on the queues are then host time, not dependencies.
2. torch.profiler over two steps with Python stacks: which call sites launch aten::copy_ / fill_ / add kernels.

Usage (GPU box): python tools/host_trace.py [--steps 8]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--no-profiler", action="store_true")
    a = ap.parse_args()
    import bench
    import virtex_amd.factories as vf
    from virtex_amd import distributed as vd
    from virtex_amd.optim import FusedPretrainOptimizer

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = vf.build_bicaptioning_model(dropout=0.1, compute_dtype=torch.bfloat16).to(dev).train()
    buckets = vd.GradientBuckets(model)
    opt = FusedPretrainOptimizer(model, buckets, start_step=100)
    batches = [bench.device_batch(a.batch, dev, i) for i in range(2)]
    marks = []

    def mark(tag):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((tag, time.perf_counter(), e))

    def step(i, trace=False):
        if trace:
            mark("start")
        buckets.zero()
        buckets.begin()
        out = model(batches[i % 2])
        if trace:
            mark("forward")
        out["loss"].backward()
        if trace:
            mark("backward")
        scale = buckets.finish()
        opt.step(grad_scale=scale)
        if trace:
            mark("optimizer")

    for i in range(12):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    host_only = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"untraced: host enqueue {host_only / a.steps * 1e3:.2f} ms/step, wall {wall / a.steps * 1e3:.2f} ms/step")

    base = torch.cuda.Event(enable_timing=True)
    base.record()
    h0 = time.perf_counter()
    for i in range(a.steps):
        step(i, trace=True)
    torch.cuda.synchronize()
    print("milestone      host_ms   gpu_ms   lead_ms (gpu - host; > 0: the host was ahead)")
    for tag, th, e in marks:
        hm = (th - h0) * 1e3
        gm = base.elapsed_time(e)
        print(f"  {tag:10s} {hm:9.2f} {gm:9.2f} {gm - hm:8.2f}")

    if a.no_profiler:
        return
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step(0)
        step(1)
        torch.cuda.synchronize()
    want = ("aten::copy_", "aten::clone", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::zeros",
            "aten::contiguous", "aten::to", "aten::_to_copy", "aten::mul", "aten::sum")
    sites = {}
    for ev in prof.events():
        if ev.name not in want or ev.device_time_total <= 0 and not any(k.name for k in ev.kernels):
            continue
        frames = [f for f in (ev.stack or []) if "virtex_amd" in f or "bench.py" in f or "host_trace" in f]
        key = (ev.name, frames[0] if frames else (ev.stack[0] if ev.stack else "?"))
        o = sites.setdefault(key, [0, 0.0, set()])
        o[0] += 1
        o[1] += ev.device_time_total
        for k in ev.kernels:
            o[2].add(k.name[:60])
    print("\ncall sites of torch-native kernels over 2 steps (count, device us, kernels):")
    for (name, site), (n, us, ks) in sorted(sites.items(), key=lambda kv: -kv[1][0]):
        print(f"  {n:4d} {us:9.1f}  {name:18s} {site}  {sorted(ks)[:2]}")


if __name__ == "__main__":
    main()
