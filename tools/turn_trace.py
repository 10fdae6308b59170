"""Which stream waits make the two caption heads' backward passes run one after the other?

One step under torch.profiler; prints, in host order, every autograd node boundary, every hipStreamWaitEvent /
hipEventRecord and every kernel launch (with the stream it went to) from the start of backward to the backbone's node.
Usage (GPU box): python tools/turn_trace.py > gpurun_out/turn_trace.txt"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    import virtex_amd.factories as vf
    from virtex_amd import distributed as vd
    from virtex_amd.optim import FusedPretrainOptimizer
    from torch.profiler import ProfilerActivity, profile

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = vf.build_bicaptioning_model(dropout=0.1, compute_dtype=torch.bfloat16).to(dev).train()
    buckets = vd.GradientBuckets(model)
    opt = FusedPretrainOptimizer(model, buckets, start_step=100)
    batches = [bench.device_batch(256, dev, i) for i in range(2)]

    def step(i):
        buckets.zero(); buckets.begin()
        out = model(batches[i % 2])
        out["loss"].backward()
        opt.step(grad_scale=buckets.finish())

    for i in range(6):
        step(i)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step(0)
        step(1)
        torch.cuda.synchronize()
    path = os.path.join(ROOT, "gpurun_out", "turn_trace.json")
    prof.export_chrome_trace(path)
    ev = json.load(open(path))["traceEvents"]
    os.remove(path)
    kern = {}
    for e in ev:
        if e.get("cat") in ("kernel", "gpu_memcpy", "gpu_memset") and "args" in e:
            kern[e["args"].get("correlation")] = e
    nodes = [e for e in ev if e.get("cat") == "cpu_op" and e.get("name", "").startswith("autograd::engine::evaluate_function")]
    if not nodes:
        print("no autograd nodes in the trace"); return
    # second step's backward
    firsts = [e for e in nodes if "AddBackward" in e["name"]]
    t0 = firsts[-1]["ts"]
    ends = [e for e in nodes if "_ResNetFn" in e["name"] and e["ts"] > t0]
    t1 = ends[0]["ts"] if ends else t0 + 20000
    rows = []
    for e in ev:
        ts = e.get("ts")
        if ts is None or ts < t0 or ts > t1:
            continue
        cat, name = e.get("cat"), e.get("name", "")
        if cat == "cpu_op" and name.startswith("autograd::engine::evaluate_function"):
            rows.append((ts, "NODE " + name.split(": ", 1)[-1]))
        elif cat in ("cuda_runtime", "cuda_driver"):
            a = e.get("args", {})
            if "WaitEvent" in name or "EventRecord" in name or "Synchronize" in name or "Malloc" in name or "Free" in name:
                rows.append((ts, f"  {name} {({k: v for k, v in a.items() if k not in ('External id', 'cbid')})}"))
            elif "Launch" in name or "Memcpy" in name or "Memset" in name:
                k = kern.get(a.get("correlation"))
                if k is not None:
                    ka = k["args"]
                    rows.append((ts, f"  launch -> stream {ka.get('stream')}  gpu_start +{(k['ts'] - t0) / 1e3:8.3f} ms  "
                                     f"dur {k.get('dur', 0):7.1f} us  {k['name'][:70]}"))
    rows.sort(key=lambda r: r[0])
    for ts, line in rows:
        print(f"{(ts - t0) / 1e3:8.3f} ms  {line}")


if __name__ == "__main__":
    main()
