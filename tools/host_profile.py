"""Where does the host's time per step go?  At a tiny batch the GPU is far ahead, so the wall time of a step IS the host's
enqueue time; cProfile over 20 such steps, sorted by own time.   python tools/host_profile.py [--batch 8] [--top 45]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--visual", default="torchvision::resnet50")
    ap.add_argument("--textual", default="transdec_postnorm::L1_H1024_A16_F4096")
    a = ap.parse_args()
    import bench
    import virtex_amd.factories as vf
    from virtex_amd import distributed as vd
    from virtex_amd.optim import FusedPretrainOptimizer

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = vf.build_bicaptioning_model(visual=a.visual, textual=a.textual, dropout=0.1, compute_dtype=torch.bfloat16).to(dev).train()
    buckets = vd.GradientBuckets(model)
    opt = FusedPretrainOptimizer(model, buckets, start_step=100)
    batches = [bench.device_batch(a.batch, dev, i) for i in range(2)]

    def step(i):
        buckets.zero(); buckets.begin()
        out = model(batches[i % 2])
        out["loss"].backward()
        opt.step(grad_scale=buckets.finish())

    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    host = (time.perf_counter() - t0) / a.steps * 1e3
    torch.cuda.synchronize()
    print(f"host enqueue time per step at batch {a.batch}: {host:.2f} ms")
    pr = cProfile.Profile()
    pr.enable()
    for i in range(a.steps):
        step(i)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats("tottime").print_stats(a.top)


if __name__ == "__main__":
    main()
