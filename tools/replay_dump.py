"""The recorded launch list of one training step (virtex_amd.replay), one line per op: C entry point / stream or event
operation (with the stream it was issued on) / ATen operator.  python tools/replay_dump.py [--batch 8] > list.txt"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    import virtex_amd.factories as vf
    from virtex_amd import distributed as vd
    from virtex_amd.optim import FusedPretrainOptimizer
    from virtex_amd.replay import StepReplay
    from virtex_amd.synthetic import synthetic_batch
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = vf.build_bicaptioning_model(dropout=0.1, compute_dtype=torch.bfloat16).to(dev).train()
    buckets = vd.GradientBuckets(model)
    opt = FusedPretrainOptimizer(model, buckets, start_step=100)
    batch = synthetic_batch(a.batch, dev, image_size=224, max_len=30, vocab_size=10000, seed=0)
    rp = StepReplay(model, buckets, opt, batch, warmup=2, validate=False)
    print(f"# {len(rp.rec.ops)} ops: {rp.rec.counts}")
    for i, l in enumerate(rp.rec.labels):
        print(f"{i:5d} {l}")


if __name__ == "__main__":
    main()
