"""Do the two caption heads' backward passes overlap on the GPU when NO profiler is attached?

HIP events are recorded on each head's own stream at the start of its loss node's backward and at the end of its decoder
node's backward; their times relative to a common base event show whether the two chains run side by side (rocprofv3's
kernel trace shows them one after the other, torch.profiler's shows them side by side).
Usage (GPU box): python tools/head_overlap.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    import virtex_amd.factories as vf
    from virtex_amd import distributed as vd, models
    from virtex_amd.modules import textual_heads as th
    from virtex_amd.optim import FusedPretrainOptimizer

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = vf.build_bicaptioning_model(dropout=0.1, compute_dtype=torch.bfloat16).to(dev).train()
    buckets = vd.GradientBuckets(model)
    opt = FusedPretrainOptimizer(model, buckets, start_step=100)
    batches = [bench.device_batch(256, dev, i) for i in range(2)]
    marks = []

    def ev(tag):
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        marks.append((tag, torch.cuda.current_stream().cuda_stream, e))

    ce = models._FusedTiedCrossEntropyFn
    ce_bwd, dec_bwd = ce.backward, th._DecoderFn.backward

    def ce_backward(ctx, gout):
        ev("loss node start")
        return ce_bwd(ctx, gout)

    def dec_backward(ctx, dhid):
        out = dec_bwd(ctx, dhid)
        ev("decoder node end")
        return out
    ce.backward = staticmethod(ce_backward)
    th._DecoderFn.backward = staticmethod(dec_backward)

    def step(i):
        buckets.zero(); buckets.begin()
        out = model(batches[i % 2])
        out["loss"].backward()
        opt.step(grad_scale=buckets.finish())

    for i in range(10):
        step(i)
    torch.cuda.synchronize()
    for i in range(4):
        del marks[:]
        base = torch.cuda.Event(enable_timing=True)
        base.record()
        step(i)
        end = torch.cuda.Event(enable_timing=True)
        end.record()
        torch.cuda.synchronize()
        print(f"step {i}: {base.elapsed_time(end):.2f} ms")
        for tag, st, e in marks:
            print(f"   stream {st:#x}  {tag:18s} t = {base.elapsed_time(e):7.3f} ms")


if __name__ == "__main__":
    main()
