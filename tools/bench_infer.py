"""Eval-mode throughput: the frozen / feature-extraction backbone (BatchNorm folded into
the convolutions, ReLU and residual in the epilogue) and the whole model's validation forward."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import virtex_amd.factories as vf  # noqa: E402
from virtex_amd import synthetic  # noqa: E402


def timed(fn, iters, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda:0")
    model = vf.build_bicaptioning_model(compute_dtype=torch.bfloat16).to(dev).eval()
    batch = synthetic.synthetic_batch(B, dev)
    with torch.no_grad():
        t_vis = timed(lambda: model.visual(batch["image"]), 20, 5)
        t_all = timed(lambda: model(batch), 10, 3)
    gflop = 2 * 4.087  # forward conv MACs per image
    print(json.dumps({"backbone_eval_images_per_sec": B / t_vis, "ms": t_vis * 1e3,
                      "backbone_tflops": B * gflop / t_vis / 1e3, "validation_forward_images_per_sec": B / t_all,
                      "validation_ms": t_all * 1e3, "batch": B, "dtype": "bf16"}))


if __name__ == "__main__":
    main()
