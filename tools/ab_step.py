"""Step-level A/B inside ONE process: the bench step (bench.py's `step`) timed under several settings of the Python-level
switches of virtex_amd.modules.visual_backbones / models, interleaved round after round (the boxes of the pool differ by
+-0.3 ms and drift with temperature: only interleaved rounds in one process decide anything below 1 %).

    python tools/ab_step.py [--steps 20] [--rounds 3] [--batch 256] name[:FLAG=v,FLAG=v] ...

FLAG is an attribute of visual_backbones (FUSE_STEM_FWD, RELU_BITS, STEM_STATS, ...) or `models.X` / `textual.X`.
Prints per variant: every round's ms/step, the minimum and the median."""
import argparse
import contextlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--visual", default="torchvision::resnet50")
    ap.add_argument("--textual", default="transdec_postnorm::L1_H1024_A16_F4096")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    import virtex_amd.factories as vf
    from virtex_amd import distributed as vd, models
    from virtex_amd.modules import textual_heads, visual_backbones as vb
    from virtex_amd.optim import FusedPretrainOptimizer
    from virtex_amd.synthetic import synthetic_batch

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = vf.build_bicaptioning_model(visual=a.visual, textual=a.textual, dropout=0.1, compute_dtype=torch.bfloat16).to(dev).train()
    buckets = vd.GradientBuckets(model)
    opt = FusedPretrainOptimizer(model, buckets, start_step=100)
    batches = [synthetic_batch(a.batch, dev, image_size=224, max_len=30, vocab_size=10000, seed=i) for i in range(2)]

    def step(i):
        buckets.zero(); buckets.begin()
        out = model(batches[i % 2])
        out["loss"].backward()
        opt.step(grad_scale=buckets.finish())

    import ctypes
    from virtex_amd import _lib

    class _Switch:                      # `sw.NAME=v`: vtx_set_switch of the library loaded at that moment; `lib=PATH`: which library
        pass
    default_lib = os.environ.get("VIRTEX_AMD_LIB", _lib.DEFAULT_LIB)
    from virtex_amd import ops as _ops
    mods = {"models": models, "textual": textual_heads, "splitk": _ops.splitk_batch}      # `splitk.enabled=0`: one reduce launch per weight gradient
    variants = []
    for v in a.variants:
        name, _, flags = v.partition(":")
        kv, sw, lib = [], [], default_lib
        for f in filter(None, flags.split(",")):
            k, val = f.split("=")
            if k in ("wgrad_cus", "branch_cus", "compute_cus"):   # CU-masked streams: wgrad [0, n), branch [0, n), compute [256 - n, 256)
                kv.append(("__" + k + "__", None, int(val)))
            elif k == "serial":                       # serial=1: every kernel on the compute stream (what the kernels cost without overlap)
                kv.append(("__serial__", None, int(val)))
            elif k == "lib":
                lib = val if os.path.isabs(val) else os.path.join(os.path.dirname(_lib.DEFAULT_LIB), val)
            elif k.startswith("sw."):
                sw.append((k[3:], int(val)))
            else:
                mod, attr = (mods[k.split(".")[0]], k.split(".")[1]) if "." in k else (vb, k)
                kv.append((mod, attr, type(getattr(mod, attr))(int(val))))
        variants.append((name, kv, sw, lib))
    from virtex_amd import streams as _streams

    def _set_serial(on):
        _streams.wgrad_stream.enabled = not on
        _streams.branch_stream.enabled = not on
        models.HEAD_STREAMS = not on
    plain = {"wgrad": None, "branch": None}
    masked = {}

    def _set_masks(w, b):
        """swap the side streams of the device for CU-masked ones (0 = the ordinary streams)"""
        torch.cuda.synchronize()
        if plain["wgrad"] is None:
            plain["wgrad"] = _streams.wgrad_stream.side(dev)
            plain["branch"] = _streams.branch_stream.side(dev)
        for kind, n, table in (("wgrad", w, _streams._side_streams), ("branch", b, _streams._branch_streams)):
            if n:
                if (kind, n) not in masked:
                    masked[(kind, n)] = _streams.masked_stream(dev, 0, n)
                table[dev] = masked[(kind, n)]
            else:
                table[dev] = plain[kind]
    compute_streams = {}
    defaults = {(m, k): getattr(m, k) for _, kv, _, _ in variants for m, k, _ in kv if not (isinstance(m, str) and m.startswith("__"))}
    sw_defaults = {"wgrad3x3": int(os.environ.get("VIRTEX_AMD_WGRAD3X3", "1")), "stem_stream": 1, "expand1x1": 1, "splitk_blocks": 512, "mc_eff128": 70, "bn_fin_wide": 0, "stats_tile": 4, "bn_adj": 1, "bn_grid": 8192, "tile64x256": 1, "tile_order": 0, "conv3x3_shared": 1, "gen3": 80, "gen3_mc": 800, "gen3_s2": 0, "bn_red_adj": 0, "epi_regs": 0, "gen3_pers": 0, "conv3_bwd": 1}
    res = {n: [] for n, _, _, _ in variants}
    for r in range(a.rounds):
        for name, kv, sw, lib in variants:
            torch.cuda.synchronize()
            _lib.use_library(lib)
            for k, d in sw_defaults.items():
                if hasattr(_lib.lib(), "vtx_set_switch"):
                    _lib.lib().vtx_set_switch(k.encode(), ctypes.c_int(d))
            for k, val in sw:
                _lib.call("vtx_set_switch", k.encode(), ctypes.c_int(val))
            for (m, k), d in defaults.items():
                setattr(m, k, d)
            _set_serial(False)
            masks = {"wgrad": 0, "branch": 0, "compute": 0}
            for m, k, val in kv:
                if m == "__serial__":
                    _set_serial(bool(val))
                elif isinstance(m, str):
                    masks[m.strip("_").split("_")[0]] = val
                else:
                    setattr(m, k, val)
            _set_masks(masks["wgrad"], masks["branch"])
            cs = None
            if masks["compute"]:
                cs = compute_streams.setdefault(masks["compute"], _streams.masked_stream(dev, 256 - masks["compute"], 256))
            with torch.cuda.stream(cs) if cs is not None else contextlib.nullcontext():
                for i in range(a.warmup):
                    step(i)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(a.steps):
                    step(i)
                torch.cuda.synchronize()
                res[name].append((time.perf_counter() - t0) / a.steps * 1e3)
    for name, _, _, _ in variants:
        v = sorted(res[name])
        print(f"{name:28s} " + " ".join(f"{x:7.3f}" for x in res[name]) + f"   min {v[0]:7.3f}  median {v[len(v) // 2]:7.3f} ms/step", flush=True)


if __name__ == "__main__":
    main()
