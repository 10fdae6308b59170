#!/usr/bin/env python
"""Headline benchmark: pretrain images/sec of bicaptioning_R_50_L1_H1024 (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = the reference's whole training-step body (scripts/pretrain_virtex.py:145-163):
zero_grad -> forward (dropout 0.1 on) -> backward -> gradient all-reduce (N > 1) -> clip 10.0 ->
SGD(m .9, per-tensor lr/wd) -> Lookahead(5, .5) -> LR schedule (fused HIP optimizer kernels), on synthetic COCO-shaped batches
(224x224 fp32 images, 30-token captions) that are resident in HBM before the timed region.
Workload at N=1 = BASELINE.json configs[1]: bf16 compute, 256 images per GPU.  For N > 1 the
driver launches this file under torch.distributed.run, one rank per GPU (weak scaling).

Rank 0 prints ONE JSON line; besides the contract fields it carries
  "roofline"     : the step's dominant kernel -- the kernel (contraction-kernel instantiation, or BatchNorm / LayerNorm
                   / embedding / loss / optimizer ... kernel) with the largest summed time -- algorithmic FLOPs and bytes /
                   HIP-event time per launch measured live on the launch streams (vtx_profile_start/stop), against the
                   dense bf16 MFMA peak and the HBM peak; "hbm_kernels": GB/s, fraction of the HBM peak and ms per step of
                   every HBM-bound kernel family; "families": the same summed per family (BatchNorm = its five kernels);
                   "step_model": sum over all launches of max(FLOPs / MFMA peak, bytes / HBM peak) next to the measured
                   kernel time
  "step_mfma"    : whole-step algorithmic FLOP/s (35.17 GFLOP/img) against the same peak
  "fidelity"     : the bf16 step against the fp32 step (the mode pinned to the oracle) on the timed batch: loss and
                   per-tensor gradient distance / cosine (virtex_amd/fidelity.py)
  "cpu_baseline" : the oracle port of the reference step timed on this box's host cores (B = 16, and config 1's B = 2).
  "stock_pytorch_baseline" (--stock-pytorch-baseline only): the same port on THIS GPU through stock PyTorch-ROCm (MIOpen /
                   hipBLASLt / ATen, autocast bf16) -- what the reference's own code reaches on the part.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_IMG = {"L1_H1024": 35.17, "L4_H1024": 54.89}
PEAK_BF16_TFLOPS = 2500.0                                  # MI355X dense bf16 MFMA
PEAK_F32_TFLOPS = 157.3


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)       # >= 20 warm-up, >= 50 timed steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--textual", default="transdec_postnorm::L1_H1024_A16_F4096")
    ap.add_argument("--visual", default="torchvision::resnet50")
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stock-pytorch-baseline", action="store_true",
                    help="also time the reference's module graph (the oracle port) on this GPU through stock PyTorch-ROCm "
                         "(MIOpen / hipBLASLt, autocast bf16): `stock_pytorch_baseline` in the record")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-fidelity", action="store_true")
    ap.add_argument("--bn-fusion", default=None, choices=["none", "bwd", "fwd", "both"],
                    help="A/B switch: BatchNorm sums taken in the convolution epilogues (default: the module's setting)")
    ap.add_argument("--roofline-steps", type=int, default=3,
                    help="steps run with per-launch HIP events (after the timed region) for the roofline object")
    ap.add_argument("--serial-streams", action="store_true",
                    help="run everything on the compute stream (no weight-gradient / branch streams): the kernel-trace of "
                         "this mode shows every kernel un-contended, which is what the roofline object reports")
    ap.add_argument("--roofline-live", action="store_true",
                    help="take the per-launch events inside the timed region itself (adds the event overhead to `value`)")
    ap.add_argument("--launch", choices=("auto", "eager", "replay", "graph"), default="auto",
                    help="how the timed steps are issued.  eager: the Python step (autograd + ctypes per launch, ~10 ms of host "
                         "time per step).  replay: the same launches on the same three streams re-issued from a recorded list "
                         "(virtex_amd/replay.py; validated against the eager step at construction; ~2 ms of host time) -- what "
                         "BASELINE configs 4 / 5 need, neutral at bs 256.  graph: ONE hipGraph (virtex_amd/graph.py) -- measured "
                         "SLOWER than eager on ROCm 7.2 (profiles/r04_hipgraph_configs_2_4_5.txt).  auto: replay on single-process "
                         "GPU runs with the eager step as the fall-back")
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--vocab-size", type=int, default=10000)
    ap.add_argument("--cpu-batch", type=int, default=16)
    ap.add_argument("--cpu-steps", type=int, default=8)
    return ap.parse_args(argv)


def device_batch(B, dev, seed, image_size=224, vocab_size=10000):
    from virtex_amd.synthetic import synthetic_batch

    return synthetic_batch(B, dev, image_size=image_size, max_len=30, vocab_size=vocab_size, seed=seed)


PEAK_HBM_TBS = 8.0                                         # MI355X HBM3E (MI355X_MICROARCH.md)
# HBM bytes per launch of every kernel class in the DEFAULT workload, from separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE;
# units / corrections per the guide) of this same command -- profiles/traffic_table.json, written by tools/pmc_traffic.py --json
# (tools/round_end.sh) together with the sha256 of the kernel sources it was measured on.  A table measured on OTHER sources
# is not reported: roofline.traffic = null and an error on stderr (round 3 kept the numbers in this file, where a kernel change
# invalidated them silently).
TRAFFIC_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic_table.json")


# the workloads a table exists for: the default one and BASELINE configs 4 / 5 (one table per workload: the launches of a class
# differ in shape from workload to workload, so their bytes per launch do too)
TABLED_WORKLOADS = {
    (256, "bf16", "transdec_postnorm::L1_H1024_A16_F4096", "torchvision::resnet50", 1, 224, 10000): "",
    (128, "bf16", "transdec_postnorm::L4_H1024_A16_F4096", "torchvision::resnet50", 1, 224, 10000): "config4",
    (64, "bf16", "transdec_postnorm::L1_H2048_A32_F8192", "torchvision::resnet101", 1, 224, 10000): "config5",
}


def traffic_table_path(tag):
    """'' (the default workload) -> profiles/traffic_table.json; 'config4' -> profiles/traffic_table_config4.json"""
    return TRAFFIC_TABLE if not tag else TRAFFIC_TABLE[:-len(".json")] + f"_{tag}.json"


def load_traffic_table(path=None):
    """(per-launch bytes by class name, source string) or ({}, None) when the table is missing or stale."""
    path = path or TRAFFIC_TABLE
    rel = os.path.join("profiles", os.path.basename(path))
    try:
        with open(path) as fh:
            t = json.load(fh)
    except (OSError, ValueError):
        print(f"bench.py: ERROR -- {path} missing or unreadable: roofline.traffic = null (regenerate with tools/round_end.sh)", file=sys.stderr)
        return {}, None
    from virtex_amd.build import csrc_hash
    if t.get("csrc_sha256") != csrc_hash():
        print(f"bench.py: ERROR -- {rel} was measured on different kernel sources (sha256 mismatch): "
              "roofline.traffic = null until tools/round_end.sh regenerates it", file=sys.stderr)
        return {}, None
    table = dict(t.get("per_launch_bytes", {}))
    table["__per_kernel__"] = t.get("per_kernel", {})
    return table, f"{rel} ({t.get('source')}; {t.get('rule')})"


BEST_BATCH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "best_batch.json")


def load_best_batch():
    """`config.best_batch`: the per-GPU batch size at which THIS build measured its highest images/sec (tools/round_end.sh runs the
    same command at 320 / 384 / 512 images per GPU and writes profiles/best_batch.json with the hash of the kernel sources).  The
    headline stays BASELINE's bs = 256; this key says what the hardware gives when the batch is free.
    None when the file is missing or was measured on other kernel sources."""
    try:
        with open(BEST_BATCH) as fh:
            b = json.load(fh)
        from virtex_amd.build import csrc_hash
        if b.get("csrc_sha256") != csrc_hash():
            return {"stale": "profiles/best_batch.json was measured on other kernel sources", "batch": b.get("batch"), "value": b.get("value")}
        return {k: b[k] for k in ("batch", "value", "ms_per_step", "source", "others") if k in b}
    except (OSError, ValueError, KeyError):
        return None


def lookup_traffic(table, name, n_classes, launches_per_step):
    """HBM bytes per launch of the kernel the focused pass timed.  A launch family of ONE instantiation: the family's entry.
    A family of several (bn_bwd_apply: the flat kernel at two unrolls, the pooled one): the focused pass timed the largest
    single instantiation, identified in the table by its launches per step.  -> (bytes or None, kernel name or None)"""
    per_kernel = table.get("__per_kernel__", {}).get(name)
    if n_classes > 1 and per_kernel and launches_per_step is not None:
        hits = [(k, v) for k, v in per_kernel.items() if abs(v["launches_per_step"] - launches_per_step) < 0.5]
        if len(hits) == 1:
            return hits[0][1]["bytes"], hits[0][0]
        return None, None                         # ambiguous: better no figure than one for a different mix of launches
    return table.get(name), None


def _kernel_name(bracket):
    """'[BM = 256, ..., AL = vtxg::PlainKC<unsigned short, 2>, ...]' -> the name rocprofv3 prints."""
    body = bracket.strip()[1:-1] if bracket.startswith("[") else bracket
    parts, depth, cur = [], 0, ""
    for ch in body:
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur); cur = ""
        else:
            cur += ch
    parts.append(cur)
    kv = {}
    for p in parts:
        if "=" in p:
            k, v = p.split("=", 1)
            kv[k.strip()] = v.strip()
    # the kernel's own template parameter order (launch_v2 lists BK/STAGES earlier); __PRETTY_FUNCTION__ drops
    # defaulted template arguments, rocprofv3 prints them
    order = ["BM", "BN", "WM", "WN", "AL", "BL", "EP", "BK", "STAGES"]
    gen3 = "BM" not in kv                           # launch_v3<BN, AL, BL, EP>: the generation-3 kernels are named by their tile
    if gen3:
        order = ["AL", "BL", "EP"]
    name = ", ".join(kv[k] for k in order if k in kv).replace("vtxg::", "").replace("unsigned short", "bf16")
    name = name.replace("EpiStore<bf16>", "EpiStore<bf16, 0>").replace("EpiStore<float>", "EpiStore<float, 0>")
    return f"contraction_v3_256x{kv.get('BN', '?')}_kernel<{name}>" if gen3 else f"contraction_v2_kernel<{name}>"


FAMILY_OF = {   # HBM-bound kernel -> family reported under roofline.families
    "bn_fwd_reduce": "batchnorm", "bn_fwd_apply": "batchnorm", "bn_bwd_reduce": "batchnorm", "bn_bwd_apply": "batchnorm",
    "bn_finalize": "batchnorm", "layernorm_fwd": "layernorm", "layernorm_bwd": "layernorm", "layernorm_bwd_finalize": "layernorm",
    "embedding_fwd": "embedding", "embedding_bwd": "embedding", "cross_entropy_fwd": "cross_entropy",
    "cross_entropy_reduce": "cross_entropy", "cross_entropy_bwd": "cross_entropy", "tied_ce_fwd": "cross_entropy",
    "tied_ce_bwd": "cross_entropy", "optimizer_sumsq": "optimizer", "optimizer_sumsq_final": "optimizer",
    "optimizer_step": "optimizer", "maxpool_fwd": "maxpool", "maxpool_bwd": "maxpool", "colsum": "bias_gradients",
    "colsum_finalize": "bias_gradients", "splitk_reduce": "splitk_reduce", "weight_prep": "weight_prep",
    "image_to_nhwc": "input_conversion", "gelu_bwd": "gelu_bwd", "add": "add", "attention_fwd": "attention",
    "attention_bwd": "attention"}


def _display_name(rec_name):
    """The FAMILY-level name of a profile class: the launch family of a VTX_KLAUNCH site ("family:<name>|<kernel>|[T = ...]"),
    the kernel name of a contraction instantiation."""
    return rec_name[len("family:"):].split("|")[0] if rec_name.startswith("family:") else _kernel_name(rec_name)


def _instantiation_name(rec_name):
    """The single kernel instantiation a profile class stands for, as close to rocprofv3's spelling as the class name allows:
    'family:bn_bwd_apply|(bn_bwd_apply_fused_kernel<T, UNR, true>)|[T = unsigned short, UNR = 2]' ->
    'bn_bwd_apply_fused_kernel<bf16, 2, true>'."""
    if not rec_name.startswith("family:"):
        return _kernel_name(rec_name)
    import re
    parts = rec_name[len("family:"):].split("|")
    if len(parts) < 2 or not parts[1].strip():
        return parts[0]
    kern = parts[1].strip()
    while kern.startswith("(") and kern.endswith(")"):
        kern = kern[1:-1].strip()
    binds = {}
    if len(parts) > 2 and parts[2].startswith("["):
        body, depth, cur, items = parts[2].strip()[1:-1], 0, "", []
        for ch in body:
            depth += ch == "<"; depth -= ch == ">"
            if ch == "," and depth == 0:
                items.append(cur); cur = ""
            else:
                cur += ch
        items.append(cur)
        for it in items:
            if "=" in it:
                k, v = it.split("=", 1)
                binds[k.strip()] = v.strip()
    kern = re.sub(r"\b[A-Za-z_][A-Za-z_0-9]*\b", lambda m: binds.get(m.group(0), m.group(0)), kern)
    return kern.replace("vtxg::", "").replace("unsigned short", "bf16").replace("bf16_t", "bf16")


def merge_classes(recs):
    """One record per kernel NAME (a templated launch site registers one class per instantiation)."""
    out = {}
    for r in recs:
        k = _display_name(r["name"])
        o = out.setdefault(k, {"name": k, "contraction": not r["name"].startswith("family:"), "launches": 0, "seconds": 0.0,
                               "flops": 0.0, "bytes": 0.0, "cls": []})
        o["launches"] += r["launches"]; o["seconds"] += r["seconds"]; o["flops"] += r["flops"]; o["bytes"] += r["bytes"]
        o["cls"].append(r["cls"])
    return out


def dominant_class(recs):
    """The profile class (= ONE kernel instantiation at one launch site) with the largest summed time: the row a
    `rocprofv3 --kernel-trace --stats` summary of the same step lists first."""
    return max(recs, key=lambda r: r["seconds"])


def _roof_entry(flops, bytes_, seconds, peak_tf):
    tf = flops / seconds / 1e12
    tbs = bytes_ / seconds / 1e12
    return tf, tbs, tf / peak_tf, tbs / PEAK_HBM_TBS


def roofline_violations(roof):
    """Every fraction of a roofline object that exceeds 1: a physically impossible entry means the ACCOUNTING (algorithmic
    FLOPs / bytes of a launch) is wrong, whatever the cause (round 5: the optimizer counted padded chunks and reported 1.14)."""
    bad = []
    def walk(o, path):
        if isinstance(o, dict):
            for k, v in o.items():
                if k in ("frac", "mfma_frac", "hbm_frac") and isinstance(v, (int, float)) and v > 1.0:
                    bad.append(f"{'.'.join(path + [k])} = {v}")
                else:
                    walk(v, path + [str(k)])
    walk(roof, ["roofline"])
    return bad


def step_roofline(recs, dtype, workload_tag, focused=None, survey_steps=1, focused_steps=None):
    """Roofline of the step's dominant kernel, measured LIVE: HIP events attached to every launch of the step function
    the timed region runs (`recs` = ops.profile_stop() of `survey_steps` fully timed steps).  The dominant kernel is the
    single kernel INSTANTIATION (profile class) with the largest summed time -- what a kernel-trace summary of the step lists
    first, flattering or not; achieved = its algorithmic FLOPs (or bytes) / its summed launch time; the binding roof is
    whichever fraction is larger.  `dominant_family` keeps the family-merged pick of rounds 1-5 as a secondary key."""
    if not recs:
        return None
    peak_tf = PEAK_BF16_TFLOPS if dtype == "bf16" else PEAK_F32_TFLOPS
    by_name = merge_classes(recs)
    total = sum(r["seconds"] for r in recs)
    drec = dominant_class(recs)
    share = drec["seconds"] / total
    fam_name = _display_name(drec["name"])
    inst_name = _instantiation_name(drec["name"])
    dom = dict(drec)
    if focused is not None:
        dom = dict(dom, launches=focused["launches"], seconds=focused["seconds"], flops=focused["flops"], bytes=focused["bytes"])
    tf, tbs, f_mfma, f_hbm = _roof_entry(dom["flops"], dom["bytes"], dom["seconds"], peak_tf)
    bound = "mfma" if f_mfma >= f_hbm else "hbm"
    default_workload = workload_tag is not None          # a workload with a PMC table of its own (TABLED_WORKLOADS)
    table, table_source = load_traffic_table(traffic_table_path(workload_tag)) if default_workload else ({}, None)
    steps_timed = focused_steps if (focused is not None and focused_steps) else survey_steps
    per_step = dom["launches"] / steps_timed
    traffic, instantiation = lookup_traffic(table, fam_name, len(by_name[fam_name]["cls"]), per_step)
    if default_workload and table_source and traffic is None:
        print(f"bench.py: ERROR -- {os.path.basename(traffic_table_path(workload_tag))} has no entry for the dominant kernel {inst_name!r}; "
              "reporting traffic = null", file=sys.stderr)
    out = {"bound": bound, "kernel": inst_name, "kernel_in_traffic_table": instantiation or fam_name,
           "achieved": round(tf if bound == "mfma" else tbs * 1e3, 2), "peak": peak_tf if bound == "mfma" else PEAK_HBM_TBS * 1e3,
           "unit": "TFLOP/s" if bound == "mfma" else "GB/s", "frac": round(max(f_mfma, f_hbm), 4),
           "traffic": traffic, "traffic_unit": "bytes/launch",
           "traffic_source": (table_source if traffic is not None else None),
           "launches": dom["launches"], "launches_per_step": round(per_step, 2), "avg_launch_us": round(dom["seconds"] / dom["launches"] * 1e6, 1),
           "flops_per_launch": dom["flops"] / dom["launches"], "algorithmic_bytes": dom["bytes"] / dom["launches"],
           "mfma_frac": round(f_mfma, 4), "hbm_frac": round(f_hbm, 4), "share_of_kernel_time": round(share, 3),
           "selection": "the single kernel instantiation (profile class) with the largest summed time in one fully timed step, side streams off"}
    # the family-merged pick of rounds 1-5 (all instantiations of a launch family summed), survey step's own times
    fdom = max(by_name.values(), key=lambda r: r["seconds"])
    ftf, ftbs, ff_mfma, ff_hbm = _roof_entry(fdom["flops"], fdom["bytes"], fdom["seconds"], peak_tf)
    out["dominant_family"] = {"family": fdom["name"], "instantiations": len(fdom["cls"]), "launches_per_step": fdom["launches"] // survey_steps,
                              "ms_per_step": round(fdom["seconds"] / survey_steps * 1e3, 3), "TFLOP/s": round(ftf, 1), "GB/s": round(ftbs * 1e3, 1),
                              "mfma_frac": round(ff_mfma, 4), "hbm_frac": round(ff_hbm, 4),
                              "bound": "mfma" if ff_mfma >= ff_hbm else "hbm", "frac": round(max(ff_mfma, ff_hbm), 4),
                              "share_of_kernel_time": round(fdom["seconds"] / total, 3)}
    # every HBM-bound kernel, and families
    hbm, fam = {}, {}
    for r in by_name.values():
        ms = r["seconds"] / survey_steps * 1e3
        if not r["contraction"]:
            gbs = r["bytes"] / r["seconds"] / 1e9 if r["seconds"] > 0 else 0.0
            hbm[r["name"]] = {"GB/s": round(gbs, 1), "frac": round(gbs / (PEAK_HBM_TBS * 1e3), 4), "ms_per_step": round(ms, 3),
                              "launches_per_step": r["launches"] // survey_steps}
        f = "contractions (MFMA)" if r["contraction"] else FAMILY_OF.get(r["name"], r["name"])
        o = fam.setdefault(f, {"ms_per_step": 0.0, "bytes": 0.0, "flops": 0.0, "seconds": 0.0})
        o["ms_per_step"] += ms; o["bytes"] += r["bytes"]; o["flops"] += r["flops"]; o["seconds"] += r["seconds"]
    out["hbm_kernels"] = dict(sorted(hbm.items(), key=lambda kv: -kv[1]["ms_per_step"]))
    # every contraction class (one per kernel instantiation): MFMA utilisation next to the HBM fraction
    mk = {}
    for r in by_name.values():
        if (r["contraction"] or r["flops"] > 0) and r["seconds"] > 0:      # incl. the streaming MFMA kernels (expand / stem / 3x3 weight gradient)
            mk[r["name"]] = {"launches_per_step": r["launches"] // survey_steps, "ms_per_step": round(r["seconds"] / survey_steps * 1e3, 3),
                             "TFLOP/s": round(r["flops"] / r["seconds"] / 1e12, 1), "mfma_frac": round(r["flops"] / r["seconds"] / 1e12 / peak_tf, 4),
                             "GB/s": round(r["bytes"] / r["seconds"] / 1e9, 1), "hbm_frac": round(r["bytes"] / r["seconds"] / 1e12 / PEAK_HBM_TBS, 4)}
    out["mfma_kernels"] = dict(sorted(mk.items(), key=lambda kv: -kv[1]["ms_per_step"]))
    # The largest SINGLE contraction instantiation by summed time (round 4's gauge; equal to the headline whenever the step's
    # largest kernel is a contraction).  Times are the survey step's (every launch timed, side streams off).
    if mk:
        dk, dv = max(mk.items(), key=lambda kv: kv[1]["ms_per_step"])
        dtraffic = table.get(dk) if table else None
        out["dominant_contraction"] = {
            "kernel": dk, "launches_per_step": dv["launches_per_step"], "ms_per_step": dv["ms_per_step"],
            "avg_launch_us": round(dv["ms_per_step"] / max(dv["launches_per_step"], 1) * 1e3, 1),
            "TFLOP/s": dv["TFLOP/s"], "GB/s": dv["GB/s"], "mfma_frac": dv["mfma_frac"], "hbm_frac": dv["hbm_frac"],
            "bound": "mfma" if dv["mfma_frac"] >= dv["hbm_frac"] else "hbm", "frac": max(dv["mfma_frac"], dv["hbm_frac"]),
            "algorithmic_bytes": round(by_name[dk]["bytes"] / by_name[dk]["launches"], 1),
            "traffic": dtraffic, "traffic_ratio": (round(dtraffic / (by_name[dk]["bytes"] / by_name[dk]["launches"]), 3) if dtraffic else None)}
    out["families"] = {k: {"ms_per_step": round(v["ms_per_step"], 3), "GB/s": round(v["bytes"] / v["seconds"] / 1e9, 1) if v["seconds"] else 0.0,
                           "TFLOP/s": round(v["flops"] / v["seconds"] / 1e12, 1) if v["seconds"] else 0.0}
                       for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms_per_step"])}
    # step-level model: every launch at its binding roof
    model_s = sum(max(r["flops"] / (peak_tf * 1e12), r["bytes"] / (PEAK_HBM_TBS * 1e12)) for r in by_name.values()) / survey_steps
    out["step_model"] = {"sum_max_mfma_hbm_ms": round(model_s * 1e3, 3), "measured_kernel_ms": round(total / survey_steps * 1e3, 3),
                         "frac": round(model_s / (total / survey_steps), 4),
                         "note": "sum over all launches of one step of max(algorithmic FLOPs / MFMA peak, algorithmic bytes / 8 TB/s) "
                                 "against the sum of their measured durations (side streams off)"}
    out["step_model_frac"] = out["step_model"]["frac"]
    return out


def cpu_baseline(batch, steps, budget_s=40.0):
    """Reference step (oracle port, torch CPU fp32, dropout 0.1) on the host cores.  The box reports
    256 logical CPUs but the container's quota is smaller: more threads than the quota make torch's
    CPU path slower, so the thread count is calibrated first (one step each) and the best one is used
    for the timed sample; `cores` reports what was actually used."""
    from oracle import bicaptioning as port, synth

    torch.manual_seed(0)
    model = port.build_model(dropout=0.1).train()
    step = port.TrainStep(model, start_step=100)
    b = synth.synthetic_batch(batch, seed=0)
    ncpu = os.cpu_count() or 1
    best_t, best_threads = None, 1
    t_start = time.time()
    candidates = [t for t in (16, 32, 64, 128, 256) if t <= ncpu] or [ncpu]      # a host with fewer than 16 CPUs
    for threads in candidates:
        if best_t is not None and time.time() - t_start > budget_s / 2:
            break
        torch.set_num_threads(threads)
        step(b)                                   # warm-up at this thread count
        t0 = time.time()
        step(b)
        dt = time.time() - t0
        if best_t is None or dt < best_t:
            best_t, best_threads = dt, threads
        elif dt > 1.5 * best_t:
            break
    torch.set_num_threads(best_threads)
    step(b)
    n = max(1, min(steps, int((budget_s / 2) / max(best_t, 1e-3))))
    t0 = time.time()
    for _ in range(n):
        step(b)
    dt = time.time() - t0
    out = {"value": round(batch * n / dt, 2), "unit": "images/sec", "cores": best_threads, "kind": "port",
           "sample": f"{n} steps of B={batch} (fp32, dropout 0.1, SGD+Lookahead+clip), oracle/bicaptioning.py; "
                     f"thread count calibrated over {candidates[0]}..{candidates[-1]} threads ({ncpu} logical CPUs)"}
    # BASELINE.json configs[0]: the reference's own CPU-runnable case, bs = 2 (SURVEY.md 8d "Config 1")
    b2 = synth.synthetic_batch(2, seed=1)
    step(b2)
    n2 = 5
    t0 = time.time()
    for _ in range(n2):
        step(b2)
    out["config1_bs2"] = {"value": round(2 * n2 / (time.time() - t0), 2), "unit": "images/sec", "cores": best_threads,
                          "sample": f"{n2} steps of B=2, same model and step"}
    return out


def stock_pytorch_baseline(batch, dev, steps=10, warmup=3, visual="torchvision::resnet50",
                           textual="transdec_postnorm::L1_H1024_A16_F4096", vocab_size=10000):
    """The reference step as the reference itself would run on this GPU: the oracle port (the reference's module graph built
    from torch.nn, oracle/bicaptioning.py) through stock PyTorch-ROCm -- MIOpen convolutions / BatchNorm, hipBLASLt GEMMs, ATen
    attention -- under `torch.autocast(bfloat16)` (the reference's AMP loop, scripts/pretrain_virtex.py:150-161; bf16 needs no
    GradScaler), clip + SGD + Lookahead with the reference's 202 parameter groups; in the reference's own memory format (NCHW)
    and in channels_last, `value` = the faster of the two.  SURVEY.md 8(d) "secondary comparison": a reported baseline like
    `cpu_baseline` (and like it the only other place the oracle is executed), never the measured product.  Off by default
    (--stock-pytorch-baseline): MIOpen's solver search makes the first steps slow."""
    from oracle import bicaptioning as port, synth

    def run(fmt):
        torch.manual_seed(0)
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats(dev)
        model = port.build_model(visual=visual, textual=textual, vocab_size=vocab_size, dropout=0.1).to(dev).to(memory_format=fmt).train()
        inner = model.forward

        def amp_forward(b):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return inner(b)
        model.forward = amp_forward
        step = port.TrainStep(model, start_step=100)
        b = {k: v.to(dev) for k, v in synth.synthetic_batch(batch, seed=0, vocab_size=vocab_size).items()}
        b["image"] = b["image"].contiguous(memory_format=fmt)
        t_w = time.time()
        for _ in range(warmup):
            step(b)
        torch.cuda.synchronize()
        t_w = time.time() - t_w
        t0 = time.time()
        for _ in range(steps):
            loss = step(b)
        torch.cuda.synchronize()
        dt = time.time() - t0
        return {"value": round(batch * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 2), "warmup_s": round(t_w, 1),
                "final_loss": round(float(loss), 4), "peak_memory_gb": round(torch.cuda.max_memory_allocated(dev) / 2**30, 1)}
    layouts = {"nchw (the reference's layout)": torch.contiguous_format, "channels_last": torch.channels_last}
    only = os.environ.get("VTX_STOCK_LAYOUT")                 # (kernel traces of ONE layout: tools/r05_s19.sh)
    runs = {k: run(v) for k, v in layouts.items() if not only or k.startswith(only)}
    best = max(runs, key=lambda k: runs[k]["value"])
    return {"value": runs[best]["value"], "unit": "images/sec", "ms_per_step": runs[best]["ms_per_step"], "layout": best, "runs": runs,
            "what": f"oracle port on cuda through stock PyTorch-ROCm {torch.__version__} (MIOpen / hipBLASLt / ATen), autocast bf16, "
                    f"B={batch}, {steps} steps after {warmup} warm-up steps (MIOpen's solver search is in warmup_s)"}


def set_streams(concurrent: bool):
    """Turn the side streams of the step on / off at run time (virtex_amd/streams.py, models.HEAD_STREAMS)."""
    from virtex_amd import models, streams
    streams.wgrad_stream.enabled = concurrent
    streams.branch_stream.enabled = concurrent
    models.HEAD_STREAMS = concurrent


def main(argv=None, device=None, backend=None):
    """`device` / `backend` are for tests/bench_flow_runner.py only: it drives this exact control flow (single- and
    multi-rank) on CPU tensors after pointing the bindings at the kernel emulator; normal runs leave them None."""
    a = parse(argv)
    injected = device is not None
    if a.serial_streams:
        set_streams(False)
    if a.bn_fusion is not None:
        from virtex_amd.modules import visual_backbones as vbm
        vbm.FUSE_BN_BWD = a.bn_fusion in ("bwd", "both")
        vbm.FUSE_BN_STATS = a.bn_fusion in ("fwd", "both")
    from virtex_amd import distributed as vd
    import virtex_amd.factories as vf
    from virtex_amd.optim import FusedPretrainOptimizer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and a.gpus > 1:
        print(f"bench.py: --gpus {a.gpus} must be launched with torch.distributed.run (one rank per GPU)",
              file=sys.stderr)
        sys.exit(2)
    if injected:
        vd.init_process_group(backend if world > 1 else None)
        dev = device
    else:
        local_rank = vd.init_process_group("nccl" if world > 1 else None)
        if not torch.cuda.is_available():
            raise RuntimeError("bench.py needs an MI355X: the HIP path has no CPU fallback")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)

    def device_sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()

    rank = vd.rank()

    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    model = vf.build_bicaptioning_model(visual=a.visual, textual=a.textual, dropout=a.dropout, vocab_size=a.vocab_size,
                                        compute_dtype=dt).to(dev).train()
    vd.broadcast_parameters(model)
    buckets = vd.GradientBuckets(model)
    buckets.measure_exposed = True      # data_parallel.comm_exposed_ms_per_rank (two events per step; off in production)
    opt = FusedPretrainOptimizer(model, buckets, start_step=100)   # inside warm-up: non-zero LR
    batches = [device_batch(a.batch, dev, 1000 * rank + i, a.image_size, a.vocab_size) for i in range(2)]

    def step(i):
        buckets.zero()
        buckets.begin()
        out = model(batches[i % 2])
        out["loss"].backward()
        scale = buckets.finish()
        opt.step(grad_scale=scale)
        return out["loss"].detach()        # (no autograd graph of an eager step may be alive when a hipGraph capture starts)

    loss = None
    for i in range(a.warmup):
        loss = step(i)
    device_sync()
    # Launch replay (any N: the recorded list carries the bucket all-reduces) / hipGraph (single process).  The issued step is the SAME function;
    # dropout epoch, LR multiplier and Lookahead phase advance on the device (virtex_amd/replay.py, graph.py).
    gstep, launch_mode, launch_fallback = None, "eager", None
    want = a.launch
    if want == "auto":
        # Launch replay is the default only where it has EXECUTED on hardware: the single-process GPU run.  The recorded list can
        # carry the bucket all-reduces (tests: gloo x 2, RCCL x 1), but RCCL x N has never run it and a mismatched collective is
        # a hang, not an exception -- the first multi-GPU record is not bet on it.  Opt in with
        # VIRTEX_AMD_REPLAY_DP=1; the gain at bs 256 is 0.005 ms per step.
        replay_ok = world == 1 or os.environ.get("VIRTEX_AMD_REPLAY_DP", "0") == "1"
        want = "replay" if (dev.type == "cuda" and not a.roofline_live and replay_ok) else "eager"
    if want == "graph" and world > 1:
        want = "eager"               # the hipGraph of the step is single-process; launch replay carries the all-reduces (round 5)
    if want in ("replay", "graph") and not a.roofline_live:
        del loss
        try:
            if want == "graph":
                from virtex_amd.graph import GraphedTrainStep
                gstep = GraphedTrainStep(model, buckets, opt, batches[0], warmup=2)
            else:
                from virtex_amd.replay import StepReplay
                gstep = StepReplay(model, buckets, opt, batches[0], warmup=1, validate=True)
            for i in range(3):
                loss = gstep(batches[i % 2])
            device_sync()
            launch_mode = "hipgraph" if want == "graph" else "replay"
        except Exception as e:                  # an optimisation: the eager step is the product path either way
            if a.launch != "auto":
                raise
            print(f"bench.py: launch {want} failed ({type(e).__name__}: {e}); timing the eager step", file=sys.stderr)
            launch_fallback = f"{want} failed: {type(e).__name__}: {e}"[:400]
            gstep = None
            opt.disable_device_schedule()
            device_sync()
            loss = step(0)
    run_step = (lambda i: gstep(batches[i % 2])) if gstep is not None else step
    vd.synchronize()
    device_sync()
    from virtex_amd import ops
    live = a.roofline_live and not a.no_roofline and rank == 0
    if live:
        ops.profile_start()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = run_step(i)
    host_issue = time.perf_counter() - t0          # the host has enqueued every step (the GPU is still working when it is ahead)
    device_sync()
    vd.synchronize()
    device_sync()
    elapsed = time.perf_counter() - t0
    eager_ms = None
    if gstep is not None:
        gstep.sync()
        # the same step issued by its Python (what N > 1 and any training loop without a fixed batch shape run): timed beside
        # the replayed one, same process, same state -- `value` stays the replayed step, config.eager_ms_per_step says what
        # the difference is
        n_eager = min(a.steps, 20)
        step(0)
        device_sync()
        t1 = time.perf_counter()
        for i in range(n_eager):
            step(i)
        device_sync()
        eager_ms = (time.perf_counter() - t1) / n_eager * 1e3
    live_recs = None
    if live:
        live_recs = ops.profile_stop()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = loss.item()
    seen = vd.ranks_seen(dev if dev.type == "cuda" else None)      # did the transport connect N ranks?  (all-reduce of a one)
    # the part of the gradient exchange the backward pass did not hide (events around finish()'s wait), per rank
    comm_exposed = buckets.comm_exposed_ms(last=a.steps)
    if world > 1:
        ce = torch.tensor([comm_exposed if comm_exposed is not None else -1.0], device=dev, dtype=torch.float64)
        gathered = [torch.zeros_like(ce) for _ in range(world)]
        torch.distributed.all_gather(gathered, ce)
        comm_exposed = [round(g.item(), 3) for g in gathered]
    elif comm_exposed is not None:
        comm_exposed = [round(comm_exposed, 3)]

    # ---- roofline leg.  EVERY rank runs the same further steps (a step contains the gradient all-reduces: a rank
    # stepping alone would wait for its peers forever); only rank 0 attaches events and reports.
    # The roofline is about the kernel, so it is timed WITHOUT co-running kernels: the side streams are switched off
    # for these steps (same kernels, same shapes, same order, one stream).
    #   pass 1 (one step, every contraction launch timed): which kernel class dominates, and the totals;
    #   pass 2 (roofline_steps steps, only that class timed): its per-launch duration;
    #   pass 3 (streams back on, same class): what the same launches take while sharing the chip.
    survey = focused = concurrent = None
    if not a.no_roofline and live_recs is None:
        prof = rank == 0
        set_streams(False)
        step(0)
        if prof:
            ops.profile_start()
        step(0)
        dom_cls = -1
        if prof:
            survey = ops.profile_stop()
            dom_cls = -1
            if survey:          # the single kernel instantiation with the largest summed time; the focused pass times it alone
                dom_cls = dominant_class(survey)["cls"]
            ops.profile_start(only_class=dom_cls)
        for i in range(a.roofline_steps):
            step(i)
        if prof:
            focused = [r for r in ops.profile_stop() if r["cls"] == dom_cls]
        if not a.serial_streams:
            set_streams(True)
            step(0)
            if prof:
                ops.profile_start(only_class=dom_cls)
            for i in range(a.roofline_steps):
                step(i)
            if prof:
                concurrent = [r for r in ops.profile_stop() if r["cls"] == dom_cls]
        if prof:
            ops.profile_start(only_class=-1)
            ops.profile_stop()
        device_sync()

    # ---- fidelity leg (every rank: the engine's finish() is collective; rank 0 reports): the bf16 step against the
    # fp32 step on the batch the timed region used
    # DDP broadcasts rank 0's buffers before EVERY forward (reference: scripts/pretrain_virtex.py:123, broadcast_buffers=True);
    # this engine never does inside the step (BatchNorm running statistics are not read by a training forward), so they are
    # made rank-0's HERE, where they start to matter: before anything validation-like reads them.
    vd.broadcast_buffers(model)
    fid = None
    if a.dtype == "bf16" and not a.no_fidelity:
        from virtex_amd import fidelity
        try:
            fid = fidelity.bf16_vs_fp32(model, batches[0], buckets)
        except Exception as e:      # never lose the bench line to the auxiliary leg
            fid = {"error": f"{type(e).__name__}: {e}"}
        device_sync()

    if rank == 0:
        ips = a.batch * world * a.steps / elapsed
        arch = a.textual.split("::")[1]
        key = "_".join(arch.split("_")[:2])
        cnn = a.visual.split("::")[1]
        gflop = GFLOP_PER_IMG.get(key) if cnn == "resnet50" else (81.35 if (cnn, key) == ("resnet101", "L1_H2048") else None)
        peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
        rec = {
            "metric": "pretrain images/sec", "value": round(ips, 2), "unit": "images/sec", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
            "data": "synthetic" if not injected else "synthetic (injected test device: control-flow test, not a measurement)",
            "config": {"workload": f"bicaptioning_{cnn}_{key} {a.dtype}, bs={a.batch}/GPU, {a.image_size}x{a.image_size} synthetic images + "
                                   "30-tok captions, full step (fwd+bwd+clip+SGD+Lookahead), dropout "
                                   f"{a.dropout}", "global_batch": a.batch * world,
                       "parallelism": f"dp{world}", "final_loss": round(final_loss, 4), "launch": launch_mode,
                       "eager_ms_per_step": (round(eager_ms, 3) if eager_ms is not None else (round(elapsed / a.steps * 1e3, 3) if launch_mode == "eager" else None)),
                       "launch_fallback_reason": launch_fallback,
                       "best_batch": (load_best_batch() if (world == 1 and not injected) else None),
                       # peak of torch's allocator over the whole run: under launch replay this is the validation's peak (two recordings
                       # of a step pinned at once + an eager step), the steady state holds one recording
                       "peak_memory_gb": (round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2) if dev.type == "cuda" else None),
                       "steady_memory_gb": (round(torch.cuda.memory_allocated(dev) / 2 ** 30, 2) if dev.type == "cuda" else None),
                       "host_enqueue_ms_per_step": round(host_issue / a.steps * 1e3, 3),
                       "host_enqueue_note": "wall time of the issuing loop / steps; a host faster than the GPU spends the difference "
                                            "blocked on the full HIP queue, so a value near ms_per_step means 'host not the limit' "
                                            "(un-blocked cost: profiles/r04_launch_replay_configs_2_4_5.txt)"},
        }
        if world > 1 or vd.active():
            rec["data_parallel"] = dict(vd.transport_description(), ranks_seen=seen, launch=launch_mode,
                                        buffers_broadcast_before_validation=True)
        if comm_exposed is not None:
            rec.setdefault("data_parallel", {}).update({"comm_exposed_ms_per_rank": comm_exposed, "payload": buckets.payload,
                                    "buckets_mb": [round((e - s0) * 4 / 2 ** 20, 1) for (s0, e, _) in buckets.buckets],
                                    "note": "GPU time the compute stream waited in GradientBuckets.finish() for the outstanding "
                                            "all-reduces, mean over the timed steps"})
        if gflop:
            tf = ips * gflop / 1e3
            rec["step_mfma"] = {"gflop_per_image": gflop, "achieved_tflops": round(tf, 1),
                                "peak_tflops": peak * world, "frac": round(tf / (peak * world), 4)}
        if not a.no_roofline:
            default_workload = TABLED_WORKLOADS.get((a.batch, a.dtype, a.textual, a.visual, world, a.image_size, a.vocab_size))
            if live_recs is not None:
                rec["roofline"] = step_roofline(live_recs, a.dtype, default_workload)
                if rec["roofline"]:
                    rec["roofline"]["measured"] = "inside the timed region"
            else:
                rec["roofline"] = step_roofline(survey, a.dtype, default_workload, focused[0] if focused else None,
                                                focused_steps=a.roofline_steps)
                if rec["roofline"]:
                    rec["roofline"]["measured"] = (f"{a.roofline_steps} further steps right after the timed region, side streams off, "
                                                   "begin/end HIP events on this kernel class only (class chosen from one fully timed step)")
                    if concurrent:
                        c = concurrent[0]
                        rec["roofline"]["concurrent_avg_launch_us"] = round(c["seconds"] / c["launches"] * 1e6, 1)
            if rec.get("roofline"):
                # the two step-level fractions next to the kernel's, and the consistency check: no
                # fraction of the object may exceed 1
                rec["roofline"]["step_mfma_frac"] = rec.get("step_mfma", {}).get("frac")
                bad = roofline_violations(rec["roofline"])
                rec["roofline"]["consistency"] = "ok: no fraction above 1" if not bad else {"violations": bad}
                if bad:
                    print("bench.py: ERROR -- physically impossible roofline fractions (accounting defect): " + "; ".join(bad), file=sys.stderr)
        if fid is not None:
            rec["fidelity"] = fid
        from virtex_amd.modules import visual_backbones as vbm
        rec["config"]["bn_fusion"] = {"backward_sums_in_dgrad_epilogue": bool(vbm.FUSE_BN_BWD and a.dtype == "bf16"),
                                      "forward_statistics_in_conv_epilogue": bool(vbm.FUSE_BN_STATS and a.dtype == "bf16")}
        if world == 1 and not a.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(a.cpu_batch, a.cpu_steps)
        if world == 1 and a.stock_pytorch_baseline and dev.type == "cuda":
            del model, buckets, opt, batches                     # the product's memory back before the baseline allocates its own
            torch.cuda.empty_cache()
            try:
                rec["stock_pytorch_baseline"] = stock_pytorch_baseline(a.batch, dev, visual=a.visual, textual=a.textual,
                                                                        vocab_size=a.vocab_size)
            except Exception as e:                               # (an out-of-memory or MIOpen failure of the BASELINE is not the product's)
                rec["stock_pytorch_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        print(json.dumps(rec), flush=True)
    vd.synchronize()
    vd.shutdown()


if __name__ == "__main__":
    main()
