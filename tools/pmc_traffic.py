"""Merge the FETCH_SIZE and WRITE_SIZE dumps of tools/pmc_dump.py (two separate rocprofv3 --pmc passes of the same command)
into HBM bytes per launch:  bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 -- on gfx950 FETCH_SIZE reports half the bytes
of a wide coalesced stream (MI355X_MICROARCH.md, HBM section; calibrated on the BatchNorm apply kernels, whose byte counts
are exact).   python tools/pmc_traffic.py fetch.txt write.txt [--json profiles/traffic_table.json] > profiles/rNN_pmc_traffic.txt
--json: the table bench.py reads for roofline.traffic -- bytes per launch per kernel class (contraction instantiations by the
name bench.py derives from the library's profile classes, every other kernel by its launch family), stamped with the sha256
of the kernel sources it was measured on (virtex_amd.build.csrc_hash): bench.py reports traffic = null when the sources have
changed since."""
import json
import os
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("vtxg::", "").replace("unsigned short", "bf16")
    return re.sub(r"\(.*\)$", "", name).strip()


# launch family (VTX_KLAUNCH name in virtex_amd/csrc) of the kernels that are not contraction instantiations
FAMILIES = [("bn_bwd_apply_fused_kernel", "bn_bwd_apply"), ("pool_bn_bwd_apply_kernel", "bn_bwd_apply"), ("bn_bwd_apply_kernel", "bn_bwd_apply"),
            ("bn_relu_maxpool_fwd_kernel", "bn_fwd_apply"), ("bn_apply_kernel", "bn_fwd_apply"), ("pool_bn_bwd_reduce_kernel", "bn_bwd_reduce"),
            ("bn_reduce_kernel", "bn_reduce"), ("ln_fwd_kernel", "layernorm_fwd"), ("ln_bwd_kernel", "layernorm_bwd"),
            ("embed_fwd_kernel", "embedding_fwd"), ("embed_bwd_kernel", "embedding_bwd"), ("sgd_lookahead_kernel", "optimizer_step"),
            ("expand1x1_fwd_kernel", "expand1x1_fwd"), ("conv3_bwd_fused_kernel", "conv3_bwd_fused"), ("stem_stream_fwd_kernel", "stem_stream_fwd"),
            ("conv3x3_wgrad_stream_kernel", "conv3x3_wgrad_stream"), ("splitk_reduce_kernel", "splitk_reduce"),
            ("maxpool_fwd_kernel", "maxpool_fwd"), ("maxpool_bwd_kernel", "maxpool_bwd"), ("weight_prep_batched_kernel", "weight_prep")]


def class_name(kernel):
    """the name bench.py uses for the class: contraction instantiations without `void `, the trailing LEAN flag and spaces
    normalised; other kernels by launch family"""
    k = kernel[5:] if kernel.startswith("void ") else kernel
    if k.startswith("contraction_") or k.startswith("conv3x3_shared_kernel"):
        k = re.sub(r"(, (true|false))+>$", ">", k)            # trailing compile-time flags (LEAN, persistent form)
        return re.sub(r"\s*>\s*$", ">", k).replace(" >", ">")
    if k.startswith("bn_reduce_kernel"):     # one kernel, two launch families: <T, BWD = true> is the stand-alone BACKWARD reduction
        return "bn_bwd_reduce" if re.search(r",\s*true\s*>", k) else "bn_fwd_reduce"
    for prefix, fam in FAMILIES:
        if k.startswith(prefix):
            return fam
    return k


def parse(path, counter):
    out, name = {}, None
    for line in open(path):
        line = line.rstrip("\n")
        m = re.match(r"\s+%s\s+n=\s*(\d+)\s+avg=([0-9.e+]+)" % counter, line)
        if m and name:
            out[name] = (int(m.group(1)), float(m.group(2)))
        elif line and not line.startswith(" ") and not line.startswith("["):
            name = short(line)
    return out


args = [a for a in sys.argv[1:] if not a.startswith("--")]
json_out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
if json_out in args:
    args.remove(json_out)
fetch, write = parse(args[0], "FETCH_SIZE"), parse(args[1], "WRITE_SIZE")
rows = []
for k, (n, f) in fetch.items():
    if k in write:
        w = write[k][1]
        rows.append(((2 * f + w) * 1024 * n, n, f, w, k))
for total, n, f, w, k in sorted(rows, reverse=True):
    if (2 * f + w) * 1024 < 5e6:
        continue
    print(f"n={n:4d}  FETCH_SIZE {f:10.1f} KiB  WRITE_SIZE {w:10.1f} KiB  -> {(2 * f + w) * 1024 / 1e6:8.1f} MB/launch   {k}")

if json_out:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from virtex_amd.build import csrc_hash
    agg, per_kernel = {}, {}
    steps = max([n for total, n, f, w, k in rows if short(k).startswith("sgd_lookahead_kernel")] or [1])     # one optimizer launch per step
    for total, n, f, w, k in rows:
        c = class_name(k)
        a = agg.setdefault(c, [0.0, 0])
        a[0] += total; a[1] += n
        per_kernel.setdefault(c, {})[k] = {"launches_per_step": round(n / steps, 2), "bytes": round(total / n, 1)}
    table = {"csrc_sha256": csrc_hash(),
             "rule": "bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024, separate rocprofv3 --pmc passes of `python bench.py` (MI355X_MICROARCH.md, HBM)",
             "source": os.path.basename(args[0]) + " + " + os.path.basename(args[1]),
             "per_launch_bytes": {c: round(t / n, 1) for c, (t, n) in sorted(agg.items()) if n > 0},
             "launches_profiled": {c: n for c, (t, n) in sorted(agg.items())},
             # a launch family of several instantiations (bn_bwd_apply = the flat kernel at two unrolls + the pooled one): bench.py's
             # focused pass times ONE of them and looks its bytes up here by launches per step
             "steps_profiled": steps,
             "per_kernel": {c: v for c, v in sorted(per_kernel.items()) if len(v) > 1}}
    with open(json_out, "w") as fh:
        json.dump(table, fh, indent=1, sort_keys=True)
        fh.write("\n")
