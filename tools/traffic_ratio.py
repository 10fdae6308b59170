"""Algorithmic bytes per launch (bench.py's live survey: roofline.mfma_kernels / hbm_kernels of a bench line) against the HBM bytes
the PMC passes measured for the same kernel class (profiles/traffic_table.json): where a class fetches more than it has to.
    python tools/traffic_ratio.py profiles/r04_bench_default.json [profiles/traffic_table.json]"""
import json
import sys

b = json.load(open(sys.argv[1]))
t = json.load(open(sys.argv[2] if len(sys.argv) > 2 else "profiles/traffic_table.json"))["per_launch_bytes"]
r = b["roofline"]
rows = []
for src in (r["mfma_kernels"], r["hbm_kernels"]):
    for k, v in src.items():
        if any(k == x[1] for x in rows) or not v["launches_per_step"]:
            continue
        us = v["ms_per_step"] * 1e3 / v["launches_per_step"]
        alg = v["GB/s"] * 1e3 * us / 1e6                      # MB per launch
        rows.append((v["ms_per_step"], k, v["launches_per_step"], us, v.get("TFLOP/s", 0.0), v["GB/s"], alg, (t.get(k) or 0.0) / 1e6))
print(f"{'kernel class':104s} {'n':>3s} {'ms/step':>7s} {'us':>6s} {'TF/s':>5s} {'alg MB':>7s} {'PMC MB':>7s} {'ratio':>5s} {'PMC TB/s':>8s}")
for ms, k, n, us, tf, gb, alg, pm in sorted(rows, reverse=True):
    if ms < 0.04:
        continue
    ratio = f"{pm / alg:5.2f}" if (pm and alg) else "    -"
    rate = f"{pm / us:8.2f}" if pm else "       -"
    print(f"{k[:104]:104s} {n:3d} {ms:7.3f} {us:6.1f} {tf:5.0f} {alg:7.1f} {pm:7.1f} {ratio} {rate}")
print("# alg = algorithmic bytes (operands once + epilogue tensors) per launch; PMC = (2 x FETCH_SIZE + WRITE_SIZE) per launch; GEMM classes")
print("# re-read operand panels through L2 / the Infinity Cache, which these counters include: ratio > 1 is waste only for the HBM-bound classes")
