"""profiles/best_batch.json from bench lines of the same build at several per-GPU batch sizes (tools/round_end.sh):
    python tools/best_batch.py gpurun_out/bench_default.json gpurun_out/bench_b320.json ... > profiles/best_batch.json
The entry with the highest images/sec wins; the file carries the hash of the kernel sources it was measured on (bench.py reports
`config.best_batch` only while it matches)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virtex_amd.build import csrc_hash  # noqa: E402

rows = []
for path in sys.argv[1:]:
    try:
        rec = json.loads([l for l in open(path) if l.startswith("{")][-1])
    except (OSError, IndexError, ValueError):
        continue
    batch = rec["config"]["global_batch"] // rec["n_gpus"]
    rows.append({"batch": batch, "value": rec["value"], "ms_per_step": rec["ms_per_step"], "source": os.path.basename(path),
                 "final_loss": rec["config"].get("final_loss")})
if not rows:
    sys.exit("no bench lines")
best = max(rows, key=lambda r: r["value"])
print(json.dumps(dict(best, others=[{k: r[k] for k in ("batch", "value", "ms_per_step")} for r in sorted(rows, key=lambda r: r["batch"])],
                      csrc_sha256=csrc_hash()), indent=1))
