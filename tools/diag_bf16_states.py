"""bf16 gradient fidelity vs the fp32 oracle for differently conditioned parameter states."""
import sys

import torch

sys.path.insert(0, "."); sys.path.insert(0, "tests")
from backends import rel_err, select  # noqa: E402
from oracle import bicaptioning as port, synth  # noqa: E402
import virtex_amd.factories as vf  # noqa: E402

dev = select("gpu")
B, S = 16, 224
for state in ("reference_init", "random_bn3x0.2", "random_bn3x1.0"):
    om = synth.seeded_model(port.build_model, seed=0, dropout=0.0, randomize=(state != "reference_init"))
    if state == "random_bn3x0.2":
        with torch.no_grad():
            for n, p in om.named_parameters():
                if n.endswith("bn3.weight"):
                    p.mul_(0.2)
    batch = synth.synthetic_batch(B, image_size=S, seed=3, ragged=True)
    om.train()
    lo = om(batch); lo["loss"].backward()
    for dt in (torch.float32, torch.bfloat16):
        m = vf.build_bicaptioning_model(dropout=0.0, compute_dtype=dt)
        m.load_state_dict(om.state_dict()); m = m.to(dev).train()
        out = m({k: v.to(dev) for k, v in batch.items()}); out["loss"].backward()
        rows = []
        for (n, p), (_, q) in zip(m.named_parameters(), om.named_parameters()):
            a, b = p.grad.cpu().double().flatten(), q.grad.double().flatten()
            if b.norm() == 0:
                continue
            rows.append((rel_err(a, b), (a @ b / (a.norm() * b.norm() + 1e-30)).item(), n))
        cnn = [r for r in rows if "cnn" in r[2]]; txt = [r for r in rows if "cnn" not in r[2]]
        print(f"{state:16s} {str(dt)[6:]:9s} loss {out['loss'].item():.5f} (oracle {lo['loss'].item():.5f}) | "
              f"text: max rel {max(r[0] for r in txt):.2e} min cos {min(r[1] for r in txt):.4f} | "
              f"cnn({len(cnn)}): median rel {sorted(r[0] for r in cnn)[len(cnn)//2]:.2e} max rel {max(r[0] for r in cnn):.2e} "
              f"min cos {min(r[1] for r in cnn):.4f}", flush=True)
