"""bf16-vs-fp32-oracle accuracy diagnostics of the whole step on a real MI355X."""
import sys

import torch

sys.path.insert(0, "."); sys.path.insert(0, "tests")
from backends import rel_err, select  # noqa: E402
from oracle import bicaptioning as port, synth  # noqa: E402
import virtex_amd.factories as vf  # noqa: E402

dev = select("gpu")
for (B, S) in [(2, 224), (16, 224)]:
    mkw = dict(textual="transdec_postnorm::L1_H1024_A16_F4096", vocab_size=10000)
    om = synth.seeded_model(port.build_model, seed=0, dropout=0.0, **mkw)
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
            rows.append((rel_err(a, b), (a @ b / (a.norm() * b.norm() + 1e-30)).item(), n))
        cnn = [r for r in rows if "cnn" in r[2]]; txt = [r for r in rows if "cnn" not in r[2]]
        print(f"B={B} {str(dt)[6:]:9s} loss {out['loss'].item():.5f} (oracle {lo['loss'].item():.5f}) | "
              f"text grads: max rel {max(r[0] for r in txt):.2e} min cos {min(r[1] for r in txt):.4f} | "
              f"cnn grads: median rel {sorted(r[0] for r in cnn)[len(cnn)//2]:.2e} max rel {max(r[0] for r in cnn):.2e} "
              f"min cos {min(r[1] for r in cnn):.4f}", flush=True)
        worst = sorted(txt)[-3:]
        print("     worst text:", [(f"{r[0]:.2e}", r[2]) for r in worst])
