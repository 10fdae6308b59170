"""Diagnostic: whole-model gradient agreement vs the CPU oracle at several batch/image sizes,
next to the oracle's own fp32-vs-fp64 disagreement (conditioning of the problem)."""
import copy
import sys

import torch

sys.path.insert(0, "."); sys.path.insert(0, "tests")
from backends import rel_err, select  # noqa: E402
from oracle import bicaptioning as port, synth  # noqa: E402
import virtex_amd.factories as vf  # noqa: E402

backend = sys.argv[1] if len(sys.argv) > 1 else "gpu"
dev = select(backend)
configs = [(2, 224), (8, 128), (16, 224)] if backend == "gpu" else [(3, 64)]
for (B, S) in configs:
    mkw = dict(textual="transdec_postnorm::L1_H1024_A16_F4096", vocab_size=10000)
    if backend != "gpu":
        mkw = dict(textual="transdec_postnorm::L2_H128_A2_F256", vocab_size=1000)
    om = synth.seeded_model(port.build_model, seed=0, dropout=0.0, **mkw)
    od = copy.deepcopy(om).double()
    m = vf.build_bicaptioning_model(textual=mkw["textual"], vocab_size=mkw["vocab_size"], dropout=0.0,
                                    compute_dtype=torch.float32)
    m.load_state_dict(om.state_dict()); m = m.to(dev)
    batch = synth.synthetic_batch(B, image_size=S, vocab_size=mkw["vocab_size"], seed=3, ragged=True)
    bd = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in batch.items()}
    om.train(); od.train(); m.train()
    lo = om(batch)["loss"]; lo.backward()
    ld = od(bd)["loss"]; ld.backward()
    lm = m({k: v.to(dev) for k, v in batch.items()})["loss"]; lm.backward()
    rows = []
    for (n, p), (_, q), (_, r) in zip(m.named_parameters(), om.named_parameters(), od.named_parameters()):
        rows.append((rel_err(p.grad.cpu(), r.grad), rel_err(q.grad, r.grad), n))
    mine = sorted(r[0] for r in rows); ref = sorted(r[1] for r in rows)
    cnn = [r for r in rows if "cnn" in r[2]]; txt = [r for r in rows if "cnn" not in r[2]]
    print(f"B={B} S={S} loss mine {lm.item():.7f} o32 {lo.item():.7f} o64 {ld.item():.7f}")
    print(f"   grad rel-err vs fp64 oracle:  mine median {mine[len(mine)//2]:.2e} max {mine[-1]:.2e} | oracle-fp32 median {ref[len(ref)//2]:.2e} max {ref[-1]:.2e}")
    print(f"   cnn: mine max {max(r[0] for r in cnn):.2e} ref max {max(r[1] for r in cnn):.2e} | text: mine max {max(r[0] for r in txt):.2e} ref max {max(r[1] for r in txt):.2e}", flush=True)
