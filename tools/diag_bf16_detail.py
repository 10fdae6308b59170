import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from backends import rel_err, select  # noqa: E402
from oracle import bicaptioning as port, synth  # noqa: E402
import virtex_amd.factories as vf  # noqa: E402
from virtex_amd.modules import visual_backbones as vb  # noqa: E402

dev = select("gpu")
om = synth.seeded_model(port.build_model, seed=0, dropout=0.0, randomize=False)
batch = synth.synthetic_batch(16, image_size=224, seed=3, ragged=True)
grab = {}
orig_b = vb._ResNetFn.backward
def bw(ctx, dfeat):
    grab[ctx.module.compute_dtype].update(dfeat=dfeat.detach().float().clone())
    return orig_b(ctx, dfeat)
vb._ResNetFn.backward = staticmethod(bw)
res = {}
for dt in (torch.float32, torch.bfloat16):
    grab[dt] = {}
    m = vf.build_bicaptioning_model(dropout=0.0, compute_dtype=dt)
    m.load_state_dict(om.state_dict()); m = m.to(dev).train()
    feats = m.visual({k: v.to(dev) for k, v in batch.items()}["image"])
    grab[dt]["feat"] = feats.detach().float().clone()
    m.zero_grad()
    out = m({k: v.to(dev) for k, v in batch.items()}); out["loss"].backward()
    res[dt] = {n: p.grad.detach().float().clone() for n, p in m.named_parameters()}
a, b = grab[torch.bfloat16], grab[torch.float32]
print("feat  bf16 vs fp32:", rel_err(a["feat"], b["feat"]))
print("dfeat bf16 vs fp32:", rel_err(a["dfeat"], b["dfeat"]), " |dfeat| rms", b["dfeat"].pow(2).mean().sqrt().item())
rows = [(rel_err(res[torch.bfloat16][n], res[torch.float32][n]), n) for n in res[torch.float32] if res[torch.float32][n].norm() > 0]
for e, n in rows:
    if "cnn" in n and ("layer4" in n or "layer1.0" in n or n.startswith("visual.cnn.conv1") or n.startswith("visual.cnn.bn1")):
        print(f"{e:.3e} {n}")
