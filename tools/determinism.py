"""Run the same forward+backward twice (same weights, same batch, same dropout seeds) and list the parameters
whose gradients are not bit-identical.  Expected: only the tensors fed by fp32 atomics (the embedding scatter into
the tied word matrix, positions, the embedding LayerNorm).  Anything else would be a stream race."""
import sys, itertools, torch
sys.path.insert(0, ".")
import virtex_amd.factories as vf
from virtex_amd import synthetic, distributed as vd
from virtex_amd.modules import textual_heads as th

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = vf.build_bicaptioning_model(compute_dtype=torch.bfloat16).to(dev).train()
buckets = vd.GradientBuckets(model)
batch = synthetic.synthetic_batch(B, dev)
bn_state = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}

def run():
    th.dropout_seed_state(0)                 # the same dropout seeds in every run
    model.load_state_dict(bn_state, strict=False)
    buckets.zero(); buckets.begin()
    out = model(batch); out["loss"].backward(); buckets.finish()
    torch.cuda.synchronize()
    return out["loss"].item(), buckets.flat.clone()

runs = [run() for _ in range(4)]
print("losses", [r[0] for r in runs])
names = {p: n for n, p in model.named_parameters()}
bad = {}
for r in runs[1:]:
    off = 0
    for p in buckets.params:
        n = p.numel()
        a, b = runs[0][1][off:off + n], r[1][off:off + n]
        if not torch.equal(a, b):
            d = (a - b).abs().max().item() / (a.abs().max().item() + 1e-30)
            bad[names[p]] = max(bad.get(names[p], 0.0), d)
        off += n
for k, v in sorted(bad.items()):
    print(f"  differs: {k:70s} max rel {v:.2e}")
print(f"{len(bad)} of {len(buckets.params)} parameter gradients differ between identical runs")
