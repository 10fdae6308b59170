"""Sustained-load check: ms/step per 100 steps over a long run (does the part throttle under the three-stream step?)."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
import virtex_amd.factories as vf
from virtex_amd import distributed as vd, synthetic
from virtex_amd.optim import FusedPretrainOptimizer

serial = "--serial" in sys.argv
steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1000
if serial:
    bench.set_streams(False)
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = vf.build_bicaptioning_model(compute_dtype=torch.bfloat16).to(dev).train()
buckets = vd.GradientBuckets(model)
opt = FusedPretrainOptimizer(model, buckets, start_step=100)
batches = [synthetic.synthetic_batch(256, dev, seed=i) for i in range(2)]
def step(i):
    buckets.zero(); buckets.begin()
    out = model(batches[i % 2]); out["loss"].backward()
    opt.step(grad_scale=buckets.finish())
for i in range(5): step(i)
torch.cuda.synchronize()
res = []
for blk in range(steps // 100):
    t0 = time.perf_counter()
    for i in range(100): step(i)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) * 10)
print(("serial " if serial else "3-stream ") + " ".join(f"{r:.1f}" for r in res))
