"""Where does hipGraph capture of the step fail?  Captures growing prefixes of the step with faulthandler on."""
import faulthandler
import sys

import torch

faulthandler.enable()
sys.path.insert(0, ".")
import virtex_amd.factories as vf
from virtex_amd import distributed as vd, streams
from virtex_amd.optim import FusedPretrainOptimizer
from virtex_amd.synthetic import synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
stage = sys.argv[2] if len(sys.argv) > 2 else "all"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = vf.build_bicaptioning_model(dropout=0.1, compute_dtype=torch.bfloat16).to(dev).train()
buckets = vd.GradientBuckets(model)
opt = FusedPretrainOptimizer(model, buckets, start_step=100)
batch = synthetic_batch(B, dev, image_size=224, max_len=30, vocab_size=10000, seed=0)
if "--no-streams" in sys.argv:
    streams.wgrad_stream.enabled = False
    streams.branch_stream.enabled = False


def fwd():
    return model(batch)["loss"]


def fwdbwd():
    buckets.zero(); buckets.begin()
    loss = model(batch)["loss"]
    loss.backward()
    return loss


def full():
    loss = fwdbwd()
    opt.step(grad_scale=buckets.finish())
    return loss


fn = {"fwd": fwd, "fwdbwd": fwdbwd, "all": full}[stage]
pre = int(sys.argv[sys.argv.index("--pre") + 1]) if "--pre" in sys.argv else 0
for _ in range(pre):                       # eager steps on the default stream first (what bench.py does)
    fn()
torch.cuda.synchronize()
if stage == "all":
    opt.enable_device_schedule()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        fn()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("warm-up done; capturing", stage, flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = fn()
print("captured", flush=True)
for i in range(3):
    g.replay()
torch.cuda.synchronize()
print("replayed, loss", loss.item(), flush=True)
