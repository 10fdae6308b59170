import faulthandler, sys
import torch
faulthandler.enable()
sys.path.insert(0, ".")
import virtex_amd.factories as vf
from virtex_amd import distributed as vd
from virtex_amd.graph import GraphedTrainStep
from virtex_amd.optim import FusedPretrainOptimizer
from virtex_amd.synthetic import synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mode = sys.argv[2] if len(sys.argv) > 2 else "plain"
dev = torch.device("cuda", 0)
if mode == "initpg":
    vd.init_process_group(None)
    torch.cuda.set_device(0)
torch.manual_seed(0)
model = vf.build_bicaptioning_model(dropout=0.1, compute_dtype=torch.bfloat16).to(dev).train()
if mode == "initpg":
    vd.broadcast_parameters(model)
buckets = vd.GradientBuckets(model)
if mode == "exposed":
    buckets.measure_exposed = True
opt = FusedPretrainOptimizer(model, buckets, start_step=100)
batches = [synthetic_batch(B, dev, image_size=224, max_len=30, vocab_size=10000, seed=i) for i in range(2)]
def step(i):
    buckets.zero(); buckets.begin()
    out = model(batches[i % 2]); out["loss"].backward(); opt.step(grad_scale=buckets.finish()); return out["loss"].detach() if "detached" in mode else out["loss"]
for i in range(0 if "nopre" in mode else 3):
    loss = step(0 if mode == "samebatch" else i)
if mode.startswith("inline"):
    loss = torch.zeros(1)
torch.cuda.synchronize()
if mode == "sync":
    vd.synchronize()
print("eager done", loss.item(), flush=True)
if mode.startswith("inline"):
    opt.enable_device_schedule()
    static = batches[0] if "noclone" in mode else {k: v.clone() for k, v in batches[0].items()}
    def st():
        buckets.zero(); buckets.begin()
        out = model(static); out["loss"].backward(); opt.step(grad_scale=buckets.finish()); return out["loss"] if "nodetach" in mode else out["loss"].detach()
    s_ = torch.cuda.Stream(); s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        for _ in range(2): st()
    torch.cuda.current_stream().wait_stream(s_); torch.cuda.synchronize()
    gg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gg):
        ll = st()
    print("inline captured", flush=True)
    for _ in range(3): gg.replay()
    torch.cuda.synchronize(); print("inline replayed", ll.item(), flush=True); sys.exit(0)
g = GraphedTrainStep(model, buckets, opt, batches[0], warmup=3 if mode == "w3" else 2)
print("captured", flush=True)
for i in range(3):
    loss = g(batches[i % 2])
torch.cuda.synchronize()
print("replayed", loss.item(), flush=True)
