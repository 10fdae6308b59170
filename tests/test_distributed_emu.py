"""world_size-2 on CPU through the kernel emulator with the REAL modules: per-rank forward/backward through the HIP
code path (in-place gradient accumulation, the backbone announcing its blocks to the engine before autograd visits
the parameters, shared parameters accumulated from several backward nodes), gloo all-reduce, then every gradient
against the oracle's average of the two ranks' gradients."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from backends import rel_err
from oracle import bicaptioning as port, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(textual="transdec_postnorm::L1_H128_A2_F256", vocab_size=304)
BK = dict(batch_size=2, image_size=64, max_len=8, vocab_size=304, ragged=True)

WORKER = r'''
import os, sys, json, torch
root = os.environ["VTX_ROOT"]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch.distributed as dist
from backends import select
from oracle import bicaptioning as port, synth
from virtex_amd import distributed as vd
import virtex_amd.factories as vf
dev = select("emu")
vd.init_process_group("gloo")
rank = vd.rank()
kw, bk = json.loads(os.environ["VTX_KW"]), json.loads(os.environ["VTX_BK"])
oracle_model = synth.seeded_model(port.build_model, seed=0, dropout=0.0, **kw)
model = vf.build_bicaptioning_model(dropout=0.0, compute_dtype=torch.float32, **kw)
model.load_state_dict(oracle_model.state_dict())
model = model.to(dev).train()
vd.broadcast_parameters(model)
buckets = vd.GradientBuckets(model, bucket_mb=2.0, payload=os.environ["VTX_PAYLOAD"])
assert buckets.buckets[-1][1] - buckets.buckets[-1][0] <= 2.0 * (1 << 20) / 4 / 2 + 1      # the tail bucket (closes last, exposed) is small
batch = synth.synthetic_batch(seed=40 + rank, **bk)
# order of host-side events of the backward pass: every bucket launch, every convolution weight-gradient enqueue
from virtex_amd import ops as _ops
order = []
_launch0, _wgrad0 = buckets._launch, _ops.conv2d_wgrad
def _launch(b):
    order.append(["bucket", b]); return _launch0(b)
def _wgrad(x, dy, dw, *a, **k):
    order.append(["wgrad", list(dw.shape)]); return _wgrad0(x, dy, dw, *a, **k)
buckets._launch = _launch; _ops.conv2d_wgrad = _wgrad
buckets.zero(); buckets.begin()
model({k: v.to(dev) for k, v in batch.items()})["loss"].backward()
scale = buckets.finish()
early = buckets.last_early
out = {n: (p.grad * scale).flatten()[:: max(1, p.numel() // 64)][:64].tolist() + [float((p.grad * scale).double().norm())]
       for n, p in model.named_parameters()}
with open(os.environ["VTX_OUT"] + f".{rank}", "w") as f:      # a file, not the pipe: the parent reads the ranks one after the other
    json.dump({"rank": rank, "nbuckets": len(buckets.buckets), "early": early, "grads": out, "order": order}, f)
dist.barrier(); dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.emu
@pytest.mark.parametrize("payload", ["fp32", "bf16"])
def test_two_ranks_real_modules_average_matches_oracle(tmp_path, payload):
    """payload = bf16: every bucket is rounded to bf16 before the exchange, summed in bf16 and widened back (half the
    xGMI bytes): per-tensor error of the averaged gradient <= 4e-3 on top of the fp32 path's."""
    port_no = _free_port()
    out_base = str(tmp_path / "rank")
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port_no), RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r),
                   VTX_ROOT=ROOT, VTX_KW=json.dumps(KW), VTX_BK=json.dumps(BK), VTX_OUT=out_base, OMP_NUM_THREADS="2",
                   VTX_PAYLOAD=payload)
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
    outs = []
    for r, p in enumerate(procs):
        _, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-3000:]
        with open(f"{out_base}.{r}") as f:
            outs.append(json.load(f))
    outs.sort(key=lambda o: o["rank"])
    assert outs[0]["nbuckets"] > 2
    assert outs[0]["early"] >= 150           # the backbone's 161 tensors were announced from inside its backward
    # Overlap by construction: every bucket except the small tail one has been handed to the process group BEFORE the host
    # enqueues the stem's weight gradient (the last kernel of the backward pass) -- i.e. while the early stages' backward is
    # still being enqueued, not at the end; the tail bucket (which holds the stem) necessarily follows it.
    for o in outs:
        seq = o["order"]
        stem = max(i for i, e in enumerate(seq) if e[0] == "wgrad" and e[1][1] == 7)          # the 7x7 stem filter (packed 7 x 8 x 4)
        launched_before = {e[1] for e in seq[:stem] if e[0] == "bucket"}
        assert launched_before == set(range(o["nbuckets"] - 1)), (sorted(launched_before), o["nbuckets"])
        assert [e for e in seq[stem:] if e[0] == "bucket"] == [["bucket", o["nbuckets"] - 1]]
    # oracle: the same seeded state, each rank's batch, gradients averaged
    grads = []
    for r in range(2):
        m = synth.seeded_model(port.build_model, seed=0, dropout=0.0, **KW).train()
        m(synth.synthetic_batch(seed=40 + r, **BK))["loss"].backward()
        grads.append({n: p.grad.clone() for n, p in m.named_parameters()})
    worst_text, cnn_err = 0.0, []
    for n in grads[0]:
        exp = (grads[0][n] + grads[1][n]) / 2
        samp = exp.flatten()[:: max(1, exp.numel() // 64)][:64]
        for r in range(2):
            got = outs[r]["grads"][n]
            assert outs[0]["grads"][n] == got                       # both ranks hold the same reduced gradient
            e = rel_err(torch.tensor(got[:-1]), samp) if samp.abs().max() > 0 else 0.0
            ne = abs(got[-1] - exp.double().norm().item()) / (exp.double().norm().item() + 1e-30)
            if "cnn" in n:
                cnn_err.append(ne)
            else:
                worst_text = max(worst_text, e, ne)
    assert worst_text < (1e-3 if payload == "fp32" else 5e-3), worst_text
    cnn_err.sort()
    assert cnn_err[len(cnn_err) // 2] < 3e-2 and cnn_err[-1] < 0.3, (cnn_err[len(cnn_err) // 2], cnn_err[-1])   # BN conditioning


REPLAY_WORKER = r'''
import os, sys, json, torch
root = os.environ["VTX_ROOT"]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch.distributed as dist
from backends import select
from oracle import synth
from virtex_amd import distributed as vd
import virtex_amd.factories as vf
from virtex_amd.optim import FusedPretrainOptimizer
from virtex_amd.replay import StepReplay
dev = select("emu")
vd.init_process_group("gloo")
rank = vd.rank()
kw, bk = json.loads(os.environ["VTX_KW"]), json.loads(os.environ["VTX_BK"])
torch.manual_seed(0)
model = vf.build_bicaptioning_model(dropout=0.0, compute_dtype=torch.float32, **kw).to(dev).train()
vd.broadcast_parameters(model)
buckets = vd.GradientBuckets(model, bucket_mb=2.0, payload=os.environ["VTX_PAYLOAD"])
opt = FusedPretrainOptimizer(model, buckets, total_steps=50, warmup_steps=6, start_step=2, lookahead_k=3)
batch = lambda s: {k: v.to(dev) for k, v in synth.synthetic_batch(seed=s, **bk).items()}
if os.environ.get("VTX_SABOTAGE_RANK") == str(rank):
    # this rank's replayed step "differs" from its eager step: both ranks must give up the recording TOGETHER
    from virtex_amd import replay as _rp
    _orig_run = _rp._run
    def _bad_run(rec):
        _orig_run(rec)
        buckets.flat.add_(1.0)
    _rp._run = _bad_run
if "VTX_SABOTAGE_RANK" in os.environ:
    def eager():
        buckets.zero(); buckets.begin()
        out = model(batch(90 + rank)); out["loss"].backward()
        opt.step(grad_scale=buckets.finish())
        return out["loss"].item()
    msg = None
    try:
        StepReplay(model, buckets, opt, batch(70 + rank), warmup=1, validate=True)
    except RuntimeError as e:
        msg = str(e)
    opt.disable_device_schedule()
    loss = eager()                      # the fallback of bench.py: an eager step on EVERY rank; a lone rank here would hang
    samp = {n: p.detach().flatten()[:: max(1, p.numel() // 32)][:32].tolist() for n, p in model.named_parameters()}
    with open(os.environ["VTX_OUT"] + f".{rank}", "w") as f:
        json.dump({"rank": rank, "error": msg, "loss": loss, "params": samp}, f)
    dist.barrier(); dist.destroy_process_group()
    sys.exit(0)
try:
    # construction records one step INCLUDING the bucket all-reduces and validates the recording against an eager step
    # (both under the world-2 exchange) on a second batch; then three replayed steps on per-rank batches
    replay = StepReplay(model, buckets, opt, batch(70 + rank), warmup=1, validate=True)
    counts = dict(replay.rec.counts)
    losses = [replay(batch(80 + 10 * i + rank)).item() for i in range(3)]
    replay.sync()
finally:
    opt.disable_device_schedule()
samp = {n: p.detach().flatten()[:: max(1, p.numel() // 32)][:32].tolist() for n, p in model.named_parameters()}
with open(os.environ["VTX_OUT"] + f".{rank}", "w") as f:
    json.dump({"rank": rank, "nbuckets": len(buckets.buckets), "counts": counts, "losses": losses, "params": samp, "step": opt.step_idx}, f)
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.emu
@pytest.mark.parametrize("payload", ["fp32", "bf16"])
def test_two_ranks_launch_replay_carries_the_gradient_exchange(tmp_path, payload):
    """N > 1 runs the SAME step the 1-GPU headline times: the recorded launch list contains every bucket all-reduce (and the wait
    on it); the recording validates against the eager step on both ranks, and after replayed steps on DIFFERENT per-rank
    batches both ranks hold identical parameters (the averaged gradient reached both optimizers)."""
    port_no = _free_port()
    out_base = str(tmp_path / "rank")
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port_no), RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r),
                   VTX_ROOT=ROOT, VTX_KW=json.dumps(KW), VTX_BK=json.dumps(BK), VTX_OUT=out_base, OMP_NUM_THREADS="2",
                   VTX_PAYLOAD=payload)
        procs.append(subprocess.Popen([sys.executable, "-c", REPLAY_WORKER], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
    outs = []
    for r, p in enumerate(procs):
        _, se = p.communicate(timeout=900)
        assert p.returncode == 0, se[-3000:]
        with open(f"{out_base}.{r}") as f:
            outs.append(json.load(f))
    outs.sort(key=lambda o: o["rank"])
    for o in outs:
        assert o["counts"]["collective"] == 2 * o["nbuckets"], o["counts"]        # every bucket: the all-reduce and its wait
        assert o["step"] == outs[0]["step"]
    assert outs[0]["losses"] != outs[1]["losses"]                                 # different batches per rank ...
    for n, v in outs[0]["params"].items():                                        # ... the same parameters after the exchange
        assert v == outs[1]["params"][n], n


@pytest.mark.emu
def test_two_ranks_give_up_a_recording_together(tmp_path):
    """A recording that fails its validation on ONE rank is refused on every rank (the verdict is reduced over the ranks): the
    ranks then take the eager step together -- a rank deciding alone would issue a different number of gradient exchanges than
    its peer and the job would hang (bench.py's fallback at N > 1)."""
    port_no = _free_port()
    out_base = str(tmp_path / "rank")
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port_no), RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r),
                   VTX_ROOT=ROOT, VTX_KW=json.dumps(KW), VTX_BK=json.dumps(BK), VTX_OUT=out_base, OMP_NUM_THREADS="2",
                   VTX_PAYLOAD="fp32", VTX_SABOTAGE_RANK="1")
        procs.append(subprocess.Popen([sys.executable, "-c", REPLAY_WORKER], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
    outs = []
    for r, p in enumerate(procs):
        _, se = p.communicate(timeout=900)
        assert p.returncode == 0, se[-3000:]
        with open(f"{out_base}.{r}") as f:
            outs.append(json.load(f))
    outs.sort(key=lambda o: o["rank"])
    assert "on another rank" in outs[0]["error"]
    assert "gradients differ" in outs[1]["error"]
    for n, v in outs[0]["params"].items():                                        # the eager step after the refusal exchanged gradients
        assert v == outs[1]["params"][n], n
