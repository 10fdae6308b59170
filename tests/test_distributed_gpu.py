"""Two ranks on ONE MI355X (gloo transport, CUDA tensors): exercises the data-parallel engine's
event/stream choreography, the flat-buffer fused optimizer and the full native step under
world_size 2 on device.  (RCCL refuses two ranks on one GPU; the collective transport itself is
torch.distributed's -- what is ours is everything around it.)"""
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json, torch
sys.path.insert(0, os.environ["VTX_ROOT"])
import torch.distributed as dist
from virtex_amd import distributed as vd
import virtex_amd.factories as vf
from virtex_amd.optim import FusedPretrainOptimizer
from virtex_amd.synthetic import synthetic_batch
vd.init_process_group("gloo")
rank, world = vd.rank(), vd.world_size()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.manual_seed(0)
model = vf.build_bicaptioning_model(textual="transdec_postnorm::L1_H128_A2_F256", vocab_size=1000, dropout=0.0,
                                    compute_dtype=torch.bfloat16).to(dev).train()
vd.broadcast_parameters(model)
buckets = vd.GradientBuckets(model, bucket_mb=8.0)
opt = FusedPretrainOptimizer(model, buckets, total_steps=100, warmup_steps=10, start_step=5)
losses = []
for it in range(3):
    batch = synthetic_batch(4, dev, image_size=64, max_len=12, vocab_size=1000, seed=10 * it + rank)
    buckets.zero(); buckets.begin()
    out = model(batch); out["loss"].backward()
    scale = buckets.finish()
    if it == 0:
        g0 = (buckets.flat * scale).double().norm().item()
    opt.step(grad_scale=scale)
    losses.append(out["loss"].item())
torch.cuda.synchronize()
checksum = sum(p.detach().double().sum().item() for p in model.parameters())
avg = vd.average_across_processes({"loss": out["loss"].detach()})
print("RESULT " + json.dumps({"rank": rank, "world": world, "losses": losses, "checksum": checksum, "gnorm": g0,
                               "nbuckets": len(buckets.buckets), "avg_loss": avg["loss"].item()}), flush=True)
dist.barrier(); dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.gpu
def test_two_ranks_one_gpu_full_step():
    import json
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), VTX_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out[-3000:]
        line = [l for l in out.splitlines() if l.startswith("RESULT ")][-1]
        outs.append(json.loads(line[7:]))
    a, b = sorted(outs, key=lambda o: o["rank"])
    assert a["world"] == 2 and a["nbuckets"] >= 2
    # identical parameters on both ranks after 3 synchronised steps (different data per rank)
    assert a["checksum"] == pytest.approx(b["checksum"], rel=1e-9)
    assert a["gnorm"] == pytest.approx(b["gnorm"], rel=1e-6)
    assert a["losses"] != b["losses"]
    assert a["avg_loss"] == pytest.approx((a["losses"][-1] + b["losses"][-1]) / 2, rel=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("launch", ["eager", "replay"])
@pytest.mark.parametrize("payload", ["fp32", "bf16"])
def test_bench_through_rccl_with_one_rank(payload, launch):
    """The whole N > 1 path of bench.py on the ONE GPU of the test box, through RCCL: `init_process_group("nccl")`
    (communicator creation), `broadcast_parameters`, the bucketed all-reduces on the communication stream fenced by
    events against the compute / weight-gradient / branch streams, the barrier with `device_ids`, the roofline and
    fidelity legs (collective: every rank takes part).  VIRTEX_AMD_FORCE_DIST=nccl keeps the data-parallel engine
    enabled at world size 1, so the first multi-GPU run of the driver does not execute any line for the first time.
    The result must equal the plain single-process run bit for bit in fp32 payload mode (a 1-rank SUM is the identity)."""
    import json
    args = ["--gpus", "1", "--steps", "3", "--warmup", "2", "--batch", "8", "--image-size", "64", "--vocab-size", "1000",
            "--textual", "transdec_postnorm::L1_H128_A2_F256", "--no-cpu-baseline", "--roofline-steps", "1", "--dropout", "0.0",
            "--launch", launch]         # replay (round 5): the recorded launch list carries the RCCL all-reduces and their waits
    def run(extra_env):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                           timeout=900, cwd=ROOT, env=env)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    plain = run({})
    dist_env = {"VIRTEX_AMD_FORCE_DIST": "nccl", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": str(_free_port()), "VIRTEX_AMD_DP_PAYLOAD": payload}
    rccl = run(dist_env)
    assert rccl["n_gpus"] == 1 and rccl["value"] > 0 and "error" not in rccl.get("fidelity", {})
    assert rccl["config"]["launch"] == launch and plain["config"]["launch"] == launch, (rccl["config"], plain["config"])
    if payload == "fp32":       # (the embedding's fp32 atomics make runs differ at 1e-7, nothing more)
        assert abs(rccl["config"]["final_loss"] - plain["config"]["final_loss"]) <= 2e-4
    else:
        assert abs(rccl["config"]["final_loss"] - plain["config"]["final_loss"]) < 2e-3 * abs(plain["config"]["final_loss"])
