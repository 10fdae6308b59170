"""world_size-2 tests of the data-parallel engine on CPU (gloo): bucketed, hook-driven gradient
all-reduce must equal the average of the per-rank gradients, and the fused scalar averaging must
equal the reference's per-key all_reduce/world (virtex/utils/distributed.py:140-160)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _Toy(torch.nn.Module):
    """Parameter names mimic the real model: a `visual.` part and a text part."""

    def __init__(self):
        super().__init__()
        self.visual = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8),
                                          torch.nn.Conv2d(8, 8, 3, padding=1))
        self.textual = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 4))
        self.unused = torch.nn.Parameter(torch.zeros(3))

    def forward(self, x):
        f = self.visual(x).mean((2, 3))
        return self.textual(f).pow(2).mean()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from virtex_amd import distributed as vd

    vd.init_process_group("gloo")
    torch.manual_seed(0)
    model = _Toy()
    vd.broadcast_parameters(model)
    buckets = vd.GradientBuckets(model, bucket_mb=0.0005)   # tiny buckets -> several collectives
    assert len(buckets.buckets) > 2
    results = []
    # a second engine for another model in the same process must not steal the first one's announcements
    other = _Toy()
    other_buckets = vd.GradientBuckets(other, bucket_mb=0.0005)
    for step in range(2):
        buckets.zero()
        if step == 0:
            buckets.begin()          # optional: finish() re-arms the counters, so step 1 runs without it
        x = torch.randn(4, 3, 6, 6, generator=torch.Generator().manual_seed(100 * step + rank))
        model(x).backward()
        scale = buckets.finish()
        results.append([(n, (p.grad * scale).tolist()) for n, p in model.named_parameters() if p.grad is not None])
    del other_buckets
    # bf16 wire format: same step, gradients rounded to bf16 before the exchange and widened back afterwards
    torch.manual_seed(0)
    model16 = _Toy()
    vd.broadcast_parameters(model16)
    b16 = vd.GradientBuckets(model16, bucket_mb=0.0005, payload="bf16")
    b16.zero()
    x = torch.randn(4, 3, 6, 6, generator=torch.Generator().manual_seed(rank))
    model16(x).backward()
    sc = b16.finish()
    results.append([(n, (p.grad * sc).tolist()) for n, p in model16.named_parameters() if p.grad is not None])
    # the reference helper averages IN PLACE and its callers ignore the return value (pretrain_virtex.py:213)
    d = {"a": torch.tensor(float(rank)), "b": torch.tensor(2.0 * rank + 1)}
    vd.average_across_processes(d)
    bare = torch.tensor([float(rank), 10.0 * rank])
    vd.average_across_processes(bare)
    assert bare.tolist() == [0.5, 5.0]
    # buffers: rank 0's running statistics everywhere, version counters bumped
    bn = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    if bn:
        bn[0].running_mean.fill_(float(rank) + 1.0)
        v0 = bn[0].running_mean._version
        vd.broadcast_buffers(model)
        assert bn[0].running_mean.eq(1.0).all() and bn[0].running_mean._version > v0
    q.put((rank, results, {k: v.item() for k, v in d.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_buckets_match_averaged_gradients():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    outs.sort(key=lambda t: t[0])
    # single-process reference: average of the two ranks' gradients
    torch.manual_seed(0)
    ref = _Toy()
    for step in range(2):
        grads = []
        for r in range(world):
            ref.zero_grad()
            x = torch.randn(4, 3, 6, 6, generator=torch.Generator().manual_seed(100 * step + r))
            ref(x).backward()
            grads.append({n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None})
        for r in range(world):
            got = dict(outs[r][1][step])
            for n in grads[0]:
                exp = (grads[0][n] + grads[1][n]) / 2
                assert torch.allclose(torch.tensor(got[n]), exp, rtol=1e-5, atol=1e-7), (step, r, n)
    # bf16 payload (third result = step 0 again through the bf16 wire): relative L2 error per tensor <= 4e-3
    for r in range(world):
        got = dict(outs[r][1][2])
        ref.zero_grad()
        gs = []
        for rr in range(world):
            ref.zero_grad()
            x = torch.randn(4, 3, 6, 6, generator=torch.Generator().manual_seed(rr))
            ref(x).backward()
            gs.append({n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None})
        differs = 0
        for n in gs[0]:
            exp = (gs[0][n] + gs[1][n]) / 2
            g = torch.tensor(got[n])
            if exp.norm() > 0:
                # two roundings of the addends + one of the bf16 sum, 2^-9 each at worst: <= 4e-3 relative L2 on any
                # tensor large enough to average (tiny bias vectors can sit at the worst case of their largest element)
                assert ((g - exp).norm() / exp.norm()).item() <= (4e-3 if exp.numel() >= 64 else 8e-3), (r, n)
                differs += int(not torch.equal(g, exp))
        assert differs > 0                                           # it really went through bf16
    for r in range(world):
        assert outs[r][2] == {"a": pytest.approx(0.5), "b": pytest.approx(2.0)}


def _deadline_worker(rank, world, port, q):
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), VIRTEX_AMD_COLLECTIVE_TIMEOUT_S="4")
    from virtex_amd import distributed as vd

    vd.init_process_group("gloo")
    assert vd.ranks_seen() == world                      # everybody arrives: the transport connects `world` ranks
    if rank == 1:
        time.sleep(25)                                   # ... and now rank 1 never arrives at the next collective
        q.put((rank, "slept", 25.0))
        return
    t0 = time.time()
    try:
        vd.ranks_seen()
        q.put((rank, "no error", time.time() - t0))
    except Exception as e:                               # gloo raises in the waiting thread (nccl: the watchdog tears the process down)
        q.put((rank, type(e).__name__, time.time() - t0))


def test_a_rank_that_never_arrives_is_an_error_not_a_hang():
    """Every collective of the process group carries the deadline of init_process_group (VIRTEX_AMD_COLLECTIVE_TIMEOUT_S):
    when a peer never reaches a collective, the waiting rank fails after the deadline instead of waiting for ever -- what lets
    a multi-GPU bench run end with an error the driver can read.  `ranks_seen` (an all-reduce of
    ones) is the record's "did the transport connect N ranks" answer."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_deadline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, what, dt = q.get(timeout=120)
        got[r] = (what, dt)
    for p in procs:
        p.join(timeout=60)
    assert got[0][0] != "no error", got
    assert got[0][1] < 20.0, got                         # the 4-s deadline, not rank 1's 25-s nap
