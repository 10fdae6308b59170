"""The fused HIP optimizer tail (clip + SGD + Lookahead + warm-up/cosine schedule over flat buffers)
against the oracle's restatement of the reference chain (oracle.bicaptioning.TrainStep, itself checked
against the reference's own Lookahead / scheduler in tests/test_oracle.py)."""
import copy

import pytest
import torch

from backends import BACKENDS, select
from oracle import bicaptioning as port
from virtex_amd import distributed as vd
from virtex_amd.optim import FusedPretrainOptimizer, PretrainOptimizer


class _Toy(torch.nn.Module):
    """Names exercise every branch of the reference grouping: cnn LR, NO_DECAY regex, 4-D weights."""

    def __init__(self):
        super().__init__()
        self.visual = torch.nn.Module()
        self.visual.cnn = torch.nn.Sequential(torch.nn.Conv2d(8, 16, 3, bias=False), torch.nn.BatchNorm2d(16))
        self.textual = torch.nn.Module()
        self.textual.embedding = torch.nn.Module()
        self.textual.embedding.layer_norm = torch.nn.LayerNorm(24)
        self.textual.transformer = torch.nn.Linear(24, 5000)      # spans several 4096-element chunks
        self.textual.output = torch.nn.Linear(24, 7)


def _grads(model, seed):
    g = torch.Generator().manual_seed(seed)
    return [3.0 * torch.randn(p.shape, generator=g) for p in model.parameters()]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("fused", [True, False])
def test_optimizer_matches_reference_chain(backend, fused):
    if backend == "emu" and not fused:
        pytest.skip("torch foreach optimizer needs no emulator")
    dev = select(backend)
    torch.manual_seed(0)
    ref_model = _Toy()
    ref_model.visual.cnn[0].weight.data = ref_model.visual.cnn[0].weight.data.contiguous(memory_format=torch.channels_last)
    model = copy.deepcopy(ref_model).to(dev)
    ref = port.TrainStep(ref_model, total_steps=50, warmup_steps=10, start_step=3)
    buckets = vd.GradientBuckets(model, bucket_mb=0.01)
    cls = FusedPretrainOptimizer if fused else PretrainOptimizer
    opt = cls(model, buckets, total_steps=50, warmup_steps=10, start_step=3) if fused else \
        cls(model, total_steps=50, warmup_steps=10, start_step=3)
    ref_params = list(ref_model.parameters())
    for it in range(12):                                        # crosses two Lookahead syncs, warm-up end
        gs = _grads(ref_model, it)
        # oracle: the body of TrainStep.__call__ after backward
        for p, g in zip(ref_params, gs):
            p.grad = g.clone()
        torch.nn.utils.clip_grad_norm_(ref_params, ref.clip)
        ref.opt.step()
        ref.kc += 1
        if ref.kc >= ref.k:
            ref.kc = 0
            with torch.no_grad():
                for grp, slow in zip(ref.opt.param_groups, ref.slow):
                    q = grp["params"][0]
                    q.mul_(ref.alpha).add_(slow, alpha=1.0 - ref.alpha)
                    slow.copy_(q)
        ref.step_idx += 1
        ref._set_lr()
        # native
        buckets.zero()
        for p, g in zip(model.parameters(), gs):
            p.grad.add_(g.to(dev) * 2.0)                        # pretend a 2-rank SUM: scale 1/2 below
        opt.step(grad_scale=0.5)
    for (n, p), q in zip(model.named_parameters(), ref_params):
        assert torch.allclose(p.detach().cpu(), q.detach(), rtol=2e-5, atol=1e-6), n


def _steps(model, opt, buckets, dev, its):
    for it in its:
        buckets.zero()
        for p, g in zip(model.parameters(), _grads(model, it)):
            p.grad.add_(g.to(dev))
        opt.step()


@pytest.mark.parametrize("backend", BACKENDS)
def test_checkpoint_resume_and_reference_interchange(backend, tmp_path):
    """state_dict() has torch.optim.SGD's layout -- what the reference's Lookahead.state_dict() returns
    (lookahead.py:74-86) and CheckpointManager writes (checkpointing.py:112-123).  A run resumed from a
    checkpoint taken at a Lookahead boundary continues bit-for-bit; the same file loads into a stock
    torch.optim.SGD built with the reference's grouping, and into the unfused optimizer."""
    dev = select(backend)
    torch.manual_seed(1)
    model_a = _Toy().to(dev)
    buckets_a = vd.GradientBuckets(model_a, bucket_mb=0.01)
    opt_a = FusedPretrainOptimizer(model_a, buckets_a, total_steps=40, warmup_steps=4)
    _steps(model_a, opt_a, buckets_a, dev, range(5))                  # k = 5: slow == fast right now
    ckpt = tmp_path / "checkpoint_5.pth"
    torch.save({"model": model_a.state_dict(), "optimizer": opt_a.state_dict(), "iteration": 5}, ckpt)

    loaded = torch.load(ckpt, map_location="cpu")
    torch.manual_seed(2)
    model_b = _Toy().to(dev)
    model_b.load_state_dict(loaded["model"])
    buckets_b = vd.GradientBuckets(model_b, bucket_mb=0.01)
    opt_b = FusedPretrainOptimizer(model_b, buckets_b, total_steps=40, warmup_steps=4)
    opt_b.load_state_dict(loaded["optimizer"])
    assert opt_b.step_idx == 5 and opt_b.kc == 0
    _steps(model_a, opt_a, buckets_a, dev, range(5, 12))
    _steps(model_b, opt_b, buckets_b, dev, range(5, 12))
    for (n, p), (_, q) in zip(model_a.named_parameters(), model_b.named_parameters()):
        assert torch.equal(p, q), n

    # the reference side: a stock SGD over the reference's one-group-per-tensor layout accepts the file
    model_c = _Toy()
    model_c.load_state_dict(loaded["model"])
    sgd = torch.optim.SGD(port.param_groups(model_c.named_parameters()), momentum=0.9)
    sgd.load_state_dict({k: v for k, v in loaded["optimizer"].items() if k in ("state", "param_groups")})
    for i, p in enumerate(model_c.parameters()):
        assert torch.equal(sgd.state[p]["momentum_buffer"], loaded["optimizer"]["state"][i]["momentum_buffer"])
    assert [g["weight_decay"] for g in sgd.param_groups] == [g["weight_decay"] for g in loaded["optimizer"]["param_groups"]]
    # ... and so does the unfused optimizer; its own state dict round-trips through the fused one
    opt_c = PretrainOptimizer(model_c, total_steps=40, warmup_steps=4)
    opt_c.load_state_dict(loaded["optimizer"])
    assert opt_c.step_idx == 5
    sd_c = opt_c.state_dict()
    assert set(sd_c["state"]) == set(loaded["optimizer"]["state"])
    opt_b.load_state_dict(sd_c)

    # slow-weight evaluation helpers (lookahead.py:104-133)
    before = [p.detach().clone() for p in model_a.parameters()]
    opt_a.load_slow_weights()
    assert any(not torch.equal(p, b) for p, b in zip(model_a.parameters(), before))     # 2 steps past the last sync
    opt_a.restore_fast_weights()
    for p, b in zip(model_a.parameters(), before):
        assert torch.equal(p, b)


@pytest.mark.parametrize("backend", BACKENDS)
def test_state_dict_is_in_named_parameter_order(backend):
    """The reference builds one param group per `model.named_parameters()` entry, in that order
    (virtex/factories.py:529-533), and Lookahead.state_dict() is the wrapped SGD's.  The fused optimizer keeps
    its buffers in backward-execution order internally, so its state dict must be re-indexed: compare momentum,
    lr, weight decay PER PARAMETER NAME against the unfused optimizer (stock SGD over named order) stepped
    with the same gradients, assert shapes, and load each one's state into the other."""
    dev = select(backend)
    torch.manual_seed(3)
    model_f = _Toy().to(dev)
    model_u = copy.deepcopy(model_f)
    buckets = vd.GradientBuckets(model_f, bucket_mb=0.01)
    assert [id(p) for p in buckets.params] != [id(p) for p in model_f.parameters()]      # the orders really differ
    fused = FusedPretrainOptimizer(model_f, buckets, total_steps=40, warmup_steps=4, start_step=2)
    unfused = PretrainOptimizer(model_u, total_steps=40, warmup_steps=4, start_step=2)
    for it in range(3):
        buckets.zero()
        for p, q, g in zip(model_f.parameters(), model_u.parameters(), _grads(model_u, it)):
            p.grad.add_(g.to(dev))
            q.grad = g.to(dev).clone()
        fused.step()
        unfused.step()
    sf, su = fused.state_dict(), unfused.state_dict()
    names = [n for n, _ in model_f.named_parameters()]
    assert len(sf["param_groups"]) == len(su["param_groups"]) == len(names)
    for i, (n, p) in enumerate(model_f.named_parameters()):
        gf, gu = sf["param_groups"][i], su["param_groups"][i]
        assert gf["params"] == [i] and gu["params"] == [i]
        assert gf["weight_decay"] == gu["weight_decay"], n
        assert abs(gf["lr"] - gu["lr"]) <= 1e-12 + 1e-6 * gu["lr"], n
        assert abs(gf["initial_lr"] - gu["initial_lr"]) <= 1e-9, n
        mf, mu = sf["state"][i]["momentum_buffer"], su["state"][i]["momentum_buffer"]
        assert tuple(mf.shape) == tuple(p.shape) == tuple(mu.shape), n
        assert torch.allclose(mf.cpu(), mu.cpu(), rtol=2e-5, atol=1e-6), n
    # cross-loading: the unfused state into a fresh fused optimizer and vice versa, then one more identical step
    torch.manual_seed(4)
    model_g = copy.deepcopy(model_u)
    buckets_g = vd.GradientBuckets(model_g, bucket_mb=0.01)
    fused_g = FusedPretrainOptimizer(model_g, buckets_g, total_steps=40, warmup_steps=4)
    fused_g.load_state_dict(su)
    model_v = copy.deepcopy(model_u)
    unfused_v = PretrainOptimizer(model_v, total_steps=40, warmup_steps=4)
    unfused_v.load_state_dict(sf)
    gs = _grads(model_u, 9)
    buckets_g.zero()
    for p, q, g in zip(model_g.parameters(), model_v.parameters(), gs):
        p.grad.add_(g.to(dev))
        q.grad = g.to(dev).clone()
    fused_g.step()
    unfused_v.step()
    for (n, p), q in zip(model_g.named_parameters(), model_v.parameters()):
        assert torch.allclose(p.detach().cpu(), q.detach().cpu(), rtol=2e-5, atol=1e-6), n
    # a stock SGD with the reference grouping takes the fused file without swapping hyper-parameters
    model_c = copy.deepcopy(model_u).cpu()
    sgd = torch.optim.SGD(port.param_groups(model_c.named_parameters()), momentum=0.9)
    sgd.load_state_dict({"state": {k: {"momentum_buffer": v["momentum_buffer"].cpu()} for k, v in sf["state"].items()},
                         "param_groups": sf["param_groups"]})
    for (n, p), g in zip(model_c.named_parameters(), sgd.param_groups):
        assert g["weight_decay"] == (0.0 if ("layer_norm" in n or n.endswith("transformer.bias")) else 1e-4), n
        assert tuple(sgd.state[p]["momentum_buffer"].shape) == tuple(p.shape), n
    # wrong shape / wrong group count are refused instead of broadcast
    bad = copy.deepcopy(sf)
    bad["state"][0]["momentum_buffer"] = torch.zeros(3)
    with pytest.raises(ValueError):
        fused_g.load_state_dict(bad)
    bad2 = {**sf, "param_groups": sf["param_groups"][:-1]}
    with pytest.raises(ValueError):
        fused_g.load_state_dict(bad2)


@pytest.mark.parametrize("backend", BACKENDS)
def test_device_side_schedule_equals_host_side_schedule(backend):
    """What a replayed hipGraph of the step runs (virtex_amd/graph.py): step counter, LR multiplier (warm-up, then cos^2) and the
    Lookahead phase live in device memory and reach the optimizer kernel through a pointer (vtx_sgd_lookahead_step_dev).
    Twelve steps across the end of the warm-up and two Lookahead syncs must equal the by-value path; the host mirrors
    (step index, Lookahead counter) are recovered by sync_host(), and a state dict taken afterwards resumes either path."""
    dev = select(backend)
    torch.manual_seed(0)
    base = _Toy()
    base.visual.cnn[0].weight.data = base.visual.cnn[0].weight.data.contiguous(memory_format=torch.channels_last)
    ma, mb = copy.deepcopy(base).to(dev), copy.deepcopy(base).to(dev)
    ba, bb = vd.GradientBuckets(ma, bucket_mb=0.01), vd.GradientBuckets(mb, bucket_mb=0.01)
    oa = FusedPretrainOptimizer(ma, ba, total_steps=50, warmup_steps=10, start_step=3)
    ob = FusedPretrainOptimizer(mb, bb, total_steps=50, warmup_steps=10, start_step=3)
    ob.enable_device_schedule()
    # the dropout epoch is ONE word per device, shared by every optimizer that is on the device-side schedule (a recording of an
    # earlier test that is still alive holds it too): what this optimizer adds to it is asserted, not its absolute value
    epoch0 = int(ob.dev["epoch"].item())
    try:
        _steps(ma, oa, ba, dev, range(12))
        _steps(mb, ob, bb, dev, range(12))
        for (n, p), (_, q) in zip(ma.named_parameters(), mb.named_parameters()):
            assert torch.allclose(p.detach().cpu(), q.detach().cpu(), rtol=1e-6, atol=1e-7), n
        ob.sync_host()
        assert ob.step_idx == oa.step_idx == 15 and ob.kc == oa.kc
        assert int(ob.dev["epoch"].item()) - epoch0 == 12         # the dropout epoch advanced once per step
        sd = ob.state_dict()
        assert sd["virtex_amd"] == oa.state_dict()["virtex_amd"]
    finally:
        ob.disable_device_schedule()
    assert ob.dev is None
    _steps(ma, oa, ba, dev, range(12, 15))
    _steps(mb, ob, bb, dev, range(12, 15))                        # back on the by-value path, same trajectory
    for (n, p), (_, q) in zip(ma.named_parameters(), mb.named_parameters()):
        assert torch.allclose(p.detach().cpu(), q.detach().cpu(), rtol=1e-6, atol=1e-7), n


@pytest.mark.parametrize("backend", BACKENDS)
def test_dropout_epoch_changes_the_masks_on_the_device(backend):
    """vtx_set_dropout_epoch: the kernels mix a device word into their (by-value) seed at entry, so that launches with
    identical arguments -- a replayed graph -- draw different masks once the word has changed, and identical ones while it
    has not (forward and backward of one step)."""
    from virtex_amd import ops
    dev = select(backend)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(64, 256, generator=g).to(dev); y = torch.randn(64, 256, generator=g).to(dev)
    gamma, beta = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    epoch = torch.zeros(1, dtype=torch.int32, device=dev)
    plain = ops.layernorm_residual_fwd(x, y, gamma, beta, 1e-5, 0.5, 1234)[0].clone()
    try:
        ops.set_dropout_epoch(epoch)
        a = ops.layernorm_residual_fwd(x, y, gamma, beta, 1e-5, 0.5, 1234)[0].clone()
        b = ops.layernorm_residual_fwd(x, y, gamma, beta, 1e-5, 0.5, 1234)[0].clone()
        epoch.add_(1)
        c = ops.layernorm_residual_fwd(x, y, gamma, beta, 1e-5, 0.5, 1234)[0].clone()
    finally:
        ops.set_dropout_epoch(None)
    assert torch.equal(a, b) and torch.equal(a, plain)            # epoch 0 adds nothing to the seed
    assert not torch.equal(a, c)
    # the backward of the same step re-derives the same mask: dy is dz where the mask kept the element, 0 elsewhere
    epoch.fill_(7)
    try:
        ops.set_dropout_epoch(epoch)
        out, mean, rstd = ops.layernorm_residual_fwd(x, y, gamma, beta, 1e-5, 0.5, 99)
        dg, db = torch.zeros(256, device=dev), torch.zeros(256, device=dev)
        dout = torch.randn(64, 256, generator=g).to(dev)
        dz, dy = ops.layernorm_residual_bwd(x, y, gamma, mean, rstd, dout, dg, db, 0.5, 99)
    finally:
        ops.set_dropout_epoch(None)
    kept_bwd = dy != 0
    frac = kept_bwd.float().mean().item()
    assert 0.4 < frac < 0.6
    assert torch.allclose(dy[kept_bwd], 2.0 * dz[kept_bwd], rtol=1e-5, atol=1e-6)
