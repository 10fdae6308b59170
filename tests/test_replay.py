"""virtex_amd.replay.StepReplay: the training step re-issued from a recorded launch list (no Python per launch, no autograd)
must BE the training step -- same loss, same gradients, same parameters step after step -- on fresh batches, with the
optimizer's schedule and the dropout epoch advancing on the device."""
import copy

import pytest
import torch

from backends import BACKENDS, select
from oracle import synth
from virtex_amd import distributed as vd
import virtex_amd.factories as vf
from virtex_amd.optim import FusedPretrainOptimizer
from virtex_amd.replay import StepReplay

KW = dict(textual="transdec_postnorm::L1_H128_A2_F256", vocab_size=304)


def _setup(dev, dropout, dtype=torch.float32):
    torch.manual_seed(0)
    model = vf.build_bicaptioning_model(dropout=dropout, compute_dtype=dtype, **KW).to(dev).train()
    buckets = vd.GradientBuckets(model, bucket_mb=1.0)
    opt = FusedPretrainOptimizer(model, buckets, total_steps=50, warmup_steps=6, start_step=2, lookahead_k=3)
    return model, buckets, opt


def _batch(seed, dev, B=2):
    b = synth.synthetic_batch(batch_size=B, image_size=64, max_len=8, vocab_size=304, seed=seed, ragged=True)
    return {k: v.to(dev) for k, v in b.items()}


def _eager_step(model, buckets, opt, batch):
    buckets.zero(); buckets.begin()
    out = model(batch)
    out["loss"].backward()
    opt.step(grad_scale=buckets.finish())
    return out["loss"].detach().clone()


@pytest.mark.parametrize("backend", BACKENDS)
def test_replayed_steps_equal_eager_steps(backend):
    dev = select(backend)
    dt = torch.float32 if backend == "emu" else torch.bfloat16
    ma, ba, oa = _setup(dev, 0.0, dt)
    mb, bb, ob = _setup(dev, 0.0, dt)
    mb.load_state_dict(ma.state_dict())
    ob.flat_slow.copy_(ob.flat_p)
    oa.flat_slow.copy_(oa.flat_p)
    oa.enable_device_schedule()
    try:
        replay = StepReplay(mb, bb, ob, _batch(100, dev), warmup=1, validate=True)      # runs (and validates) on batch 100
        assert replay.validated["kernel"] > 100 and replay.validated["ops"] >= replay.validated["kernel"]
        # the construction advanced model b by warmup + validation + recording steps on batch 100: bring a to the same point
        n_pre = int(round(ob.dev["step"].item())) - 2
        for _ in range(n_pre):
            _eager_step(ma, ba, oa, _batch(100, dev))
        for (n, p), (_, q) in zip(ma.named_parameters(), mb.named_parameters()):
            assert torch.allclose(p.detach().cpu(), q.detach().cpu(), rtol=2e-4, atol=1e-6), ("pre", n)
        for it in range(4):                                                               # crosses a Lookahead sync and the warm-up end
            batch = _batch(200 + it, dev)
            la = _eager_step(ma, ba, oa, batch)
            lb = replay(batch).clone()
            assert abs(la.item() - lb.item()) <= 1e-5 * abs(la.item()), (it, la.item(), lb.item())
        odd = _batch(300, dev)
        odd = dict(odd, caption_tokens=odd["caption_tokens"][:, :-1].contiguous(), noitpac_tokens=odd["noitpac_tokens"][:, :-1].contiguous())
        with pytest.raises(ValueError):                                                   # a recording is a list of launches on fixed shapes
            replay(odd)
        replay.sync()
        oa.sync_host()
        assert ob.step_idx == oa.step_idx and ob.kc == oa.kc
        tol = dict(rtol=1e-3, atol=1e-5) if dt == torch.float32 else dict(rtol=2e-2, atol=1e-3)
        for (n, p), (_, q) in zip(ma.named_parameters(), mb.named_parameters()):
            assert torch.allclose(p.detach().float().cpu(), q.detach().float().cpu(), **tol), n
        for (n, p), (_, q) in zip(ma.named_buffers(), mb.named_buffers()):
            assert torch.allclose(p.detach().float().cpu(), q.detach().float().cpu(), rtol=1e-3, atol=1e-5), n
    finally:
        oa.disable_device_schedule()
        ob.disable_device_schedule()


@pytest.mark.parametrize("backend", BACKENDS)
def test_replay_draws_new_dropout_masks(backend):
    """the recorded seeds are constants; the device epoch (incremented by the optimizer step of every replay) makes the masks
    differ from replay to replay: same batch, frozen parameters (LR 0) -> different losses"""
    dev = select(backend)
    dt = torch.float32 if backend == "emu" else torch.bfloat16
    torch.manual_seed(0)
    model = vf.build_bicaptioning_model(dropout=0.3, compute_dtype=dt, **KW).to(dev).train()
    buckets = vd.GradientBuckets(model, bucket_mb=1.0)
    opt = FusedPretrainOptimizer(model, buckets, cnn_lr=0.0, lr=0.0, weight_decay=0.0, total_steps=50, warmup_steps=6, start_step=2)
    try:
        replay = StepReplay(model, buckets, opt, _batch(7, dev), warmup=1, validate=backend != "emu")   # (validation: the test above)
        losses = [replay(None).item() for _ in range(4)]
        assert len({round(l, 6) for l in losses}) == 4, losses
    finally:
        opt.disable_device_schedule()


@pytest.mark.gpu
def test_every_recorded_op_is_replayed_on_the_stream_it_was_recorded_on():
    """The autograd engine switches streams with a C++ guard (no Python setter runs): the recorder has to notice and record the
    switch itself, else `event.record()` and ATen operators of a backward node land on the stream the LAST recorded switch left.
    Walk the list the way the replay does and compare with the stream every op saw when it was recorded."""
    import re
    dev = select("gpu")
    model, buckets, opt = _setup(dev, 0.1, torch.bfloat16)
    try:
        replay = StepReplay(model, buckets, opt, _batch(3, dev), warmup=1, validate=True)
        rec = replay.rec
        assert rec.counts["engine_switch"] >= 1              # the branch head's backward runs on the branch stream
        cur = rec.start[0]
        for i, (label, at) in enumerate(zip(rec.labels, rec.at_stream)):
            m = re.match(r"stream:engine switch to stream#(\d+)", label) or re.match(r"stream:_cuda_setStream .*'stream_id': (\d+)", label)
            if m:
                cur = int(m.group(1))
                continue
            assert at == cur, (i, label, at, cur)
    finally:
        opt.disable_device_schedule()


def _autograd_nodes(tensors):
    """grad_fn nodes reachable from `tensors`"""
    seen, stack = set(), [t.grad_fn for t in tensors if isinstance(t, torch.Tensor) and t.grad_fn is not None]
    while stack:
        n = stack.pop()
        if n is None or n in seen:
            continue
        seen.add(n)
        stack.extend(fn for fn, _ in n.next_functions)
    return len(seen)


@pytest.mark.parametrize("backend", BACKENDS)
def test_replays_do_not_grow_the_autograd_graph_and_two_optimizers_share_the_epoch(backend):
    """(1) The re-issued ATen operators run with grad mode off: the recorded tensors that still carry a grad_fn must not
    collect a new CopyBackwards node per replay (a host-memory leak over a 500k-step run).  (2) The dropout epoch is one
    reference-counted word per device: a second optimizer enabling / disabling its device schedule neither re-points nor
    nulls the word the first one's replays advance."""
    dev = select(backend)
    dt = torch.float32 if backend == "emu" else torch.bfloat16
    model, buckets, opt = _setup(dev, 0.1, dt)
    other_m, other_b, other = _setup(dev, 0.1, dt)
    try:
        replay = StepReplay(model, buckets, opt, _batch(11, dev), warmup=1, validate=False)
        kept = [t for t in replay.rec.keep if isinstance(t, torch.Tensor)]
        replay(None)
        n0 = _autograd_nodes(kept)
        for _ in range(3):
            replay(None)
        assert _autograd_nodes(kept) == n0
        other.enable_device_schedule()
        assert other.dev["epoch"] is opt.dev["epoch"]
        e0 = int(opt.dev["epoch"].item())
        other.disable_device_schedule()                      # the first optimizer's registration survives ...
        replay(None)
        assert int(opt.dev["epoch"].item()) == e0 + 1        # ... and its replays still advance the registered word
        la, lb = replay(None).item(), replay(None).item()
        assert la != lb                                      # masks still change from replay to replay
    finally:
        opt.disable_device_schedule()
        other.disable_device_schedule()
