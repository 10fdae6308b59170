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
