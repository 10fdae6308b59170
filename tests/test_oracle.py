"""The oracle port vs (a) committed golden fixtures generated from the verbatim reference
and (b) the live reference when /root/reference exists.  CPU only."""
import json
import os

import pytest
import torch

from oracle import bicaptioning as port
from oracle import make_goldens, reference_import, synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


@pytest.mark.parametrize("case", sorted(make_goldens.CASES))
def test_port_reproduces_reference_goldens(case):
    gold = _load(case)
    mkw, bkw = make_goldens.CASES[case]
    rec = make_goldens.run_case(case, mkw, bkw, use_reference=False)
    # same torch build => same ATen kernels => equality up to thread-order rounding
    assert rec["loss"] == pytest.approx(gold["loss"], rel=1e-6)
    for k, v in gold["loss_components"].items():
        assert rec["loss_components"][k] == pytest.approx(v, rel=1e-6)
    assert torch.allclose(torch.tensor(rec["logits_sample"]), torch.tensor(gold["logits_sample"]),
                          rtol=1e-5, atol=1e-6)
    assert torch.allclose(torch.tensor(rec["backward_logits_sample"]),
                          torch.tensor(gold["backward_logits_sample"]), rtol=1e-5, atol=1e-6)
    assert set(rec["grads"]) == set(gold["grads"])
    for name, g in gold["grads"].items():
        r = rec["grads"][name]
        assert r["norm"] == pytest.approx(g["norm"], rel=1e-4, abs=1e-7), name
        assert torch.allclose(torch.tensor(r["samples"]), torch.tensor(g["samples"]),
                              rtol=1e-3, atol=1e-6 + 1e-4 * g["norm"]), name
    for name, b in gold["buffers"].items():
        assert rec["buffers"][name] == pytest.approx(b, rel=1e-5), name


@pytest.mark.parametrize("case", sorted(make_goldens.CASES))
def test_port_reproduces_reference_eval_goldens(case):
    """Eval mode (running-statistics BatchNorm, argmax predictions) of the port vs the verbatim reference."""
    gold = _load(case + "_eval")
    mkw, bkw = make_goldens.CASES[case]
    rec = make_goldens.run_eval_case(mkw, bkw, use_reference=False)
    assert rec["loss"] == pytest.approx(gold["loss"], rel=1e-6)
    for k, v in gold["loss_components"].items():
        assert rec["loss_components"][k] == pytest.approx(v, rel=1e-6)
    assert rec["predictions"] == gold["predictions"]
    assert rec["features_norm"] == pytest.approx(gold["features_norm"], rel=1e-6)
    assert torch.allclose(torch.tensor(rec["features_sample"]), torch.tensor(gold["features_sample"]), rtol=1e-5, atol=1e-6)


def test_state_dict_layout_matches_survey():
    m = port.build_model(dropout=0.0)
    assert len(m.state_dict()) == 370
    params = list(m.parameters())
    assert len(params) == 202
    assert sum(p.numel() for p in params) == 69482320
    assert m.textual.output.weight is m.textual.embedding.words.weight
    assert m.backward_textual.embedding is m.textual.embedding
    groups = port.param_groups(m.named_parameters())
    assert sum(g["weight_decay"] == 0.0 for g in groups) == 26
    assert sum(g["lr"] == 0.2 for g in groups) == 159


@pytest.mark.reference
def test_port_is_bitwise_equal_to_live_reference():
    mkw, bkw = make_goldens.CASES["r50_l2_h128_b3_small"]
    o = synth.seeded_model(port.build_model, seed=0, dropout=0.0, **mkw)
    r = reference_import.build_reference_model(dropout=0.0, **mkw)
    assert list(r.state_dict()) == list(o.state_dict())
    r.load_state_dict(o.state_dict())
    batch = synth.synthetic_batch(**bkw)
    o.train(), r.train()
    lo, lr = o(batch)["loss"], r(batch)["loss"]
    lo.backward(), lr.backward()
    assert lo.item() == lr.item()
    for (n, p), (_, q) in zip(o.named_parameters(), r.named_parameters()):
        assert torch.equal(p.grad, q.grad), n
    for (n, p), (_, q) in zip(o.named_buffers(), r.named_buffers()):
        assert torch.equal(p, q), n


@pytest.mark.reference
def test_train_step_matches_reference_optimizer_chain():
    """TrainStep (restated SGD grouping + Lookahead + cosine warm-up + clip) against the
    reference's own Lookahead / LinearWarmupCosineAnnealingLR driven as in
    scripts/pretrain_virtex.py:145-163."""
    ref = reference_import.import_reference()
    mkw, bkw = make_goldens.CASES["r50_l2_h128_b3_small"]
    o = synth.seeded_model(port.build_model, seed=0, dropout=0.0, **mkw)
    r = reference_import.build_reference_model(dropout=0.0, **mkw)
    r.load_state_dict(o.state_dict())
    o.train(), r.train()
    # start inside warm-up so that LR is non-zero
    step = port.TrainStep(o, total_steps=100, warmup_steps=10, start_step=5)
    opt = ref.Lookahead(torch.optim.SGD(port.param_groups(r.named_parameters()), momentum=0.9),
                        k=5, alpha=0.5)
    sched = ref.LinearWarmupCosineAnnealingLR(opt, total_steps=100, warmup_steps=10)
    for _ in range(5):
        sched.step()
    for it in range(6):
        batch = synth.synthetic_batch(**{**bkw, "seed": 10 + it})
        lo = step(batch)
        opt.zero_grad()
        lr = r(batch)["loss"]
        lr.backward()
        torch.nn.utils.clip_grad_norm_(r.parameters(), 10.0)
        opt.step()
        sched.step()
        assert lo.item() == pytest.approx(lr.item(), rel=1e-6)
    for (n, p), (_, q) in zip(o.named_parameters(), r.named_parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7), n
