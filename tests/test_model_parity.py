"""Whole-model parity: the HIP modules (driven through the C ABI) against the CPU oracle and the
committed reference goldens -- loss, both loss components, every parameter gradient (202 tensors
incl. the tied word matrix), BatchNorm running statistics.  fp32 compute: the north-star bound is
1e-3 relative; bf16 compute is checked against the same oracle with a looser, documented bound."""
import json
import os

import pytest
import torch

from backends import BACKENDS, rel_err, select
from oracle import bicaptioning as port
from oracle import make_goldens, synth

import virtex_amd.factories as vf

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _build_pair(case, dev, dtype):
    mkw, bkw = make_goldens.CASES[case]
    oracle_model = synth.seeded_model(port.build_model, seed=0, dropout=0.0, **mkw)
    model = vf.build_bicaptioning_model(visual="torchvision::resnet50", textual=mkw["textual"],
                                        vocab_size=mkw["vocab_size"], dropout=0.0, compute_dtype=dtype,
                                        max_caption_length=30)
    missing = model.load_state_dict(oracle_model.state_dict())
    assert not missing.missing_keys and not missing.unexpected_keys
    model = model.to(dev)
    batch = synth.synthetic_batch(**bkw)
    return oracle_model, model, batch


def _dump_rows(name, summary, rows):
    """Measured distances of a GPU parity run -> gpurun_out/ (copied into profiles/ as the round's evidence)."""
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, name), "w") as f:
        json.dump({"summary": summary, "rows": rows}, f, indent=1)


def _run(model, batch, dev):
    model.train()
    out = model({k: v.to(dev) for k, v in batch.items()})
    out["loss"].backward()
    return out


def backbone_rule(rows, per_tensor=2.5, median=1.5):
    """The calibrated backbone-gradient rule of tests/test_fidelity.py, with NO absolute floor: rows = (name, ours vs
    the fp64 oracle, the reference's own fp32 vs its fp64 evaluation).  Every tensor within `per_tensor` x the
    reference's own distance for that tensor (a tensor whose own distance happens to lie below the median of all
    tensors is held to the median: the per-tensor distance is itself a random variable), and the median over the
    tensors within `median` x the reference's median."""
    med_ref = sorted(r[2] for r in rows)[len(rows) // 2]
    med_mine = sorted(r[1] for r in rows)[len(rows) // 2]
    worst = max(rows, key=lambda r: r[1] / max(r[2], med_ref))
    for n, mine, ref in rows:
        assert mine <= per_tensor * max(ref, med_ref), (n, mine, ref, med_ref)
    assert med_mine <= median * med_ref, (med_mine, med_ref)
    return {"median_ref": med_ref, "median_mine": med_mine, "worst": (worst[0], worst[1], worst[2])}


def _check(case, dev, dtype, text_tol, loss_tol, cnn_factor, cnn_floor):
    """Gradients of the text side (and the loss) are well conditioned: assert the north-star
    bound directly.  Gradients of the ResNet are NOT: at these batch sizes the reference's own
    fp32 CPU path differs from its fp64 evaluation by 0.3-3 % relative L2 (53 BatchNorm
    backward projections amplify rounding; see DESIGN.md "Parity").  We therefore measure that
    distance here and require ours to stay within a factor of the reference's own.
    `cnn_floor` = None (every GPU case): the per-tensor rule of tests/test_fidelity.py (`backbone_rule`), no
    absolute floor -- a loose constant can never be the bound that passes.  `cnn_floor` > 0 (ONLY the 3-image 64x64
    emulator toy: 12 samples per channel in the last stage, single ReLU sign flips worth ~1e-2 each dominate and the
    reference's own distance is not a usable scale): median / maximum over the tensors within `cnn_factor` x the
    reference's own, floored."""
    import copy

    oracle_model, model, batch = _build_pair(case, dev, dtype)
    oracle64 = copy.deepcopy(oracle_model).double()
    oracle_model.train(), oracle64.train()
    oo = oracle_model(batch)
    oo["loss"].backward()
    o64 = oracle64({k: (v.double() if v.dtype.is_floating_point else v) for k, v in batch.items()})
    o64["loss"].backward()
    out = _run(model, batch, dev)
    assert abs(out["loss"].item() - oo["loss"].item()) < loss_tol * abs(oo["loss"].item())
    for k in ("captioning_forward", "captioning_backward"):
        assert abs(out["loss_components"][k].item() - oo["loss_components"][k].item()) < loss_tol * 12
    # reference goldens (generated from the verbatim reference classes)
    with open(os.path.join(GOLDEN, case + ".json")) as f:
        gold = json.load(f)
    assert abs(out["loss"].item() - gold["loss"]) < loss_tol * abs(gold["loss"])
    names = [n for n, _ in oracle_model.named_parameters()]
    assert names == [n for n, _ in model.named_parameters()]
    worst_text = ("", 0.0)
    cnn_mine, cnn_ref, cnn_rows = [], [], []
    for (n, p), (_, q), (_, r) in zip(model.named_parameters(), oracle_model.named_parameters(),
                                      oracle64.named_parameters()):
        assert p.grad is not None and p.grad.shape == q.grad.shape, n
        if "cnn" in n:
            mine, ref = rel_err(p.grad.cpu(), r.grad), rel_err(q.grad, r.grad)
            cnn_mine.append(mine), cnn_ref.append(ref)
            if r.grad.norm() > 0:
                cnn_rows.append((n, mine, ref))
            g = gold["grads"][n]   # golden norms (verbatim reference, fp32) within the same band
            assert abs(p.grad.double().norm().item() - g["norm"]) <= 3 * max(mine, ref, cnn_floor or 0.0) * g["norm"] + 1e-9, n
        else:
            e = rel_err(p.grad.cpu(), q.grad)
            g = gold["grads"][n]
            assert abs(p.grad.double().norm().item() - g["norm"]) <= 3 * text_tol * g["norm"] + 1e-9, n
            if e > worst_text[1]:
                worst_text = (n, e)
    assert worst_text[1] < text_tol, f"worst text-side gradient mismatch {worst_text}"
    # backbone: aggregate over the 161 tensors (a single ReLU sign flip moves one tensor by ~1e-2)
    if cnn_floor is None:
        summary = backbone_rule(cnn_rows)
        _dump_rows(f"parity_fp32_{case}.json", summary, cnn_rows)
    else:
        cnn_mine, cnn_ref = sorted(cnn_mine), sorted(cnn_ref)
        med_m, med_r = cnn_mine[len(cnn_mine) // 2], cnn_ref[len(cnn_ref) // 2]
        assert med_m <= cnn_factor * max(med_r, cnn_floor), (med_m, med_r)
        assert cnn_mine[-1] <= cnn_factor * max(cnn_ref[-1], 4 * cnn_floor), (cnn_mine[-1], cnn_ref[-1])
    for (n, b), (_, c) in zip(model.named_buffers(), oracle_model.named_buffers()):
        if b.dtype.is_floating_point:
            assert rel_err(b.cpu(), c) < (1e-4 if dtype == torch.float32 else 2e-2), n
        else:
            assert int(b) == int(c), n
    # tied matrix row 0 (padding index) receives the dense projection gradient
    assert model.textual.embedding.words.weight.grad[0].abs().sum().item() > 0


def _check_eval(case, dev, dtype, feat_tol, loss_tol):
    """Eval mode: BatchNorm on running statistics, folded into the convolutions;
    the model returns loss, components and argmax predictions like the reference's validation pass
    (captioning.py:99-143).  Compared with the oracle and with the reference's own eval goldens."""
    oracle_model, model, batch = _build_pair(case, dev, dtype)
    oracle_model.eval(), model.eval()
    feats = {}
    h1 = oracle_model.visual.register_forward_hook(lambda m, i, o: feats.__setitem__("oracle", o))
    h2 = model.visual.register_forward_hook(lambda m, i, o: feats.__setitem__("ours", o))
    with torch.no_grad():
        oo = oracle_model(batch)
        out = model({k: v.to(dev) for k, v in batch.items()})
    h1.remove(), h2.remove()
    assert feats["ours"].shape == feats["oracle"].shape
    assert rel_err(feats["ours"].float().cpu(), feats["oracle"]) < feat_tol
    with open(os.path.join(GOLDEN, case + "_eval.json")) as f:
        gold = json.load(f)
    assert abs(out["loss"].item() - oo["loss"].item()) < loss_tol * abs(oo["loss"].item())
    assert abs(out["loss"].item() - gold["loss"]) < loss_tol * abs(gold["loss"])
    for k in ("captioning_forward", "captioning_backward"):
        assert abs(out["loss_components"][k].item() - gold["loss_components"][k]) < loss_tol * 12
    assert abs(feats["ours"].double().norm().item() - gold["features_norm"]) < feat_tol * gold["features_norm"]
    pred, gpred = out["predictions"].cpu(), torch.tensor(gold["predictions"])
    assert pred.shape == gpred.shape
    agree = (pred == gpred).float().mean().item()
    assert agree >= (0.999 if dtype == torch.float32 else 0.5), agree   # fp32: argmax ties only; bf16 on a random-init model (near-uniform logits) flips close calls
    # running statistics untouched by an eval pass
    for (n, b), (_, c) in zip(model.named_buffers(), oracle_model.named_buffers()):
        assert torch.equal(b.cpu(), c), n
    # a second call reuses the cached folded weights and gives the same answer
    with torch.no_grad():
        again = model({k: v.to(dev) for k, v in batch.items()})
    assert again["loss"].item() == out["loss"].item()
    # ... until a weight changes in place
    with torch.no_grad():
        model.visual.cnn.bn1.running_mean.add_(0.5)
        changed = model({k: v.to(dev) for k, v in batch.items()})
    assert changed["loss"].item() != out["loss"].item()


@pytest.mark.emu
def test_small_model_eval_fp32_emulator():
    _check_eval("r50_l2_h128_b3_small", select("emu"), torch.float32, feat_tol=1e-4, loss_tol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["r50_l2_h128_b3_small", "r50_l1_h1024_b2_full", "r50_l1_h1024_b2_ragged"])
def test_model_eval_fp32_gpu(case):
    _check_eval(case, select("gpu"), torch.float32, feat_tol=1e-4, loss_tol=1e-5)


@pytest.mark.gpu
def test_model_eval_bf16_gpu():
    _check_eval("r50_l1_h1024_b2_full", select("gpu"), torch.bfloat16, feat_tol=3e-2, loss_tol=5e-3)


def _check_frozen(case, dev, dtype, text_tol, loss_tol):
    """VISUAL.FROZEN semantics (visual_backbones.py:49-53): backbone parameters do not require grad and the
    CNN sits in eval mode while the text heads train on top of it."""
    oracle_model, model, batch = _build_pair(case, dev, dtype)
    for m in (oracle_model, model):
        m.train()
        for p in m.visual.cnn.parameters():
            p.requires_grad = False
        m.visual.cnn.eval()
    oo = oracle_model(batch)
    oo["loss"].backward()
    out = model({k: v.to(dev) for k, v in batch.items()})
    out["loss"].backward()
    assert abs(out["loss"].item() - oo["loss"].item()) < loss_tol * abs(oo["loss"].item())
    for (n, p), (_, q) in zip(model.named_parameters(), oracle_model.named_parameters()):
        if "cnn" in n:
            assert p.grad is None and q.grad is None, n
        else:
            assert rel_err(p.grad.cpu(), q.grad) < text_tol, n
    for (n, b), (_, c) in zip(model.named_buffers(), oracle_model.named_buffers()):
        assert torch.equal(b.cpu(), c), n      # eval-mode BN: running statistics untouched


@pytest.mark.emu
def test_small_model_frozen_backbone_fp32_emulator():
    _check_frozen("r50_l2_h128_b3_small", select("emu"), torch.float32, text_tol=1e-3, loss_tol=1e-5)


@pytest.mark.gpu
def test_model_frozen_backbone_fp32_gpu():
    _check_frozen("r50_l1_h1024_b2_ragged", select("gpu"), torch.float32, text_tol=1e-3, loss_tol=1e-5)


@pytest.mark.emu
def test_small_model_eval_requires_no_grad_or_frozen():
    dev = select("emu")
    _, model, batch = _build_pair("r50_l2_h128_b3_small", dev, torch.float32)
    model.eval()
    with pytest.raises(RuntimeError, match="eval-mode backbone"):
        model({k: v.to(dev) for k, v in batch.items()})


@pytest.mark.emu
def test_small_model_fp32_emulator():
    dev = select("emu")
    _check("r50_l2_h128_b3_small", dev, torch.float32, text_tol=1e-3, loss_tol=1e-5, cnn_factor=4.0, cnn_floor=1.5e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["r50_l2_h128_b3_small", "r50_l1_h1024_b2_full", "r50_l1_h1024_b2_ragged"])
def test_model_fp32_gpu(case):
    dev = select("gpu")
    # the 3-image 64x64 toy is the emulator's case run on hardware: the toy's floor (see _check); the 224x224 cases: the rule
    toy = case == "r50_l2_h128_b3_small"
    _check(case, dev, torch.float32, text_tol=1e-3, loss_tol=1e-5, cnn_factor=4.0 if toy else None, cnn_floor=1.5e-2 if toy else None)


@pytest.mark.emu
def test_small_prenorm_model_fp32_emulator():
    """the "transdec_prenorm" product (reference: factories.py:358-366): loss, every gradient (incl. the closing LayerNorm
    of the stack) and the buffers against the oracle and the reference's golden of that case"""
    _check("r50_l2_h128_b3_prenorm", select("emu"), torch.float32, text_tol=1e-3, loss_tol=1e-5, cnn_factor=4.0, cnn_floor=1.5e-2)


@pytest.mark.emu
def test_small_prenorm_model_eval_fp32_emulator():
    _check_eval("r50_l2_h128_b3_prenorm", select("emu"), torch.float32, feat_tol=1e-4, loss_tol=1e-5)


@pytest.mark.gpu
def test_prenorm_model_fp32_gpu():
    dev = select("gpu")
    _check("r50_l2_h128_b3_prenorm", dev, torch.float32, text_tol=1e-3, loss_tol=1e-5, cnn_factor=4.0, cnn_floor=1.5e-2)
    _check_eval("r50_l2_h128_b3_prenorm", dev, torch.float32, feat_tol=1e-4, loss_tol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("image_size,max_len", [(256, 30), (224, 44), (320, 40)])
def test_shapes_beyond_the_tuned_attention_envelope_fp32_gpu(image_size, max_len):
    """Another crop size (256 x 256 -> an 8 x 8 grid of 64 keys; 320 x 320 -> 100) or a longer caption limit (44 > 32 queries) takes
    the general attention kernels: loss and every text-side gradient against the oracle at the north-star bound, backbone
    gradients finite (their rule lives in the golden cases)."""
    dev = select("gpu")
    kw = dict(textual="transdec_postnorm::L1_H128_A2_F256", vocab_size=500, max_caption_length=max_len)
    oracle_model = synth.seeded_model(port.build_model, seed=0, dropout=0.0, **kw).train()
    model = vf.build_bicaptioning_model(visual="torchvision::resnet50", dropout=0.0, compute_dtype=torch.float32, **kw)
    missing = model.load_state_dict(oracle_model.state_dict())
    assert not missing.missing_keys and not missing.unexpected_keys
    model = model.to(dev)
    batch = synth.synthetic_batch(2, image_size=image_size, max_len=max_len, vocab_size=500, seed=5, ragged=True)
    oo = oracle_model(batch)
    oo["loss"].backward()
    out = _run(model, batch, dev)
    assert abs(out["loss"].item() - oo["loss"].item()) < 1e-5 * abs(oo["loss"].item())
    for (n, p), (_, q) in zip(model.named_parameters(), oracle_model.named_parameters()):
        assert torch.isfinite(p.grad).all(), n
        if "cnn" not in n:
            assert rel_err(p.grad.cpu(), q.grad) < 1e-3, n


@pytest.mark.gpu
def test_prenorm_model_bf16_gpu():
    """The pre-norm head in the throughput mode on the 3-image toy case: loss to 2e-3, every gradient finite, and every text-side
    gradient held to what `torch.autocast(bfloat16)` of the oracle itself reaches on this case (computed here, on the CPU):
    calibration cosine >= 0.99 -> ours >= 0.99 (the closing LayerNorm of the stack is in this class); 0.90 .. 0.99 (everything fed
    by the 2 x 2 grid of a bf16 backbone at 12 samples per channel: visual projection, cross-attention weights) -> ours >= the
    calibration - 0.03; below 0.90 the calibration is noise and bounds nothing -- that is norm2 of every layer, whose gradient
    in a PRE-norm layer arrives only through the cross-attention queries (softmax over 4 near-equal logits at random init:
    the stock autocast run is at cosine 0.37 .. 0.67 there, ours 0.46 .. 0.64)."""
    import copy
    dev = select("gpu")
    oracle_model, model, batch = _build_pair("r50_l2_h128_b3_prenorm", dev, torch.bfloat16)
    oracle_model.train()
    cal_model = copy.deepcopy(oracle_model)
    oo = oracle_model(batch)
    oo["loss"].backward()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        cal_loss = cal_model(batch)["loss"]
    cal_loss.backward()
    out = _run(model, batch, dev)
    assert abs(out["loss"].item() - oo["loss"].item()) < 2e-3 * abs(oo["loss"].item())

    def cos(x, y):
        x, y = x.double().flatten(), y.double().flatten()
        return (x @ y / (x.norm() * y.norm())).item()
    tight, calibrated, unbounded = [], [], []
    for (n, p), (_, q), (_, c) in zip(model.named_parameters(), oracle_model.named_parameters(), cal_model.named_parameters()):
        assert torch.isfinite(p.grad).all(), n
        if "cnn" in n:
            continue
        mine, cal = cos(p.grad.cpu(), q.grad), cos(c.grad, q.grad)
        if cal >= 0.99:
            assert mine >= 0.99, (n, mine, cal)
            tight.append(n)
        elif cal >= 0.90:
            assert mine >= cal - 0.03, (n, mine, cal)
            calibrated.append(n)
        else:
            unbounded.append(n)
    assert any("transformer.norm." in n for n in tight)
    assert all(".norm2." in n for n in unbounded), unbounded
    assert len(tight) >= 60 and len(unbounded) <= 8, (len(tight), len(calibrated), len(unbounded))


@pytest.mark.parametrize("backend", BACKENDS)
def test_prenorm_dropout_masks_of_forward_and_backward_agree(backend):
    """With dropout on, the pre-norm joins x + dropout(y) are GEMM epilogues and their backward masks come from
    vtx_dropout_bwd with the same (seed, element index): the text-side gradients must be those of the torch graph built
    from the SAME masks.  The masks are recovered from the kernels themselves (identity weights are not needed: one
    layer, compared through a directional derivative of the loss, which only matches when every mask agrees)."""
    dev = select(backend)
    import virtex_amd.modules.textual_heads as th
    torch.manual_seed(0)
    head = th.TransformerDecoderTextualHead(64, 200, 128, 2, 2, 256, dropout=0.3, norm_first=True,
                                            max_caption_length=12, compute_dtype=torch.float32).to(dev).train()
    vis = torch.randn(2, 64, 3, 3, device=dev)
    tok = torch.randint(1, 200, (2, 12), device=dev)
    lens = torch.tensor([12, 7], device=dev)
    w = torch.randn(2, 12, 128, device=dev)
    params = [p for p in head.parameters() if p.requires_grad]
    direction = [torch.randn_like(p) * 1e-2 for p in params]

    def loss_at(eps):
        if eps:
            for p, d in zip(params, direction):
                p.add_(d, alpha=eps)
        th.dropout_seed_state(1234)                      # the same masks at every evaluation
        out = (head.features(vis, tok, lens).float() * w).sum()
        if eps:
            for p, d in zip(params, direction):
                p.add_(d, alpha=-eps)
        return out
    loss_at(0.0).backward()
    analytic = sum((p.grad.double() * d.double()).sum().item() for p, d in zip(params, direction) if p.grad is not None)
    h = 1e-2
    with torch.no_grad():
        numeric = (loss_at(h).double().item() - loss_at(-h).double().item()) / (2 * h)
    assert abs(analytic - numeric) <= 2e-3 * max(abs(numeric), 1.0), (analytic, numeric)


@pytest.mark.gpu
def test_model_bf16_gpu():
    """bf16 compute mode (the throughput mode) is NOT a parity claim; on this B = 2 golden case it is bounded against the fp32
    oracle where bf16 storage permits: loss to 2e-3, every gradient finite, every text-side gradient with cosine >= 0.99.
    The BACKBONE gradients are deliberately not bounded here: at B = 2 `torch.autocast(bfloat16)` of the reference is itself
    at median rel 1.32 / min cosine -0.17 from its fp32 run (profiles/r04_parity_bf16_b2_backbone_vs_autocast.json) -- a
    calibration against it passes for uncorrelated gradients.  The backbone claim of the bf16 mode
    lives where the calibration means something: tests/test_fidelity.py (B = 32: autocast min cosine 0.979, asserted >= 0.95
    there before it is used as a bound; B = 256: median <= 0.25, min cosine >= 0.95)."""
    dev = select("gpu")
    oracle_model, model, batch = _build_pair("r50_l1_h1024_b2_full", dev, torch.bfloat16)
    oracle_model.train()
    oo = oracle_model(batch)
    oo["loss"].backward()
    out = _run(model, batch, dev)
    assert abs(out["loss"].item() - oo["loss"].item()) < 2e-3 * abs(oo["loss"].item())
    for (n, p), (_, q) in zip(model.named_parameters(), oracle_model.named_parameters()):
        assert torch.isfinite(p.grad).all(), n
        if "cnn" not in n:
            a, b = p.grad.cpu().double().flatten(), q.grad.double().flatten()
            assert (a @ b / (a.norm() * b.norm())).item() > 0.99, n


def test_state_dict_layout():
    """370 keys / 202 unique parameter tensors, reference names."""
    model = vf.build_bicaptioning_model(dropout=0.0)
    ref = port.build_model(dropout=0.0)
    assert list(model.state_dict().keys()) == list(ref.state_dict().keys())
    for (k, a), (_, b) in zip(model.state_dict().items(), ref.state_dict().items()):
        assert a.shape == b.shape, k
    assert len(list(model.parameters())) == 202
    assert sum(p.numel() for p in model.parameters()) == 69482320
    assert model.textual.output.weight is model.textual.embedding.words.weight
    assert model.backward_textual.embedding is model.textual.embedding
    assert model.backward_textual.transformer is not model.textual.transformer


@pytest.mark.parametrize("backend", BACKENDS)
def test_direct_gradient_accumulation_matches_autograd_path(backend):
    """With pre-allocated flat gradient buffers (the data-parallel / fused-optimizer set-up) the backward
    functions accumulate in place (virtex_amd/gradsink.py); results must equal the plain autograd path,
    and autograd must still fire every parameter's post-accumulate hook exactly once per step (the
    data-parallel engine launches its bucket all-reduces from that hook)."""
    from virtex_amd import distributed as vd

    dev = select(backend)
    _, model, batch = _build_pair("r50_l2_h128_b3_small", dev, torch.float32)
    _run(model, batch, dev)
    ref = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    buckets = vd.GradientBuckets(model)
    seen = []
    hooks = [p.register_post_accumulate_grad_hook(lambda q: seen.append(q)) for p in model.parameters()]
    try:
        buckets.zero(); buckets.begin()
        _run(model, batch, dev)
    finally:
        for h in hooks:
            h.remove()
    assert len(seen) == len(set(id(q) for q in seen)) == len(list(model.parameters()))
    worst = max((rel_err(p.grad.cpu(), ref[n].cpu()), n) for n, p in model.named_parameters())
    assert worst[0] < 1e-4, worst
    assert all(p.grad.data_ptr() >= buckets.flat.data_ptr() for p in model.parameters())


@pytest.mark.gpu
@pytest.mark.parametrize("visual,textual", [("torchvision::resnet50", "transdec_postnorm::L4_H1024_A16_F4096"),
                                            ("torchvision::resnet101", "transdec_postnorm::L1_H2048_A32_F8192")])
def test_other_baseline_configs_fp32_gpu(visual, textual):
    """BASELINE.json configs 4 and 5 (depth ablation L4, ResNet-101 + H2048): loss and every text-side
    gradient against the oracle at B=2, fp32 (the backbone gradients are covered by the calibrated test)."""
    dev = select("gpu")
    kw = dict(visual=visual, textual=textual, vocab_size=10000)
    oracle_model = synth.seeded_model(port.build_model, seed=0, dropout=0.0, **kw).train()
    model = vf.build_bicaptioning_model(dropout=0.0, compute_dtype=torch.float32, **kw)
    model.load_state_dict(oracle_model.state_dict())
    model = model.to(dev)
    batch = synth.synthetic_batch(2, seed=5, ragged=True)
    oo = oracle_model(batch)
    oo["loss"].backward()
    out = _run(model, batch, dev)
    assert abs(out["loss"].item() - oo["loss"].item()) < 1e-5 * abs(oo["loss"].item())
    worst = max((rel_err(p.grad.cpu(), q.grad), n) for (n, p), (_, q) in
                zip(model.named_parameters(), oracle_model.named_parameters()) if "cnn" not in n)
    assert worst[0] < 1e-3, worst
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())


@pytest.mark.gpu
@pytest.mark.parametrize("visual,textual", [("torchvision::resnet101", "transdec_postnorm::L1_H2048_A32_F8192"),
                                            ("torchvision::wide_resnet50_2", "transdec_postnorm::L1_H1024_A16_F4096")])
def test_other_backbones_every_gradient_fp32_gpu(visual, textual):
    """BASELINE.json config 5 (configs/backbone_ablations/bicaptioning_R_101_L1_H1024.yaml:1-5 +
    configs/width_ablations/bicaptioning_R_50_L1_H2048.yaml:1-5) and the wide_resnet50_2 ablation: ONE fp32 training step
    at B = 8, 224x224, through the hand-scheduled ResNet forward/backward -- loss against the oracle, every text-side
    gradient at the north-star bound, and EVERY backbone gradient (314 tensors for ResNet-101) per tensor against the fp64
    oracle with the calibrated rule of tests/test_fidelity.py (`backbone_rule`: within 2.5x the reference's own
    fp32<->fp64 distance for that tensor, median within 1.5x; no floor)."""
    import copy
    dev = select("gpu")
    kw = dict(visual=visual, textual=textual, vocab_size=10000)
    om = synth.seeded_model(port.build_model, seed=0, dropout=0.0, **kw).train()
    batch = synth.synthetic_batch(8, image_size=224, seed=11, ragged=True)
    o64 = copy.deepcopy(om).double()
    state = copy.deepcopy(om.state_dict())           # BEFORE the oracle's step updates the BatchNorm buffers
    oo = om(batch); oo["loss"].backward()
    o64({k: (v.double() if v.dtype.is_floating_point else v) for k, v in batch.items()})["loss"].backward()
    g64 = {n: p.grad for n, p in o64.named_parameters()}
    model = vf.build_bicaptioning_model(dropout=0.0, compute_dtype=torch.float32, **kw)
    missing = model.load_state_dict(state)
    assert not missing.missing_keys and not missing.unexpected_keys
    model = model.to(dev)
    out = _run(model, batch, dev)
    assert abs(out["loss"].item() - oo["loss"].item()) < 1e-5 * abs(oo["loss"].item())
    rows = []
    for (n, p), (_, q) in zip(model.named_parameters(), om.named_parameters()):
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
        if "cnn" in n:
            if g64[n].norm() > 0:
                rows.append((n, rel_err(p.grad.cpu(), g64[n]), rel_err(q.grad, g64[n])))
        else:
            assert rel_err(p.grad.cpu(), q.grad) < 1e-3, n
    assert len(rows) >= 150
    summary = backbone_rule(rows)
    worst_buf = ("", 0.0)
    for (n, b), (_, c) in zip(model.named_buffers(), om.named_buffers()):
        if b.dtype.is_floating_point:
            e = rel_err(b.cpu(), c)
            worst_buf = max(worst_buf, (n, e), key=lambda t: t[1])
        else:
            assert int(b) == int(c), n
    summary["worst_running_statistic"] = worst_buf
    _dump_rows(f"parity_fp32_b8_{visual.split('::')[1]}.json", summary, rows)
    assert worst_buf[1] < 1e-3, worst_buf          # BatchNorm running statistics after the step: the north-star bound


def _train_then_eval(dev, steps=2):
    """Two optimizer steps (fused HIP optimizer, direct-to-buffer gradients, BatchNorm statistics updated inside
    the kernels), then an eval-mode forward: the folded inference weights must reflect BOTH the updated parameters
    and the updated running statistics (neither change goes through a torch operator)."""
    from virtex_amd import distributed as vd
    from virtex_amd.optim import FusedPretrainOptimizer
    oracle_model, model, batch = _build_pair("r50_l2_h128_b3_small", dev, torch.float32)
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    with torch.no_grad():                       # populate the fold cache with the initial state
        model.eval()
        first = model(dbatch)["loss"].item()
    step = port.TrainStep(oracle_model, total_steps=50, warmup_steps=5, start_step=3)
    buckets = vd.GradientBuckets(model, bucket_mb=1.0)
    opt = FusedPretrainOptimizer(model, buckets, total_steps=50, warmup_steps=5, start_step=3)
    for _ in range(steps):
        oracle_model.train(), model.train()
        step(batch)
        buckets.zero(); buckets.begin()
        model(dbatch)["loss"].backward()
        opt.step(grad_scale=buckets.finish())
    oracle_model.eval(), model.eval()
    with torch.no_grad():
        ref = oracle_model(batch)["loss"].item()
        out = model(dbatch)["loss"].item()
    assert abs(ref - first) > 1e-3 * abs(first)            # the state really moved
    assert abs(out - ref) < 2e-3 * abs(ref), (out, ref, first)


@pytest.mark.emu
def test_eval_after_fused_training_steps_sees_updated_weights_and_statistics_emulator():
    _train_then_eval(select("emu"))


@pytest.mark.gpu
def test_eval_after_fused_training_steps_sees_updated_weights_and_statistics_gpu():
    _train_then_eval(select("gpu"))


@pytest.mark.emu
def test_uint8_image_batches_are_normalised_on_the_device_emulator():
    """The backbone accepts decoder output directly: uint8 (B,H,W,3) gives the same features as the reference's
    wire format (float CHW, normalised on the CPU with ImageNet mean/std)."""
    from virtex_amd import ops
    dev = select("emu")
    _, model, _ = _build_pair("r50_l2_h128_b3_small", dev, torch.float32)
    model.eval()
    g = torch.Generator().manual_seed(3)
    u8 = torch.randint(0, 256, (2, 64, 64, 3), generator=g, dtype=torch.uint8)
    mean = torch.tensor(ops.IMAGENET_COLOR_MEAN); std = torch.tensor(ops.IMAGENET_COLOR_STD)
    chw = ((u8.float() - 255.0 * mean) * (1.0 / (255.0 * std))).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        a = model.visual(u8.to(dev))
        b = model.visual(chw.to(dev))
    assert torch.equal(a.cpu(), b.cpu())


@pytest.mark.emu
def test_odd_image_width_uses_the_unpacked_stem_layout_emulator():
    """Odd widths cannot use the two-pixels-per-chunk stem layout (the zero-framed row would have an odd length):
    the backbone falls back to 8-channel pixels with in-kernel padding.  Train and eval forward against the oracle."""
    dev = select("emu")
    oracle_model, model, _ = _build_pair("r50_l2_h128_b3_small", dev, torch.float32)
    image = synth.synthetic_batch(batch_size=2, image_size=63, max_len=8, vocab_size=1000, seed=9)["image"]
    for mode in ("train", "eval"):
        getattr(oracle_model, mode)(), getattr(model, mode)()
        with torch.no_grad():
            ref = oracle_model.visual(image)
            out = model.visual(image.to(dev))
        assert out.shape == ref.shape
        assert rel_err(out.float().cpu(), ref) < 5e-4, mode      # 2x2x2 = 8 samples per channel in the last BatchNorms


@pytest.mark.emu
@pytest.mark.parametrize("cnn", ["wide_resnet50_2", "resnet101"])
def test_other_backbones_features_and_input_gradient_path_emulator(cnn):
    """The reference's backbone ablations (configs/backbone_ablations: resnet101, wide_resnet50_2) through the same
    hand-scheduled forward/backward: features vs the oracle in train mode, and a finite, oracle-like loss gradient
    on a text-side tensor after a full backward through the wider / deeper backbone."""
    dev = select("emu")
    kw = dict(visual=f"torchvision::{cnn}", textual="transdec_postnorm::L1_H128_A2_F256", vocab_size=300)
    oracle_model = synth.seeded_model(port.build_model, seed=0, dropout=0.0, **kw).train()
    model = vf.build_bicaptioning_model(dropout=0.0, compute_dtype=torch.float32, max_caption_length=30, **kw)
    missing = model.load_state_dict(oracle_model.state_dict())
    assert not missing.missing_keys and not missing.unexpected_keys
    model = model.to(dev).train()
    batch = synth.synthetic_batch(batch_size=3, image_size=64, max_len=8, vocab_size=300, seed=4, ragged=True)
    with torch.no_grad():
        ref = oracle_model.visual(batch["image"])
        out = model.visual(batch["image"].to(dev))
    assert out.shape == ref.shape and rel_err(out.float().cpu(), ref) < 2e-3     # 12 samples per channel in the last stage
    oo = oracle_model(batch); oo["loss"].backward()
    mo = model({k: v.to(dev) for k, v in batch.items()}); mo["loss"].backward()
    assert abs(mo["loss"].item() - oo["loss"].item()) < 1e-4 * abs(oo["loss"].item())
    a, b = model.textual.visual_projection.weight.grad.cpu(), oracle_model.textual.visual_projection.weight.grad
    assert rel_err(a, b) < 5e-3
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


@pytest.mark.parametrize("backend", BACKENDS)
def test_bf16_batchnorm_fusions_match_the_standalone_kernels(backend):
    """bf16 step with the BatchNorm sums taken in the convolution epilogues (backward: input-gradient epilogue ->
    ReLU mask + sum dz, sum dz*xhat; forward: conv epilogue -> batch statistics) against the same step on the
    stand-alone BatchNorm kernels.  The backward fusion leaves the forward bit-identical, so its gradients must agree
    to bf16 rounding.  The forward fusion moves batch means by ~1e-7 (other summation order), which bf16 rounding
    turns into single-ulp differences of some activations and those into ReLU-mask flips -- in bf16 ANY perturbation of
    the forward moves backbone gradients by 0.1-0.4 (profiles/r02_bf16_rounding_mechanism.txt) -- so it is compared on
    the loss and the statistics, and its backward fusion against ITS forward."""
    from virtex_amd.modules import visual_backbones as vb
    dev = select(backend)
    _, model, batch = _build_pair("r50_l2_h128_b3_small", dev, torch.bfloat16)
    saved = (vb.FUSE_BN_BWD, vb.FUSE_BN_STATS)
    start = {n: b.detach().clone() for n, b in model.named_buffers()}
    try:
        runs = {}
        for name, (fb, ff) in {"plain": (False, False), "bwd": (True, False), "fwd": (False, True), "both": (True, True)}.items():
            vb.FUSE_BN_BWD, vb.FUSE_BN_STATS = fb, ff
            with torch.no_grad():
                for n, b in model.named_buffers():
                    b.copy_(start[n])
            model.zero_grad(set_to_none=True)
            out = _run(model, batch, dev)
            runs[name] = (out["loss"].item(), {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters()},
                          {n: b.detach().float().cpu().clone() for n, b in model.named_buffers()})
    finally:
        vb.FUSE_BN_BWD, vb.FUSE_BN_STATS = saved
    for fused, base in (("bwd", "plain"), ("both", "fwd")):
        (l, g, _), (l0, g0, _) = runs[fused], runs[base]
        assert l == l0, fused                                     # same forward
        rels = sorted(rel_err(g[n], g0[n]) for n in g0 if "cnn" in n and g0[n].norm() > 0)
        assert rels[len(rels) // 2] < 3e-2 and rels[-1] < 0.25, (fused, rels[len(rels) // 2], rels[-1])
        for n in g0:
            if "cnn" not in n:
                assert rel_err(g[n], g0[n]) < 2e-2, (fused, n)
    (l, _, bufs), (l0, _, bufs0) = runs["fwd"], runs["plain"]
    assert abs(l - l0) < 5e-4 * abs(l0)
    for n in bufs0:       # running statistics after the step; only the first stage: deeper layers of this 3-image toy amplify the ulp-level differences
        if n.startswith("visual.cnn.bn1") or n.startswith("visual.cnn.layer1"):
            assert rel_err(bufs[n], bufs0[n]) < 5e-3, n


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_reference_amp_loop_is_transparent_gpu(dtype):
    """the reference wraps the forward in `amp.autocast` and the backward / step in a `GradScaler`
    (scripts/pretrain_virtex.py:107,150-161).  The native modules compute in their own dtype and keep fp32 parameter
    gradients, so that loop must be numerically transparent: the (2^16-scaled) backward neither overflows nor loses
    bits, `unscale_` restores the gradients, no step is skipped, and three steps with the reference's optimizer
    grouping reproduce the oracle's un-scaled fp32 loop (loss per step; the 2nd and 3rd depend on the updates)."""
    dev = select("gpu")
    kw = dict(textual="transdec_postnorm::L1_H128_A2_F256", vocab_size=1000)
    oracle_model = synth.seeded_model(port.build_model, seed=0, dropout=0.0, max_caption_length=12, **kw)
    model = vf.build_bicaptioning_model(dropout=0.0, compute_dtype=dtype, max_caption_length=12, **kw)
    model.load_state_dict(oracle_model.state_dict())
    model = model.to(dev).train()
    oracle_step = port.TrainStep(oracle_model.train(), total_steps=2000, warmup_steps=200)
    groups = port.param_groups(model.named_parameters())
    base = [g["lr"] for g in groups]
    optimizer = torch.optim.SGD(groups, momentum=0.9)
    scaler = torch.amp.GradScaler("cuda", enabled=True)
    tol = 2e-4 if dtype == torch.float32 else 5e-3
    for it in range(3):
        batch = synth.synthetic_batch(4, image_size=128, max_len=12, vocab_size=1000, seed=60 + it, ragged=True)
        expect = oracle_step(batch).item()
        mult = port.lr_multiplier(it, 2000, 200)
        for g, b in zip(optimizer.param_groups, base):
            g["lr"] = b * mult
        optimizer.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16, enabled=True):
            out = model({k: v.to(dev) for k, v in batch.items()})
            loss = out["loss"]
        assert loss.dtype == torch.float32
        scaler.scale(loss).backward()
        scaler.unscale_(optimizer)
        assert all(torch.isfinite(p.grad).all() for p in model.parameters())
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        scaler.step(optimizer)
        scaler.update()
        assert scaler.get_scale() == 65536.0                    # no inf/nan was found: nothing skipped, scale untouched
        assert abs(loss.item() - expect) < tol * abs(expect), (it, loss.item(), expect)


@pytest.mark.parametrize("backend", BACKENDS)
def test_shared_visual_projection_equals_the_per_head_evaluation(backend):
    """The tied `visual_projection` evaluated once for both heads (models.SHARE_VISUAL_PROJECTION) against the
    reference's order (once per head, textual_heads.py:245): same loss, same gradients up to fp32 summation order,
    incl. the projection's own weight / bias gradient, which now comes from ONE weight-gradient GEMM on the summed
    memory gradient."""
    from virtex_amd import models
    dev = select(backend)
    _, model, batch = _build_pair("r50_l2_h128_b3_small", dev, torch.float32)
    saved = models.SHARE_VISUAL_PROJECTION
    try:
        runs = {}
        for mode in (False, True):
            models.SHARE_VISUAL_PROJECTION = mode
            model.zero_grad(set_to_none=True)
            out = _run(model, batch, dev)
            runs[mode] = (out["loss"].item(), {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()})
    finally:
        models.SHARE_VISUAL_PROJECTION = saved
    assert abs(runs[True][0] - runs[False][0]) <= 1e-6 * abs(runs[False][0])
    for n, g in runs[False][1].items():
        if "cnn" not in n:
            assert rel_err(runs[True][1][n], g) < 1e-5, n
    assert rel_err(runs[True][1]["textual.visual_projection.weight"], runs[False][1]["textual.visual_projection.weight"]) < 1e-5
    worst = max(rel_err(runs[True][1][n], g) for n, g in runs[False][1].items() if "cnn" in n and g.norm() > 0)
    assert worst < 2e-2          # the backbone sees d(features) through one GEMM instead of two: fp32 summation order, amplified by the toy's conditioning


def _generation_counts(reset=False):
    import ctypes
    from virtex_amd import _lib
    g1, g2 = ctypes.c_long(0), ctypes.c_long(0)
    _lib.call("vtx_contraction_generation_counts", ctypes.byref(g1), ctypes.byref(g2), ctypes.c_int(1 if reset else 0))
    return g1.value, g2.value


def _assert_step_stays_on_the_dma_kernel(case, backend):
    """Every GEMM-shaped launch of a bf16 training step (53 convolutions x {forward, input gradient, weight gradient},
    the text heads' linears, the tied projection) must run on the DMA kernel with buffer-descriptor addressing: a shape
    that silently falls back to the register-staged kernel would still be correct, only slower -- so it is a test."""
    dev = select(backend)
    _, model, batch = _build_pair(case, dev, torch.bfloat16)
    _generation_counts(reset=True)
    out = _run(model, batch, dev)
    assert torch.isfinite(out["loss"])
    g1, g2 = _generation_counts()
    assert g2 >= 200 and g1 == 0, (g1, g2)
    # ... and the fp32 parity mode never touches it
    _, model32, batch32 = _build_pair(case, dev, torch.float32)
    _generation_counts(reset=True)
    _run(model32, batch32, dev)
    g1, g2 = _generation_counts()
    assert g1 >= 200 and g2 == 0, (g1, g2)


def test_bf16_step_stays_on_the_dma_kernel_emulator():
    _assert_step_stays_on_the_dma_kernel("r50_l2_h128_b3_small", "emu")


@pytest.mark.gpu
def test_bf16_step_stays_on_the_dma_kernel_gpu():
    _assert_step_stays_on_the_dma_kernel("r50_l1_h1024_b2_full", "gpu")


@pytest.mark.parametrize("backend", BACKENDS)
def test_fused_stem_forward_tail_leaves_the_step_unchanged(backend):
    """VIRTEX_AMD_FUSE_STEM_FWD (BatchNorm + ReLU + max-pool of the stem in one pass, the tensor between them never
    written): loss, every gradient and the BatchNorm buffers bit-identical to the three-kernel path."""
    from virtex_amd.modules import visual_backbones as vb
    dev = select(backend)
    _, model, batch = _build_pair("r50_l2_h128_b3_small", dev, torch.bfloat16)
    start = {n: b.detach().clone() for n, b in model.named_buffers()}
    saved = vb.FUSE_STEM_FWD
    runs = {}
    try:
        for flag in (False, True):
            vb.FUSE_STEM_FWD = flag
            with torch.no_grad():
                for n, b in model.named_buffers():
                    b.copy_(start[n])
            model.zero_grad(set_to_none=True)
            out = _run(model, batch, dev)
            runs[flag] = (out["loss"].item(), {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters()},
                          {n: b.detach().float().cpu().clone() for n, b in model.named_buffers()})
    finally:
        vb.FUSE_STEM_FWD = saved
    if backend == "emu":
        assert runs[True][0] == runs[False][0]
        for n in runs[False][1]:
            if "embedding" in n:            # fed by fp32 atomics: not bit-reproducible run to run
                continue
            assert torch.equal(runs[True][1][n], runs[False][1][n]), n
        for n in runs[False][2]:
            assert torch.equal(runs[True][2][n], runs[False][2][n]), n
    else:       # on hardware the two stem kernels are separately compiled: a pooled element may differ in its last bf16 bit,
                # and in bf16 ANY forward perturbation moves backbone gradients (see the fusion test above): loss and buffers
        assert abs(runs[True][0] - runs[False][0]) < 5e-4 * abs(runs[False][0])
        for n in runs[False][2]:
            if n.startswith("visual.cnn.bn1"):
                assert rel_err(runs[True][2][n], runs[False][2][n]) < 1e-5, n


@pytest.mark.parametrize("backend", BACKENDS)
def test_relu_mask_bits_leave_the_step_unchanged(backend):
    """VIRTEX_AMD_RELU_BITS: the fused BatchNorm backward of a Bottleneck's bn3 reads the block output's ReLU mask as one
    bit per element (written by the BatchNorm + residual + ReLU pass) instead of the whole output tensor: loss and every
    gradient bit-identical to the tensor-mask path."""
    from virtex_amd.modules import visual_backbones as vb
    dev = select(backend)
    _, model, batch = _build_pair("r50_l2_h128_b3_small", dev, torch.bfloat16)
    start = {n: b.detach().clone() for n, b in model.named_buffers()}
    saved = vb.RELU_BITS
    runs = {}
    try:
        for flag in (False, True):
            vb.RELU_BITS = flag
            with torch.no_grad():
                for n, b in model.named_buffers():
                    b.copy_(start[n])
            model.zero_grad(set_to_none=True)
            out = _run(model, batch, dev)
            runs[flag] = (out["loss"].item(), {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters()})
    finally:
        vb.RELU_BITS = saved
    assert runs[True][0] == runs[False][0]
    for n in runs[False][1]:
        if "embedding" in n:            # fed by fp32 atomics: not bit-reproducible run to run
            continue
        assert torch.equal(runs[True][1][n], runs[False][1][n]), n


@pytest.mark.parametrize("backend", BACKENDS)
def test_padding_to_a_fixed_caption_length_changes_nothing_on_the_kernels(backend):
    """data.collate_captions(pad_to=T) -- the one batch shape launch replay needs -- on the HIP path: extra padding columns must
    be masked by the attention kernels (key padding), zeroed by the embedding kernel and ignored by the fused projection + loss:
    same loss, same gradients as the batch padded to its longest caption (fp32; text side to summation order, the backbone
    through the toy's conditioning)."""
    dev = select(backend)
    _, model, batch = _build_pair("r50_l2_h128_b3_small", dev, torch.float32)
    lengths = batch["caption_lengths"].clone()
    T = batch["caption_tokens"].shape[1]
    lengths[:] = torch.tensor([max(3, T - 3 - i) for i in range(lengths.numel())])           # ragged, all shorter than T
    tok = batch["caption_tokens"].clone()
    for i, L in enumerate(lengths.tolist()):
        tok[i, L - 1] = 2; tok[i, L:] = 0
    rev = torch.zeros_like(tok)
    for i, L in enumerate(lengths.tolist()):
        rev[i, :L] = tok[i, :L].flip(0)
    longest = int(lengths.max())
    short = dict(batch, caption_tokens=tok[:, :longest].contiguous(), noitpac_tokens=rev[:, :longest].contiguous(), caption_lengths=lengths)
    fixed = dict(batch, caption_tokens=tok, noitpac_tokens=rev, caption_lengths=lengths)
    assert longest < T
    runs = []
    for b in (short, fixed):
        model.zero_grad(set_to_none=True)
        out = _run(model, b, dev)
        runs.append((out["loss"].item(), {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}))
    assert abs(runs[0][0] - runs[1][0]) <= 2e-6 * abs(runs[0][0]), (runs[0][0], runs[1][0])
    for n, g in runs[0][1].items():
        if "cnn" not in n:
            assert rel_err(runs[1][1][n], g) < 2e-5, n
    worst = max(rel_err(runs[1][1][n], g) for n, g in runs[0][1].items() if "cnn" in n and g.norm() > 0)
    assert worst < 2e-2


@pytest.mark.parametrize("backend", BACKENDS)
def test_fused_conv3_backward_leaves_the_step_unchanged(backend):
    """VIRTEX_AMD_FUSE_CONV3_BWD: bn3's backward, conv3's input gradient (+ bn2's fused backward epilogue) and conv3's weight
    gradient of the stage-1 Bottlenecks in ONE streaming kernel (csrc/conv3_bwd.hip) instead of three launches and a stored
    gradient tensor: the same forward, and every gradient equal to accumulation order (the rounded values that travel between
    the kernels -- dx3, the masked gradient -- are produced by the same formulas)."""
    from virtex_amd import ops
    from virtex_amd.modules import visual_backbones as vb
    dev = select(backend)
    _, model, batch = _build_pair("r50_l2_h128_b3_small", dev, torch.bfloat16)
    start = {n: b.detach().clone() for n, b in model.named_buffers()}
    saved = vb.FUSE_CONV3_BWD
    runs = {}
    try:
        for flag in (False, True):
            vb.FUSE_CONV3_BWD = flag
            with torch.no_grad():
                for n, b in model.named_buffers():
                    b.copy_(start[n])
            model.zero_grad(set_to_none=True)
            ops.profile_start()
            out = _run(model, batch, dev)
            rec = ops.profile_stop()
            fused_launches = sum(r["launches"] for r in rec if "conv3_bwd_fused" in r["name"])
            assert fused_launches == (3 if flag else 0), fused_launches          # the three Bottlenecks of stage 1
            runs[flag] = (out["loss"].item(), {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters()})
    finally:
        vb.FUSE_CONV3_BWD = saved
    assert runs[True][0] == runs[False][0]
    rels = {n: rel_err(runs[True][1][n], g0) for n, g0 in runs[False][1].items() if g0.norm() > 0}
    cnn = sorted(v for n, v in rels.items() if "cnn" in n)
    # 128 of the 159 tensors come out bit-identical; the rest differ through the handful of elements whose bn2 ReLU mask the two
    # forms decide differently at the threshold (the fused kernel tests the forward's expression (x - mean) * scale + beta > 0, the
    # epilogue xhat * gamma + beta > 0) -- a few 1e-3, except the stem's bn1.bias: a near-cancelling sum over all pixels of a 3-image
    # 64 x 64 toy, 0.046-0.055 depending on nothing more than the order in which the stem kernel's workgroups sum their fp32
    # statistics (plain / XCD-major strip walk).  Median and 95th percentile carry the claim; the maximum is a sanity bound.
    assert cnn[len(cnn) // 2] < 5e-3 and cnn[int(len(cnn) * 0.95)] < 2e-2 and cnn[-1] < 1e-1, \
        (cnn[len(cnn) // 2], cnn[int(len(cnn) * 0.95)], cnn[-1], max(rels, key=rels.get))
    for n, v in rels.items():
        if "cnn" not in n:
            assert v < 1e-5 or "embedding" in n, (n, v)                           # nothing upstream of the backbone moves


# ----------------------------------------------------------------------------------------------------------------------
# ONE Bottleneck, forward + backward with a supplied upstream gradient, fp32, against the fp64 oracle at a FLAT 1e-3
# (VERDICT round 5, item 1b).  The whole-backbone gradients are held to a calibrated rule (`backbone_rule`) because 53
# BatchNorm backward projections amplify rounding until the reference's own fp32 differs from its fp64 by ~2 %; a single
# block is short enough for that argument not to apply, so here the north-star bound itself is asserted: output, input
# gradient, the three (four) weight gradients, every BatchNorm gamma / beta gradient and the running statistics.
# Block = torchvision Bottleneck v1.5, reached from /root/reference/virtex/modules/visual_backbones.py:68-74.
# ----------------------------------------------------------------------------------------------------------------------
BOTTLENECK_CASES = {   # name: (stage, block, channels in, [(backend, batch, height)])
    "stage1_block0_downsample": (1, 0, 64, {"emu": (2, 8), "gpu": (8, 56)}),
    "stage1_block1_identity": (1, 1, 256, {"emu": (2, 8), "gpu": (8, 56)}),
    "stage4_block0_downsample_stride2": (4, 0, 1024, {"gpu": (8, 14)}),
    "stage4_block1_identity": (4, 1, 2048, {"gpu": (8, 7)}),
    "stage2_block0_downsample_stride2": (2, 0, 256, {"emu": (2, 8)}),
}


def _single_bottleneck(case, backend, dtype):
    import copy
    stage, blk, cin, sizes = BOTTLENECK_CASES[case]
    if backend not in sizes:
        pytest.skip(f"{case} is not run on {backend}")
    dev = select(backend)
    B, H = sizes[backend]
    oracle_model = synth.seeded_model(port.build_model, seed=0, dropout=0.0, textual="transdec_postnorm::L1_H128_A2_F256",
                                      vocab_size=304).train()
    model = vf.build_bicaptioning_model(textual="transdec_postnorm::L1_H128_A2_F256", vocab_size=304, dropout=0.0,
                                        compute_dtype=dtype)
    model.load_state_dict(oracle_model.state_dict())
    model = model.to(dev).train()
    g = torch.Generator().manual_seed(100 * stage + blk)
    x = torch.relu(torch.randn(B, cin, H, H, generator=g))                  # a block's input is a ReLU output
    oblk = getattr(oracle_model.visual.cnn, f"layer{stage}")[blk]
    o64 = copy.deepcopy(oblk).double().train()
    o32 = copy.deepcopy(oblk).train()
    x64 = x.double().requires_grad_(True)
    y64 = o64(x64)
    up = torch.randn(y64.shape, generator=g)                                # the supplied upstream gradient
    y64.backward(up.double())
    x32 = x.clone().requires_grad_(True)
    if dtype == torch.bfloat16:       # the comparison column = PyTorch's own bf16 AMP of the reference block (calibration of the bf16 bound)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            y32 = o32(x32)
        y32 = y32.float()
    else:
        y32 = o32(x32)
    y32.backward(up)
    xd = x.to(dev).requires_grad_(True)
    model.zero_grad(set_to_none=True)
    y = model.visual.forward_blocks(xd, stage, blk, 1)
    assert y.shape == y64.shape
    y.backward(up.to(dev))
    mine = getattr(model.visual.cnn, f"layer{stage}")[blk]
    rows = [("output", rel_err(y.detach().float().cpu(), y64.detach()), rel_err(y32.detach(), y64.detach())),
            ("input_gradient", rel_err(xd.grad.float().cpu(), x64.grad), rel_err(x32.grad, x64.grad))]
    for (n, p), (_, q64), (_, q32) in zip(mine.named_parameters(), o64.named_parameters(), o32.named_parameters()):
        assert p.grad is not None, n
        rows.append((n, rel_err(p.grad.float().cpu(), q64.grad), rel_err(q32.grad, q64.grad)))
    bufs = [(n, rel_err(b.float().cpu(), b64)) for (n, b), (_, b64) in zip(mine.named_buffers(), o64.named_buffers())
            if b.dtype.is_floating_point]
    counters = [(n, int(b), int(b64)) for (n, b), (_, b64) in zip(mine.named_buffers(), o64.named_buffers())
                if not b.dtype.is_floating_point]
    return rows, bufs, counters


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", sorted(BOTTLENECK_CASES))
def test_single_bottleneck_fp32_flat_1e3(backend, case):
    rows, bufs, counters = _single_bottleneck(case, backend, torch.float32)
    if backend == "gpu":
        _dump_rows(f"parity_fp32_bottleneck_{case}.json", {"bound": 1e-3, "worst": max(rows, key=lambda r: r[1])}, rows)
    for n, mine, ref in rows:
        assert mine < 1e-3, (case, n, mine, "the reference's own fp32 vs fp64:", ref)       # flat: no calibration
    for n, e in bufs:
        assert e < 1e-4, (case, n, e)
    for n, a, b in counters:
        assert a == b, (case, n, a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["stage1_block1_identity", "stage4_block0_downsample_stride2"])
def test_single_bottleneck_bf16_gpu(case):
    """The benchmarked precision through the same entry point (fused BatchNorm epilogues, bit masks).  A 16-bit forward flips
    the ReLU masks of the elements within its rounding distance of zero, and a flipped element moves its gradient by 100 %: the
    block's gradients sit 5-10 % from the fp64 oracle in ANY bf16 implementation (first hardware run: 0.03-0.10).  The bound is
    therefore measured in place, like tests/test_fidelity.py does for the whole step: the same block of the reference under
    `torch.autocast(bfloat16)` against the same fp64 oracle, per tensor; ours may not be further away than 1.5x that (output:
    bf16 storage rounding)."""
    rows, bufs, _ = _single_bottleneck(case, "gpu", torch.bfloat16)
    _dump_rows(f"parity_bf16_bottleneck_{case}.json", {"bound": "1.5 x torch.autocast(bfloat16) of the reference block, per tensor",
                                                     "worst": max(rows, key=lambda r: r[1] / max(r[2], 1e-30))}, rows)
    assert rows[0][0] == "output" and rows[0][1] < 1e-2, rows[0]
    cal_med = sorted(r[2] for r in rows[1:])[len(rows[1:]) // 2]
    assert 1e-2 < cal_med < 0.3, ("the calibration is not a usable signal", cal_med)
    for n, mine, cal in rows[1:]:
        assert mine <= 1.5 * max(cal, cal_med), (case, n, mine, cal, cal_med)
    for n, e in bufs:
        assert e < 1e-2, (case, n, e)


@pytest.mark.parametrize("backend", BACKENDS)
def test_bn3_backward_folded_into_conv3_in_the_block_schedule(backend, monkeypatch):
    """Two identity Bottlenecks of stage 2 in bf16 through the backbone's own schedule: the first one's bn3 backward arrives with
    its sums from the second one's input-gradient epilogue and is FOLDED into conv3's weights (csrc/bn_fold.hip,
    _backward_blocks::conv3_back_folded) instead of applied in a pass.  Both schedules against the fp64 oracle: the folded one
    may not be further away than the pass form beyond bf16 rounding, and every gradient of the two agrees to bf16 accuracy."""
    import copy
    from virtex_amd import ops
    from virtex_amd.modules import visual_backbones as vbm
    dev = select(backend)
    B, H = (2, 8) if backend == "emu" else (16, 28)
    oracle_model = synth.seeded_model(port.build_model, seed=0, dropout=0.0, textual="transdec_postnorm::L1_H128_A2_F256",
                                      vocab_size=304).train()
    model = vf.build_bicaptioning_model(textual="transdec_postnorm::L1_H128_A2_F256", vocab_size=304, dropout=0.0,
                                        compute_dtype=torch.bfloat16)
    model.load_state_dict(oracle_model.state_dict())
    model = model.to(dev).train()
    g = torch.Generator().manual_seed(5)
    x = torch.relu(torch.randn(B, 512, H, H, generator=g))
    up = torch.randn(B, 512, H, H, generator=g)
    o64 = copy.deepcopy(oracle_model.visual.cnn.layer2[1:3]).double().train()
    x64 = x.double().requires_grad_(True)
    o64(x64).backward(up.double())
    ref = {"input_gradient": x64.grad}
    ref.update({n: p.grad for n, p in o64.named_parameters()})
    calls = []
    real = ops.bn_bwd_fold
    monkeypatch.setattr(ops, "bn_bwd_fold", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    state = copy.deepcopy(model.state_dict())

    def run(fold):
        monkeypatch.setattr(vbm, "FUSE_BN3_FOLD", fold)
        model.load_state_dict(state)                                    # the same BatchNorm buffers for both runs
        model.zero_grad(set_to_none=True)
        xd = x.to(dev).requires_grad_(True)
        model.visual.forward_blocks(xd, 2, 1, 2).backward(up.to(dev))
        out = {"input_gradient": xd.grad.float().cpu()}
        for i in (1, 2):
            out.update({f"{i}.{n}": p.grad.float().cpu() for n, p in model.visual.cnn.layer2[i].named_parameters()})
        return out
    plain = run(False)
    assert not calls
    monkeypatch.setattr(ops.splitk_batch, "enabled", True)              # ... and inside a reduction batch: the fold READS two split-K results
    folded = run(True)
    assert len(calls) == 1                                              # the first block's bn3; the last block has no sums to fold
    worst = ("", 0.0)
    for n, r in ref.items():
        e_plain, e_fold = rel_err(plain[n], r), rel_err(folded[n], r)
        assert e_fold <= max(1.5 * e_plain, 2e-2), (n, e_plain, e_fold)
        assert rel_err(folded[n], plain[n]) < 5e-2, (n, rel_err(folded[n], plain[n]))
        if e_fold / max(e_plain, 1e-30) > worst[1]:
            worst = (n, e_fold / max(e_plain, 1e-30))
    if backend == "gpu":
        _dump_rows("parity_bf16_bn3_fold_stage2.json", {"worst_ratio_folded_over_pass": worst},
                   [(n, rel_err(plain[n], r), rel_err(folded[n], r)) for n, r in ref.items()])


@pytest.mark.gpu
def test_batches_beyond_2_24_pixels_gpu():
    """The envelope of rounds 1-5 ended at 2^24 pixels per tensor (334 images of 224 x 224: the pixel-index divisions were exact
    below 2^24 only).  With vtx_fdiv30 it ends at 2^30.  B = 384 at 224 x 224 (20.3 M pixels in the stem's haloed input):
    (1) eval mode is per-sample independent (BatchNorm folded into the convolutions), so the features of the whole batch must
    equal the features of its two halves -- every forward kernel, images whose pixel indices lie above 2^24 included;
    (2) one bf16 training step against the fp32 step of the same weights and batch (virtex_amd.fidelity), bounds of the B = 256
    test; (3) the last images of the batch contribute to the gradient: zeroing them changes it."""
    from virtex_amd import fidelity
    dev = select("gpu")
    B = 384
    om = synth.seeded_model(port.build_model, seed=0, dropout=0.0, randomize=False)
    batch = {k: v.to(dev) for k, v in synth.synthetic_batch(B, image_size=224, seed=3).items()}
    for dtype in (torch.float32, torch.bfloat16):
        m = vf.build_bicaptioning_model(dropout=0.0, compute_dtype=dtype)
        m.load_state_dict(om.state_dict())
        m = m.to(dev).eval()
        with torch.no_grad():
            whole = m.visual(batch["image"]).float()
            halves = torch.cat([m.visual(batch["image"][: B // 2]), m.visual(batch["image"][B // 2:])]).float()
        assert torch.isfinite(whole).all()
        assert rel_err(whole.cpu(), halves.cpu()) < (1e-6 if dtype == torch.float32 else 1e-3), dtype
        assert rel_err(whole[-8:].cpu(), halves[-8:].cpu()) < (1e-6 if dtype == torch.float32 else 1e-3)      # images above 2^24 pixels
        del m
    model = vf.build_bicaptioning_model(dropout=0.0, compute_dtype=torch.bfloat16)
    model.load_state_dict(om.state_dict())
    model = model.to(dev).train()
    s = fidelity.bf16_vs_fp32(model, batch)
    _dump_rows("fidelity_bf16_b384_beyond_2_24_pixels.json", s, [])
    assert s["loss_rel"] < 2e-4
    assert s["backbone"]["median_rel"] <= 0.25 and s["backbone"]["max_rel"] <= 0.35 and s["backbone"]["min_cos"] >= 0.95, s
    assert s["text"]["max_rel"] <= 6e-2 and s["text"]["min_cos"] >= 0.995, s
    _, g_all = fidelity.run_grads(model, batch)
    cut = dict(batch, image=batch["image"].clone())
    cut["image"][-16:] = 0.0
    _, g_cut = fidelity.run_grads(model, cut)
    n = "visual.cnn.conv1.weight"
    assert rel_err(g_cut[n].cpu(), g_all[n].cpu()) > 1e-3
