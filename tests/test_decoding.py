"""Caption decoders: virtex_amd.decoding against goldens generated from the reference's
own AutoRegressiveBeamSearch / AutoRegressiveNucleusSampling (oracle/make_decoding_goldens.py), against the live
reference classes when /root/reference is present, and end to end through the HIP text head."""
import json
import os
import warnings

import pytest
import torch

from backends import select
from oracle import decoding_cases as dc
from oracle import reference_import
from virtex_amd.decoding import AutoRegressiveBeamSearch, AutoRegressiveNucleusSampling

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "decoding.json")


@pytest.mark.parametrize("case", sorted(dc.CASES))
def test_decoders_reproduce_reference_goldens(case):
    with open(GOLDEN) as f:
        gold = json.load(f)[case]
    tokens, lp = dc.run(AutoRegressiveBeamSearch, AutoRegressiveNucleusSampling, case)
    assert tokens.dtype == torch.long
    assert tokens.tolist() == gold["tokens"]                    # integer work: exact
    if gold["logprobs"] is not None:
        assert torch.allclose(lp, torch.tensor(gold["logprobs"]), rtol=1e-6, atol=1e-6)


@pytest.mark.reference
@pytest.mark.parametrize("case", sorted(dc.CASES))
def test_decoders_equal_live_reference(case):
    reference_import.import_reference()
    from virtex.utils.beam_search import AutoRegressiveBeamSearch as RefBeam
    from virtex.utils.nucleus_sampling import AutoRegressiveNucleusSampling as RefNucleus
    t_ref, lp_ref = dc.run(RefBeam, RefNucleus, case)
    t, lp = dc.run(AutoRegressiveBeamSearch, AutoRegressiveNucleusSampling, case)
    assert torch.equal(t, t_ref)
    if lp_ref is not None:
        # floating point: the summation order of log-softmax depends on the thread count earlier tests leave behind
        torch.testing.assert_close(lp, lp_ref, rtol=1e-6, atol=1e-6)


def test_beam_search_edge_cases():
    V, EOS = 11, 2

    def always_eos(partial):
        n = partial.size(0)
        out = torch.full((n, V), -5.0)
        out[:, EOS] = 5.0
        return out
    start = torch.ones(3, dtype=torch.long)
    with pytest.warns(RuntimeWarning, match="Empty captions"):      # beam 1 and nothing but EOS (beam_search.py:106-113)
        toks, lp = AutoRegressiveBeamSearch(EOS, max_steps=5, beam_size=1).search(start, always_eos)
    assert toks.shape == (3, 1, 1) and (toks == EOS).all()       # the reference returns the (B, beam, 1) tensor here

    def two_valid(partial):                                         # fewer valid continuations than the beam is wide
        n = partial.size(0)
        out = torch.full((n, V), float("-inf"))
        out[:, 3] = 0.0
        out[:, EOS] = -1.0
        return out
    with pytest.warns(RuntimeWarning, match="Infinite log probs"):
        toks, lp = AutoRegressiveBeamSearch(EOS, max_steps=1, beam_size=5, per_node_beam_size=2).search(
            start, two_valid, only_return_best=False)
    assert toks.shape[:2] == (3, 5)
    # a finished beam keeps emitting EOS and its score stops changing
    toks, lp = AutoRegressiveBeamSearch(EOS, max_steps=6, beam_size=2, per_node_beam_size=2).search(start, dc.make_step(7))
    for row in toks.tolist():
        if EOS in row:
            assert all(t == EOS for t in row[row.index(EOS):])


def _decode_pair(dev, dtype, decoder):
    from oracle import bicaptioning as port, synth
    import virtex_amd.factories as vf
    textual = "transdec_postnorm::L1_H128_A2_F256"
    oracle_model = synth.seeded_model(port.build_model, seed=0, dropout=0.0, textual=textual, vocab_size=500)
    model = vf.build_bicaptioning_model(textual=textual, vocab_size=500, dropout=0.0, compute_dtype=dtype)
    model.load_state_dict(oracle_model.state_dict())
    model = model.to(dev).eval()
    oracle_model.eval()
    oracle_model.decoder = decoder
    model.decoder = decoder
    image = synth.synthetic_batch(batch_size=2, image_size=64, max_len=8, vocab_size=500, seed=5)["image"]
    with torch.no_grad():
        ref = oracle_model({"image": image})["predictions"]
        out = model({"image": image.to(dev)})["predictions"]
    return ref, out.cpu()


@pytest.mark.emu
def test_inference_branch_beam_search_through_hip_head_emulator():
    """model({"image": ...}) in eval mode: backbone (folded BN) -> decoding_step on growing prefixes -> beam search
    (captioning.py:144-162).  fp32: the HIP head's logits agree with the oracle's to ~1e-5, so do the beams."""
    ref, out = _decode_pair(select("emu"), torch.float32, AutoRegressiveBeamSearch(eos_index=2, max_steps=6, beam_size=3))
    assert out.dtype == torch.long and out.shape == ref.shape
    assert (out == ref).float().mean().item() >= 0.9


@pytest.mark.gpu
def test_inference_branch_beam_search_through_hip_head_gpu():
    ref, out = _decode_pair(select("gpu"), torch.float32, AutoRegressiveBeamSearch(eos_index=2, max_steps=10, beam_size=5))
    assert out.shape == ref.shape and (out == ref).float().mean().item() >= 0.9
    ref, out = _decode_pair(select("gpu"), torch.bfloat16, AutoRegressiveNucleusSampling(eos_index=2, max_steps=10))
    assert out.dim() == 2 and out.shape[0] == 2 and int(out.max()) < 500 and int(out.min()) >= 0


def test_missing_decoder_raises_like_the_reference():
    import virtex_amd.factories as vf
    select("emu")            # the forward up to the missing decoder runs kernels: do not depend on an earlier test's choice
    model = vf.build_bicaptioning_model(textual="transdec_postnorm::L1_H128_A2_F256", vocab_size=500).eval()
    model.visual.forward = lambda image: torch.zeros(image.size(0), 2048, 2, 2)
    with pytest.raises(ValueError, match="Decoder for predicting captions is missing"):
        model({"image": torch.zeros(1, 3, 64, 64)})


from backends import BACKENDS  # noqa: E402


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [c for c in sorted(dc.CASES) if dc.CASES[c][0].startswith("beam")])
def test_device_beam_step_reproduces_reference_goldens(backend, case):
    """The same goldens (reference's own AutoRegressiveBeamSearch on a deterministic step function) with every step's
    selection done by vtx_beam_step: tokens exact, log-probabilities to fp32 rounding."""
    dev = select(backend)
    with open(GOLDEN) as f:
        gold = json.load(f)[case]
    kind, kw, B, seed = dc.CASES[case]
    step_cpu = dc.make_step(seed)
    calls = []

    def step(partial):
        out = step_cpu(partial.cpu()).to(dev)
        calls.append(out.device.type)
        return out
    start = torch.full((B,), dc.SOS, dtype=torch.long, device=dev)
    tokens, lp = AutoRegressiveBeamSearch(**kw).search(start, step, only_return_best=(kind == "beam"))
    assert tokens.cpu().tolist() == gold["tokens"]
    assert torch.allclose(lp.cpu(), torch.tensor(gold["logprobs"]), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["transdec_postnorm", "transdec_prenorm"])
def test_incremental_decoding_step_equals_full_prefix_recompute(backend, dtype, kind):
    """The KV-cached step (one token per call) against `model.decoding_step` (the reference's semantics: the whole
    prefix through the head at every call) on the SAME model: logits of every step of a beam search with re-ordered,
    duplicated and dropped beams, driven by the prefixes alone (no `reorder` hint -- what a foreign decoder gives)."""
    from oracle import synth
    import virtex_amd.factories as vf
    from virtex_amd.decoding import IncrementalDecodingStep
    dev = select(backend)
    torch.manual_seed(3)
    model = vf.build_bicaptioning_model(textual=kind + "::L2_H128_A2_F256", vocab_size=300, dropout=0.0,
                                        compute_dtype=dtype, max_caption_length=12).to(dev).eval()
    synth.randomize_state(model, 7)
    feats = torch.randn(3, 2048, 2, 2, device=dev).to(dtype if dtype == torch.bfloat16 else torch.float32)
    feats = feats.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)          # NHWC-physical like the backbone's output
    step = IncrementalDecodingStep(model.textual, feats)
    g = torch.Generator().manual_seed(5)
    B, W = 3, 4
    prefix = torch.full((B,), 1, dtype=torch.long)
    tol = 2e-4 if dtype == torch.float32 else 6e-2
    with torch.no_grad():
        got, ref = step(prefix.to(dev)), model.decoding_step(feats, prefix.to(dev))
        assert rel_err(got.cpu(), ref.float().cpu()) < tol
        beams = torch.cat([prefix.view(B, 1, 1).expand(B, W, 1), torch.randint(4, 300, (B, W, 1), generator=g)], -1)
        for t in range(2, 9):
            flat = beams.reshape(B * W, -1)
            got, ref = step(flat.to(dev)), model.decoding_step(feats, flat.to(dev))
            assert got.shape == ref.shape == (B * W, 300)
            assert rel_err(got.cpu(), ref.float().cpu()) < tol, t
            parent = torch.randint(0, W, (B, W), generator=g)                    # beams continue arbitrary beams of their image
            beams = torch.cat([beams.gather(1, parent.unsqueeze(-1).expand(B, W, beams.size(-1))),
                               torch.randint(4, 300, (B, W, 1), generator=g)], -1)


def rel_err(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
