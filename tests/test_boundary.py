"""The drop-in boundary: C-ABI surface, loud failure without the HIP extension, factory registration."""
import ctypes
import os
import re
import types

import pytest
import torch

from virtex_amd import _lib, build
import virtex_amd.factories as vf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    header = open(os.path.join(ROOT, "include", "virtex_amd.h")).read()
    return sorted(set(re.findall(r"\b(vtx_[a-z0-9_]+)\s*\(", header)))


def test_hip_library_exports_every_declared_symbol():
    """No compute calls here (no GPU): the gfx950 library must load and export the whole header."""
    path = build.build_hip()
    lib = ctypes.CDLL(path)
    missing = [s for s in _header_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    lib.vtx_backend.restype = ctypes.c_char_p
    assert lib.vtx_backend() == b"hip:gfx950"
    assert len(_header_symbols()) >= 29


def test_no_cpu_fallback_in_product_build():
    """CPU tensors handed to the HIP build must raise, never silently run somewhere else."""
    _lib.use_library(build.build_hip())
    from virtex_amd import ops

    x = torch.zeros(4, 64)
    with pytest.raises(_lib.VtxError):
        ops.layernorm_residual_fwd(x, None, torch.ones(64), torch.zeros(64), 1e-5)
    with pytest.raises(_lib.VtxError):
        _lib.use_library("/nonexistent/libvirtex_amd.so")


def test_error_reporting_through_the_abi():
    from backends import select
    from virtex_amd import ops

    select("emu")
    with pytest.raises(_lib.VtxError, match="multiple"):
        ops.layernorm_residual_fwd(torch.zeros(2, 30), None, torch.ones(30), torch.zeros(30), 1e-5)


def test_factories_mirror_reference_semantics():
    with pytest.raises(KeyError):
        vf.VisualBackboneFactory.create("does-not-exist")
    with pytest.raises(ValueError):
        vf.PretrainingModelFactory()
    assert vf.parse_textual_architecture("L4_H1024_A16_F4096") == dict(
        num_layers=4, hidden_size=1024, attention_heads=16, feedforward_size=4096)


REF_BASE_YAML = "/root/reference/configs/_base_bicaptioning_R_50_L1_H1024.yaml"


@pytest.fixture
def reference_registries():
    """The reference's OWN `virtex.factories` / `virtex.config.Config`, imported verbatim from /root/reference
    (fvcore / loguru / albumentations / cv2 / torchvision stubbed: oracle/reference_import.py), with
    `virtex_amd.factories.register()` applied; the registries are restored afterwards."""
    from oracle import reference_import

    ref_f, Config = reference_import.import_reference_factories()
    facs = (ref_f.VisualBackboneFactory, ref_f.TextualHeadFactory, ref_f.PretrainingModelFactory, ref_f.CaptionDecoderFactory)
    saved = [dict(f.PRODUCTS) for f in facs]
    replaced = vf.register(ref_f)
    try:
        yield ref_f, Config, replaced, saved
    finally:
        for f, d in zip(facs, saved):
            f.PRODUCTS.clear()
            f.PRODUCTS.update(d)


@pytest.mark.reference
def test_register_installs_native_products_in_the_reference_registries(reference_registries):
    """INTEGRATION.md section 1 on the real module: after `register(virtex.factories)` the reference's own
    `PretrainingModelFactory.from_config` (factories.py:428-466) builds native modules from the reference's own
    Config + shipped YAML, its `OptimizerFactory` (:529-545) sees the parameter names it groups by, and products we
    do not provide stay the reference's."""
    from virtex_amd import decoding, models
    from virtex_amd.modules import textual_heads, visual_backbones

    ref_f, Config, replaced, saved = reference_registries
    assert len(replaced) == 8
    assert ref_f.VisualBackboneFactory.PRODUCTS["torchvision"] is visual_backbones.TorchvisionVisualBackbone
    assert set(ref_f.TextualHeadFactory.PRODUCTS) == {"transdec_prenorm", "transdec_postnorm", "none"}
    assert ref_f.TextualHeadFactory.PRODUCTS["transdec_prenorm"].func is textual_heads.TransformerDecoderTextualHead
    assert ref_f.TextualHeadFactory.PRODUCTS["transdec_prenorm"].keywords == {"norm_first": True}
    assert ref_f.TextualHeadFactory.PRODUCTS["none"] is saved[1]["none"]                              # untouched
    assert ref_f.PretrainingModelFactory.PRODUCTS["masked_lm"] is saved[2]["masked_lm"]
    assert ref_f.CaptionDecoderFactory.PRODUCTS["beam_search"] is decoding.AutoRegressiveBeamSearch

    _C = Config(REF_BASE_YAML, [])                                # the shipped R_50_L1_H1024 configuration
    model = ref_f.PretrainingModelFactory.from_config(_C)        # the REFERENCE's factory code runs here
    assert type(model) is models.VirTexModel
    assert type(model.visual) is visual_backbones.TorchvisionVisualBackbone
    assert type(model.textual) is textual_heads.TransformerDecoderTextualHead
    assert isinstance(model.decoder, decoding.AutoRegressiveBeamSearch) and model.decoder.beam_size == _C.MODEL.DECODER.BEAM_SIZE
    assert model.textual.mask_future_positions and model.textual.dropout == _C.MODEL.TEXTUAL.DROPOUT
    assert len(model.state_dict()) == 370 and sum(p.numel() for p in model.parameters()) == 69482320
    # the reference's optimizer factory classifies parameters BY NAME: 26 no-decay tensors, 159 at CNN_LR
    optimizer = ref_f.OptimizerFactory.from_config(_C, model.named_parameters())
    groups = optimizer.param_groups
    assert len(groups) == 202
    assert sum(1 for g in groups if g["weight_decay"] == 0.0) == 26
    assert sum(1 for g in groups if g["lr"] == _C.OPTIM.CNN_LR) == 159
    assert type(optimizer).__name__ == "Lookahead"
    scheduler = ref_f.LRSchedulerFactory.from_config(_C, optimizer)
    assert type(scheduler).__name__ == "LinearWarmupCosineAnnealingLR"
    # unknown names still fail the reference's way
    with pytest.raises(KeyError):
        ref_f.VisualBackboneFactory.create("does-not-exist")


@pytest.mark.reference
@pytest.mark.emu
def test_reference_training_loop_drives_native_modules(reference_registries):
    """The loop body of scripts/pretrain_virtex.py:145-163, restated line for line, around a model built by the
    reference's factories after register(): reference Config, reference OptimizerFactory (Lookahead(SGD)),
    reference LRSchedulerFactory, `amp.autocast` + `amp.GradScaler` exactly as the script creates them (on a
    CUDA-less host both disable themselves, which IS the reference's CPU path), `clip_grad_norm_`.  Three steps;
    every step's loss must equal the oracle's TrainStep on the same batches (the 2nd and 3rd depend on the updates)."""
    from backends import select
    from oracle import bicaptioning as port, synth
    from torch.cuda import amp

    dev = select("emu")
    ref_f, Config, _, _ = reference_registries
    over = ["MODEL.TEXTUAL.NAME", "transdec_postnorm::L1_H128_A2_F256", "DATA.VOCAB_SIZE", 1000, "MODEL.TEXTUAL.DROPOUT", 0.0,
            "DATA.MAX_CAPTION_LENGTH", 12, "OPTIM.WARMUP_STEPS", 200, "OPTIM.NUM_ITERATIONS", 2000, "OPTIM.LOOKAHEAD.STEPS", 2]
    _C = Config(REF_BASE_YAML, over)
    oracle_model = synth.seeded_model(port.build_model, seed=0, dropout=0.0, textual=_C.MODEL.TEXTUAL.NAME, vocab_size=1000,
                                      max_caption_length=12)
    model = ref_f.PretrainingModelFactory.from_config(_C)
    for m in (model.visual, model.textual, model.backward_textual):
        m.compute_dtype = torch.float32                       # parity mode of the native modules
    model.load_state_dict(oracle_model.state_dict())
    model = model.to(dev).train()
    optimizer = ref_f.OptimizerFactory.from_config(_C, model.named_parameters())
    scheduler = ref_f.LRSchedulerFactory.from_config(_C, optimizer)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        scaler = amp.GradScaler(enabled=_C.AMP)
    oracle_step = port.TrainStep(oracle_model.train(), clip=_C.OPTIM.CLIP_GRAD_NORM, k=_C.OPTIM.LOOKAHEAD.STEPS,
                                 alpha=_C.OPTIM.LOOKAHEAD.ALPHA, total_steps=_C.OPTIM.NUM_ITERATIONS,
                                 warmup_steps=_C.OPTIM.WARMUP_STEPS)
    for it in range(3):
        # 4 images of 128x128: the smallest case whose backbone gradients are conditioned (36 samples per channel in
        # the last stage); the long warm-up keeps the CNN_LR = 0.2 steps small enough to stay in that regime
        batch = synth.synthetic_batch(4, image_size=128, max_len=12, vocab_size=1000, seed=40 + it, ragged=True)
        expect = oracle_step(batch).item()
        # ---- scripts/pretrain_virtex.py:147-162
        optimizer.zero_grad()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with amp.autocast(enabled=_C.AMP):
                output_dict = model({k: v.to(dev) for k, v in batch.items()})
                loss = output_dict["loss"]
        scaler.scale(loss).backward()
        scaler.unscale_(optimizer)
        torch.nn.utils.clip_grad_norm_(model.parameters(), _C.OPTIM.CLIP_GRAD_NORM)
        scaler.step(optimizer)
        scaler.update()
        scheduler.step()
        assert abs(loss.item() - expect) < 2e-4 * abs(expect), (it, loss.item(), expect)
    from backends import rel_err
    for (n, p), (_, q) in zip(model.named_parameters(), oracle_model.named_parameters()):
        # relative L2 of the whole tensor (the backbone's gradients are only conditioned to ~1e-2 at 2 images of
        # 64x64; three clipped steps move a weight by far less than that of its norm)
        assert rel_err(p.detach().cpu(), q.detach()) < 5e-3, n


def test_from_config_builds_the_reference_default_model():
    ns = types.SimpleNamespace
    cfg = ns(MODEL=ns(NAME="virtex", VISUAL=ns(NAME="torchvision::resnet50", FEATURE_SIZE=2048, PRETRAINED=False,
                                               FROZEN=False),
                      TEXTUAL=ns(NAME="transdec_postnorm::L1_H1024_A16_F4096", DROPOUT=0.1)),
             DATA=ns(VOCAB_SIZE=10000, MAX_CAPTION_LENGTH=30, UNK_INDEX=0, SOS_INDEX=1, EOS_INDEX=2))
    cfg.MODEL.DECODER = ns(NAME="beam_search", BEAM_SIZE=5, NUCLEUS_SIZE=0.9, MAX_DECODING_STEPS=50)   # config.py defaults
    model = vf.PretrainingModelFactory.from_config(cfg)
    assert isinstance(model.decoder, vf.decoding.AutoRegressiveBeamSearch) and model.decoder.beam_size == 5
    cfg.MODEL.DECODER.NAME = "nucleus_sampling"
    assert vf.CaptionDecoderFactory.from_config(cfg).nucleus_size == 0.9
    assert sum(p.numel() for p in model.parameters()) == 69482320
    assert model.textual.mask_future_positions and model.backward_textual.embedding is model.textual.embedding


def test_detectron2_backbone_state_dict_names():
    """Reference: visual_backbones.py:76-120 (substring renames applied in dict order).  The expectation below
    re-derives the names with the reference's rule so the two implementations are checked against each other."""
    from virtex_amd.modules import TorchvisionVisualBackbone
    vb = TorchvisionVisualBackbone("resnet50")
    d2 = vb.detectron2_backbone_state_dict()
    assert set(d2) == {"model", "__author__", "matching_heuristics"} and d2["matching_heuristics"] is True
    rule = [("layer1", "res2"), ("layer2", "res3"), ("layer3", "res4"), ("layer4", "res5"), ("bn1", "conv1.norm"),
            ("bn2", "conv2.norm"), ("bn3", "conv3.norm"), ("downsample.0", "shortcut"), ("downsample.1", "shortcut.norm")]
    expect = {}
    for k, v in vb.cnn.state_dict().items():
        n = k
        for a, b in rule:
            n = n.replace(a, b)
        expect[n if n.startswith("res") else "stem." + n] = v
    assert list(d2["model"]) == list(expect)
    for k in expect:
        assert d2["model"][k].data_ptr() == expect[k].data_ptr()
    assert "stem.conv1.weight" in d2["model"] and "stem.conv1.norm.running_var" in d2["model"]
    assert "res2.0.shortcut.norm.weight" in d2["model"] and "res5.2.conv3.norm.bias" in d2["model"]
    assert d2["model"]["res3.0.conv2.weight"].shape == (128, 128, 3, 3)
