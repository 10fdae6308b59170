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


def test_factories_mirror_reference_semantics_and_register():
    with pytest.raises(KeyError):
        vf.VisualBackboneFactory.create("does-not-exist")
    with pytest.raises(ValueError):
        vf.PretrainingModelFactory()
    assert vf.parse_textual_architecture("L4_H1024_A16_F4096") == dict(
        num_layers=4, hidden_size=1024, attention_heads=16, feedforward_size=4096)
    # a stand-in for the reference's `virtex.factories` module (needs fvcore, absent here)
    fake = types.SimpleNamespace(
        VisualBackboneFactory=type("VisualBackboneFactory", (), {"PRODUCTS": {"torchvision": object}}),
        TextualHeadFactory=type("TextualHeadFactory", (), {"PRODUCTS": {"transdec_prenorm": object, "none": object}}),
        PretrainingModelFactory=type("PretrainingModelFactory", (), {"PRODUCTS": {"masked_lm": object}}),
        CaptionDecoderFactory=type("CaptionDecoderFactory", (), {"PRODUCTS": {"beam_search": object, "nucleus_sampling": object}}))
    replaced = vf.register(fake)
    assert fake.VisualBackboneFactory.PRODUCTS["torchvision"] is vf.VisualBackboneFactory.PRODUCTS["torchvision"]
    assert "transdec_postnorm" in fake.TextualHeadFactory.PRODUCTS and "none" in fake.TextualHeadFactory.PRODUCTS
    assert set(fake.PretrainingModelFactory.PRODUCTS) == {"masked_lm", "virtex", "bicaptioning", "captioning"}
    assert fake.CaptionDecoderFactory.PRODUCTS["beam_search"] is vf.decoding.AutoRegressiveBeamSearch
    assert len(replaced) == 7


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
