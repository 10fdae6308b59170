"""Factory registries mirroring the reference's plugin seam
(/root/reference/virtex/factories.py:40-78 base class; VisualBackboneFactory :306-341,
TextualHeadFactory :344-407, PretrainingModelFactory :410-466).

``Factory.create(name, *args, **kwargs)`` looks a callable up in ``PRODUCTS`` (``KeyError`` on
unknown names, ``ValueError`` on instantiation) exactly like the reference.  ``register()``
assigns the MI355X-native products into the *reference's* registries when that package is
importable, so ``scripts/pretrain_virtex.py`` / ``virtex.model_zoo.get`` build the native
modules unmodified.
"""
import re
from functools import partial
from typing import Any, Callable, Dict, List

import torch

from . import decoding, models
from .modules import textual_heads, visual_backbones


class Factory:
    PRODUCTS: Dict[str, Callable] = {}

    def __init__(self):
        raise ValueError(f"Cannot instantiate {self.__class__.__name__} object, use `create` classmethod.")

    @classmethod
    def create(cls, name: str, *args, **kwargs) -> Any:
        if name not in cls.PRODUCTS:
            raise KeyError(f"{cls.__class__.__name__} cannot create {name}.")
        return cls.PRODUCTS[name](*args, **kwargs)

    @classmethod
    def from_config(cls, config) -> Any:
        raise NotImplementedError


class VisualBackboneFactory(Factory):
    PRODUCTS: Dict[str, Callable] = {"torchvision": visual_backbones.TorchvisionVisualBackbone}

    @classmethod
    def from_config(cls, config) -> visual_backbones.VisualBackbone:
        _C = config
        kwargs = {"visual_feature_size": _C.MODEL.VISUAL.FEATURE_SIZE}
        if "torchvision" in _C.MODEL.VISUAL.NAME:
            zoo_name, cnn_name = _C.MODEL.VISUAL.NAME.split("::")
            kwargs["pretrained"] = _C.MODEL.VISUAL.PRETRAINED
            kwargs["frozen"] = _C.MODEL.VISUAL.FROZEN
            return cls.create(zoo_name, cnn_name, **kwargs)
        return cls.create(_C.MODEL.VISUAL.NAME, **kwargs)


class TextualHeadFactory(Factory):
    PRODUCTS: Dict[str, Callable] = {
        "transdec_prenorm": partial(textual_heads.TransformerDecoderTextualHead, norm_first=True),
        "transdec_postnorm": partial(textual_heads.TransformerDecoderTextualHead, norm_first=False),
    }

    @classmethod
    def from_config(cls, config) -> textual_heads.TextualHead:
        _C = config
        name = _C.MODEL.TEXTUAL.NAME
        kwargs = {"visual_feature_size": _C.MODEL.VISUAL.FEATURE_SIZE, "vocab_size": _C.DATA.VOCAB_SIZE}
        if "trans" in _C.MODEL.TEXTUAL.NAME:
            name, architecture = name.split("::")
            kwargs.update(parse_textual_architecture(architecture))
            kwargs.update(dropout=_C.MODEL.TEXTUAL.DROPOUT,
                          mask_future_positions="captioning" in _C.MODEL.NAME or _C.MODEL.NAME == "virtex",
                          max_caption_length=_C.DATA.MAX_CAPTION_LENGTH, padding_idx=_C.DATA.UNK_INDEX)
        return cls.create(name, **kwargs)


class PretrainingModelFactory(Factory):
    PRODUCTS: Dict[str, Callable] = {
        "virtex": models.VirTexModel,
        "bicaptioning": models.BidirectionalCaptioningModel,
        "captioning": models.ForwardCaptioningModel,
    }

    @classmethod
    def from_config(cls, config):
        _C = config
        visual = VisualBackboneFactory.from_config(_C)
        textual = TextualHeadFactory.from_config(_C)
        kwargs = {"sos_index": _C.DATA.SOS_INDEX, "eos_index": _C.DATA.EOS_INDEX,
                  "decoder": CaptionDecoderFactory.from_config(_C)}
        return cls.create(_C.MODEL.NAME, visual, textual, **kwargs)


class CaptionDecoderFactory(Factory):
    """``{"beam_search", "nucleus_sampling"}`` (reference: factories.py:469-500)."""
    PRODUCTS: Dict[str, Callable] = {
        "beam_search": decoding.AutoRegressiveBeamSearch,
        "nucleus_sampling": decoding.AutoRegressiveNucleusSampling,
    }

    @classmethod
    def from_config(cls, config):
        _C = config
        kwargs = {"eos_index": _C.DATA.EOS_INDEX, "max_steps": _C.MODEL.DECODER.MAX_DECODING_STEPS}
        if _C.MODEL.DECODER.NAME == "beam_search":
            kwargs["beam_size"] = _C.MODEL.DECODER.BEAM_SIZE
        elif _C.MODEL.DECODER.NAME == "nucleus_sampling":
            kwargs["nucleus_size"] = _C.MODEL.DECODER.NUCLEUS_SIZE
        return cls.create(_C.MODEL.DECODER.NAME, **kwargs)


def parse_textual_architecture(architecture: str) -> Dict[str, int]:
    """'L1_H1024_A16_F4096' -> kwargs (reference regex: factories.py:387)."""
    m = re.match(r"L(\d+)_H(\d+)_A(\d+)_F(\d+)", architecture)
    if m is None:
        raise ValueError(f"cannot parse textual architecture {architecture!r}")
    L, H, A, F = (int(g) for g in m.groups())
    return {"num_layers": L, "hidden_size": H, "attention_heads": A, "feedforward_size": F}


def build_bicaptioning_model(visual: str = "torchvision::resnet50",
                             textual: str = "transdec_postnorm::L1_H1024_A16_F4096",
                             vocab_size: int = 10000, dropout: float = 0.1, max_caption_length: int = 30,
                             compute_dtype: torch.dtype = torch.bfloat16, model_name: str = "virtex"):
    """The defaults of virtex/config.py without fvcore."""
    zoo, cnn = visual.split("::")
    vb = VisualBackboneFactory.create(zoo, cnn, visual_feature_size=2048, compute_dtype=compute_dtype)
    vb.visual_feature_size = vb.cnn.out_channels
    name, arch = textual.split("::")
    th = TextualHeadFactory.create(name, visual_feature_size=vb.visual_feature_size, vocab_size=vocab_size,
                                   dropout=dropout, mask_future_positions=True,
                                   max_caption_length=max_caption_length, padding_idx=0,
                                   compute_dtype=compute_dtype, **parse_textual_architecture(arch))
    return PretrainingModelFactory.create(model_name, vb, th, sos_index=1, eos_index=2, decoder=None)


def register(reference_factories=None) -> List[str]:
    """Install the native products into the reference's registries (drop-in).  Pass the imported
    ``virtex.factories`` module, or let it be imported.  Returns the keys that were replaced."""
    if reference_factories is None:
        import virtex.factories as reference_factories  # noqa: the reference package
    replaced = []
    for ours, theirs in ((VisualBackboneFactory, reference_factories.VisualBackboneFactory),
                         (TextualHeadFactory, reference_factories.TextualHeadFactory),
                         (PretrainingModelFactory, reference_factories.PretrainingModelFactory),
                         (CaptionDecoderFactory, reference_factories.CaptionDecoderFactory)):
        for key, product in ours.PRODUCTS.items():
            theirs.PRODUCTS[key] = product
            replaced.append(f"{theirs.__name__}.{key}")
    return replaced
