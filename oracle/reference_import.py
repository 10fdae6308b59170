"""Import the reference's own classes verbatim from /root/reference (build container only).

TEST INFRASTRUCTURE.  The reference needs torchvision / albumentations / cv2, none of
which is installed here.  We register minimal ``sys.modules`` stand-ins:

* ``torchvision.models.<name>(pretrained, zero_init_residual=...)`` -> the torch-only
  ResNet restatement of ``oracle/bicaptioning.py`` (torchvision is third-party and not
  vendored by the reference, so its graph can only be restated -- Appendix A.1);
* ``albumentations`` / ``cv2``: inert placeholders, needed only because
  ``virtex/models/captioning.py:8`` imports ``virtex.data`` whose transforms subclass
  albumentations classes at import time.  No data-pipeline code runs.

Nothing here is reachable from the GPU box (``/root/reference`` does not exist there);
``available()`` lets tests skip.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("VIRTEX_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "virtex", "models"))


def _install_stubs():
    from oracle import bicaptioning as port

    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        models = types.ModuleType("torchvision.models")
        for name in port.RESNET_SPECS:
            def ctor(pretrained=False, zero_init_residual=False, _n=name, **kw):
                assert not pretrained
                return port.ResNet(_n, zero_init_residual=zero_init_residual)
            setattr(models, name, ctor)
        datasets = types.ModuleType("torchvision.datasets")
        datasets.ImageNet = type("ImageNet", (), {})
        tv.models, tv.datasets = models, datasets
        sys.modules.update({"torchvision": tv, "torchvision.models": models,
                            "torchvision.datasets": datasets})
    if "albumentations" not in sys.modules:
        alb = types.ModuleType("albumentations")

        class _T:
            def __init__(self, *a, **k):
                pass

        for cls in ("BasicTransform", "ImageOnlyTransform", "RandomResizedCrop", "CenterCrop",
                    "Resize", "SmallestMaxSize", "Normalize", "ColorJitter", "Compose",
                    "HorizontalFlip", "ToFloat"):
            setattr(alb, cls, type(cls, (_T,), {}))
        sys.modules["albumentations"] = alb
    if "cv2" not in sys.modules:
        cv2 = types.ModuleType("cv2")
        cv2.INTER_LINEAR = cv2.INTER_CUBIC = cv2.BORDER_CONSTANT = 0
        sys.modules["cv2"] = cv2
    if "fvcore" not in sys.modules:
        # virtex/config.py:3 needs fvcore.common.config.CfgNode (yacs underneath): attribute-style nested dict with
        # merge_from_file (YAML, `_BASE_` chains), merge_from_list ("A.B.C", value pairs), freeze, dump, str.
        fv, fvc, fvcc = (types.ModuleType(n) for n in ("fvcore", "fvcore.common", "fvcore.common.config"))
        fvcc.CfgNode = _CfgNode
        fv.common, fvc.config = fvc, fvcc
        sys.modules.update({"fvcore": fv, "fvcore.common": fvc, "fvcore.common.config": fvcc})
    if "loguru" not in sys.modules:
        import logging
        lg = types.ModuleType("loguru")
        lg.logger = logging.getLogger("virtex-reference")
        lg.logger.success = lg.logger.info
        sys.modules["loguru"] = lg


class _CfgNode(dict):
    """Minimal stand-in for fvcore's CfgNode, enough for virtex/config.py:36-236."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if self.__dict__.get("_frozen"):
            raise AttributeError(f"config is frozen: cannot set {k}")
        self[k] = v

    def _merge(self, other):
        for k, v in other.items():
            if k not in self:
                raise KeyError(f"non-existent config key: {k}")
            if isinstance(self[k], _CfgNode):
                self[k]._merge(v)
            else:
                self[k] = type(self[k])(v) if isinstance(self[k], (float, tuple)) and not isinstance(v, str) else v

    def merge_from_file(self, path):
        import yaml
        with open(path) as f:
            data = yaml.safe_load(f) or {}
        base = data.pop("_BASE_", None)
        if base:
            self.merge_from_file(os.path.join(os.path.dirname(path), base))
        self._merge(data)

    def merge_from_list(self, lst):
        import ast
        assert len(lst) % 2 == 0
        for key, val in zip(lst[0::2], lst[1::2]):
            node = self
            *parents, leaf = key.split(".")
            for part in parents:
                node = node[part]
            if leaf not in node:
                raise KeyError(f"non-existent config key: {key}")
            if isinstance(val, str) and not isinstance(node[leaf], str):
                try:
                    val = ast.literal_eval(val)
                except (ValueError, SyntaxError):
                    pass
            node[leaf] = type(node[leaf])(val) if isinstance(node[leaf], float) else val

    def freeze(self):
        self.__dict__["_frozen"] = True
        for v in self.values():
            if isinstance(v, _CfgNode):
                v.freeze()

    def dump(self, stream=None):
        import yaml
        def plain(n):
            return {k: plain(v) if isinstance(v, _CfgNode) else v for k, v in n.items()}
        return yaml.safe_dump(plain(self), stream)

    def __str__(self):
        return self.dump()


def import_reference():
    """Returns the reference's ``virtex`` package objects needed by the hot path."""
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from virtex.models.captioning import VirTexModel  # noqa
    from virtex.modules.textual_heads import TransformerDecoderTextualHead  # noqa
    from virtex.modules.visual_backbones import TorchvisionVisualBackbone  # noqa
    from virtex.optim import Lookahead  # noqa
    from virtex.optim.lr_scheduler import LinearWarmupCosineAnnealingLR  # noqa

    return types.SimpleNamespace(
        VirTexModel=VirTexModel, TransformerDecoderTextualHead=TransformerDecoderTextualHead,
        TorchvisionVisualBackbone=TorchvisionVisualBackbone, Lookahead=Lookahead,
        LinearWarmupCosineAnnealingLR=LinearWarmupCosineAnnealingLR)


def import_reference_factories():
    """The reference's own `virtex.factories` and `virtex.config.Config`, imported verbatim (registries the native
    products are installed into: factories.py:317-319, 358-366, 418-426, 469-478)."""
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import virtex.factories as factories  # noqa
    from virtex.config import Config  # noqa
    return factories, Config


def build_reference_model(visual="torchvision::resnet50",
                          textual="transdec_postnorm::L1_H1024_A16_F4096",
                          vocab_size=10000, dropout=0.1, max_caption_length=30):
    """What PretrainingModelFactory.from_config does (virtex/factories.py:428-466), with
    the verbatim reference classes."""
    from oracle.bicaptioning import parse_textual_name

    ref = import_reference()
    vb = ref.TorchvisionVisualBackbone(visual.split("::")[-1], visual_feature_size=2048)
    feat = vb.cnn.layer4[-1].bn3.num_features
    vb.visual_feature_size = feat
    th = ref.TransformerDecoderTextualHead(
        visual_feature_size=feat, vocab_size=vocab_size, dropout=dropout,
        mask_future_positions=True, max_caption_length=max_caption_length, padding_idx=0,
        **parse_textual_name(textual))
    return ref.VirTexModel(vb, th, sos_index=1, eos_index=2, decoder=None)
