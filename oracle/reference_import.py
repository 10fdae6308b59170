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
