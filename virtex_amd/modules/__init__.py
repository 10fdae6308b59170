from .visual_backbones import TorchvisionVisualBackbone, VisualBackbone  # noqa: F401
from .textual_heads import (TextualHead, TransformerDecoderTextualHead,  # noqa: F401
                            WordAndPositionalEmbedding)
