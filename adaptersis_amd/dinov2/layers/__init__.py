"""Mirror of `dinov2/layers/__init__.py` for the modules on the hot path."""
from .blocks import (Attention, Block, LayerScale, MemEffAttention, Mlp, NestedTensorBlock, PatchEmbed,  # noqa: F401
                     SwiGLUFFN, SwiGLUFFNFused)
