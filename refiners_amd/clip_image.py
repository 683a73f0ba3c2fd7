"""Host-side mirror of refiners' CLIP image encoders (`src/refiners/foundationals/clip/image_encoder.py:9-239`): the
ViT that turns an IP-Adapter image prompt into the 1024-wide embedding `ImageProjection` expands to 4 image tokens
(`latent_diffusion/image_prompt.py:24-45, 500-510`) -- SURVEY.md section 8(f) next-2.  Same Chain layout and state-dict
keys as the reference (tests/golden/clip_image_h_keys.json); lowered by refiners_amd/engine/image_prompt.py."""
from __future__ import annotations

from typing import Any, Callable

from torch import Tensor

from .clip import FeedForward, PositionalEncoder
from .fluxion import layers as fl


class ClassToken(fl.Chain):
    def __init__(self, embedding_dim: int, device: Any = None, dtype: Any = None) -> None:
        self.embedding_dim = embedding_dim
        super().__init__(fl.Parameter(1, embedding_dim, device=device, dtype=dtype))


class PatchEncoder(fl.Chain):
    """Non-overlapping P x P patches -> embedding (a stride-P convolution), channels last."""

    def __init__(self, in_channels: int, out_channels: int, patch_size: int = 16, use_bias: bool = True, device: Any = None, dtype: Any = None) -> None:
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.patch_size = patch_size
        self.use_bias = use_bias
        super().__init__(
            fl.Conv2d(in_channels, out_channels, kernel_size=(patch_size, patch_size), stride=(patch_size, patch_size), use_bias=use_bias, device=device, dtype=dtype),
            fl.Permute(0, 2, 3, 1),
        )


class TransformerLayer(fl.Chain):
    """x += SelfAttention(LN(x)); x += FeedForward(LN(x)), bidirectional (`clip/image_encoder.py:63-92`)."""

    def __init__(self, embedding_dim: int = 768, feedforward_dim: int = 3072, num_attention_heads: int = 12, layer_norm_eps: float = 1e-5,
                 device: Any = None, dtype: Any = None) -> None:
        self.embedding_dim = embedding_dim
        self.feedforward_dim = feedforward_dim
        self.num_attention_heads = num_attention_heads
        self.layer_norm_eps = layer_norm_eps
        super().__init__(
            fl.Residual(
                fl.LayerNorm(embedding_dim, eps=layer_norm_eps, device=device, dtype=dtype),
                fl.SelfAttention(embedding_dim=embedding_dim, num_heads=num_attention_heads, device=device, dtype=dtype),
            ),
            fl.Residual(
                fl.LayerNorm(embedding_dim, eps=layer_norm_eps, device=device, dtype=dtype),
                FeedForward(embedding_dim=embedding_dim, feedforward_dim=feedforward_dim, device=device, dtype=dtype),
            ),
        )


class ViTEmbeddings(fl.Chain):
    """[class token ; patch embeddings] + learned positions (`clip/image_encoder.py:95-131`)."""

    def __init__(self, image_size: int = 224, embedding_dim: int = 768, patch_size: int = 32, device: Any = None, dtype: Any = None) -> None:
        self.image_size = image_size
        self.embedding_dim = embedding_dim
        self.patch_size = patch_size
        n = (image_size // patch_size) ** 2
        super().__init__(
            fl.Concatenate(
                ClassToken(embedding_dim, device=device, dtype=dtype),
                fl.Chain(
                    PatchEncoder(3, embedding_dim, patch_size=patch_size, use_bias=False, device=device, dtype=dtype),
                    fl.Reshape(n, embedding_dim),
                ),
                dim=1,
            ),
            fl.Residual(PositionalEncoder(max_sequence_length=n + 1, embedding_dim=embedding_dim, device=device, dtype=dtype)),
        )


class CLIPImageEncoder(fl.Chain):
    """embeddings -> LayerNorm -> N transformer layers -> class token -> LayerNorm -> bias-free projection
    (`clip/image_encoder.py:134-197`)."""

    def __init__(self, image_size: int = 224, embedding_dim: int = 768, output_dim: int = 512, patch_size: int = 32, num_layers: int = 12,
                 num_attention_heads: int = 12, feedforward_dim: int = 3072, layer_norm_eps: float = 1e-5, device: Any = None, dtype: Any = None) -> None:
        self.image_size = image_size
        self.embedding_dim = embedding_dim
        self.output_dim = output_dim
        self.patch_size = patch_size
        self.num_layers = num_layers
        self.num_attention_heads = num_attention_heads
        self.feedforward_dim = feedforward_dim
        cls_token_pooling: Callable[[Tensor], Tensor] = lambda x: x[:, 0, :]  # noqa: E731
        super().__init__(
            ViTEmbeddings(image_size=image_size, embedding_dim=embedding_dim, patch_size=patch_size, device=device, dtype=dtype),
            fl.LayerNorm(embedding_dim, eps=layer_norm_eps, device=device, dtype=dtype),
            fl.Chain(
                TransformerLayer(embedding_dim, feedforward_dim, num_attention_heads, layer_norm_eps, device=device, dtype=dtype)
                for _ in range(num_layers)
            ),
            fl.Lambda(func=cls_token_pooling),
            fl.LayerNorm(embedding_dim, eps=layer_norm_eps, device=device, dtype=dtype),
            fl.Linear(embedding_dim, output_dim, bias=False, device=device, dtype=dtype),
        )


class CLIPImageEncoderH(CLIPImageEncoder):
    """ViT-H/14: 1280 wide, 32 layers, 16 heads of 80, 1024-wide output (`clip/image_encoder.py:200-226`)."""

    def __init__(self, device: Any = None, dtype: Any = None) -> None:
        super().__init__(embedding_dim=1280, output_dim=1024, patch_size=14, num_layers=32, num_attention_heads=16, feedforward_dim=5120, device=device, dtype=dtype)


class CLIPImageEncoderG(CLIPImageEncoder):
    """ViT-bigG/14: 1664 wide, 48 layers, 16 heads of 104, 1280-wide output (`clip/image_encoder.py:229-239`)."""

    def __init__(self, device: Any = None, dtype: Any = None) -> None:
        super().__init__(embedding_dim=1664, output_dim=1280, patch_size=14, num_layers=48, num_attention_heads=16, feedforward_dim=8192, device=device, dtype=dtype)
