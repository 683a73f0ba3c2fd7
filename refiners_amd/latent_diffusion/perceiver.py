"""Host-side mirror of the fine-grained ("plus") IP-Adapter's image projection: a Perceiver resampler over the image encoder's
patch features (`src/refiners/foundationals/latent_diffusion/image_prompt.py:48-234, 553-564`) -- SURVEY.md section 8(f) next-2.
Class names, constructor signatures and child order follow the reference so that state-dict keys (`Transformer.TransformerLayer_N
.Residual_1.PerceiverAttention...`) and `repr` are identical; the forwards are the unfused torch path, lowered for the MI355X by
refiners_amd/engine/image_prompt.py (lower_perceiver)."""
import math
from typing import Any

import torch
from torch import Tensor

import refiners_amd.fluxion.layers as fl
from refiners_amd.fluxion.tree import Contexts


class FeedForward(fl.Chain):
    """Linear -> GeLU -> Linear, no biases (image_prompt.py:48-77; named FeedForward there)."""

    def __init__(self, embedding_dim: int, feedforward_dim: int, device: Any = None, dtype: Any = None) -> None:
        self.embedding_dim = embedding_dim
        self.feedforward_dim = feedforward_dim
        super().__init__(
            fl.Linear(embedding_dim, feedforward_dim, bias=False, device=device, dtype=dtype),
            fl.GeLU(),
            fl.Linear(feedforward_dim, embedding_dim, bias=False, device=device, dtype=dtype),
        )


class PerceiverScaledDotProductAttention(fl.Module):
    """softmax((q s)(k s)^T) v with s = head_dim^-1/4 on both operands ("more stable with f16 than dividing afterwards"), the
    softmax evaluated in float32; inputs (key_value [B, Lk, 2 * inner], query [B, Lq, inner]) -- image_prompt.py:84-121."""

    def __init__(self, head_dim: int, num_heads: int) -> None:
        super().__init__()
        self.num_heads = num_heads
        self.scale = 1 / math.sqrt(math.sqrt(head_dim))

    def forward(self, key_value: Tensor, query: Tensor) -> Tensor:
        bs, length, _ = query.shape
        key, value = key_value.chunk(2, dim=-1)
        q, k, v = self.reshape_tensor(query), self.reshape_tensor(key), self.reshape_tensor(value)
        attention = (q * self.scale) @ (k * self.scale).transpose(-2, -1)
        attention = torch.softmax(input=attention.float(), dim=-1).type(attention.dtype)
        attention = attention @ v
        return attention.permute(0, 2, 1, 3).reshape(bs, length, -1)

    def reshape_tensor(self, x: Tensor) -> Tensor:
        bs, length, _ = x.shape
        return x.view(bs, length, self.num_heads, -1).transpose(1, 2).reshape(bs, self.num_heads, length, -1)


class PerceiverAttention(fl.Chain):
    """(x, latents) -> Wo attention(q = Wq LN2(latents), kv = Wkv [LN1(x) ; LN2(latents)])   (image_prompt.py:124-175)."""

    def __init__(self, embedding_dim: int, head_dim: int = 64, num_heads: int = 8, device: Any = None, dtype: Any = None) -> None:
        self.embedding_dim = embedding_dim
        self.head_dim = head_dim
        self.inner_dim = head_dim * num_heads
        kw = dict(device=device, dtype=dtype)
        super().__init__(
            fl.Distribute(fl.LayerNorm(embedding_dim, **kw), fl.LayerNorm(embedding_dim, **kw)),
            fl.Parallel(
                fl.Chain(fl.Lambda(func=self.to_kv), fl.Linear(embedding_dim, 2 * self.inner_dim, bias=False, **kw)),
                fl.Chain(fl.GetArg(index=1), fl.Linear(embedding_dim, self.inner_dim, bias=False, **kw)),
            ),
            PerceiverScaledDotProductAttention(head_dim=head_dim, num_heads=num_heads),
            fl.Linear(self.inner_dim, embedding_dim, bias=False, **kw),
        )

    def to_kv(self, x: torch.Tensor, latents: torch.Tensor) -> torch.Tensor:  # (annotations as in the reference: they are part of repr())
        return torch.cat((x, latents), dim=-2)


class LatentsToken(fl.Chain):
    def __init__(self, num_tokens: int, latents_dim: int, device: Any = None, dtype: Any = None) -> None:
        self.num_tokens = num_tokens
        self.latents_dim = latents_dim
        super().__init__(fl.Parameter(num_tokens, latents_dim, device=device, dtype=dtype))


class Transformer(fl.Chain):
    pass


class TransformerLayer(fl.Chain):
    pass


class PerceiverResampler(fl.Chain):
    """Image encoder patch features (B, 257, input_dim) -> `num_tokens` image tokens (B, num_tokens, output_dim): learned latent
    queries cross-attend to [features ; latents] for `num_attention_layers` layers (image_prompt.py:178-234).  SDXL "plus"
    adapters use latents_dim 1280, 4 layers, 20 heads of 64, 16 tokens, input 1280, output 2048 (xl/image_prompt.py:43-53)."""

    def __init__(self, latents_dim: int = 1024, num_attention_layers: int = 8, num_attention_heads: int = 16, head_dim: int = 64, num_tokens: int = 8,
                 input_dim: int = 768, output_dim: int = 1024, device: Any = None, dtype: Any = None) -> None:
        self.latents_dim = latents_dim
        self.num_attention_layers = num_attention_layers
        self.head_dim = head_dim
        self.num_attention_heads = num_attention_heads
        self.num_tokens = num_tokens
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.feedforward_dim = 4 * latents_dim
        kw = dict(device=device, dtype=dtype)
        layers = [
            TransformerLayer(
                fl.Residual(
                    fl.Parallel(fl.UseContext(context="perceiver_resampler", key="x"), fl.Identity()),
                    PerceiverAttention(embedding_dim=latents_dim, head_dim=head_dim, num_heads=num_attention_heads, **kw),
                ),
                fl.Residual(fl.LayerNorm(latents_dim, **kw), FeedForward(embedding_dim=latents_dim, feedforward_dim=self.feedforward_dim, **kw)),
            )
            for _ in range(num_attention_layers)
        ]
        super().__init__(
            fl.Linear(input_dim, latents_dim, **kw),
            fl.SetContext(context="perceiver_resampler", key="x"),
            LatentsToken(num_tokens, latents_dim, **kw),
            Transformer(*layers),
            fl.Linear(latents_dim, output_dim, **kw),
            fl.LayerNorm(output_dim, **kw),
        )

    def init_context(self) -> Contexts:
        return {"perceiver_resampler": {"x": None}}


def convert_to_grid_features(clip_image_encoder: fl.Chain) -> fl.Chain:
    """The image encoder WITHOUT class-token pooling, final LayerNorm, projection and last transformer layer: its output is the
    penultimate layer's (B, 257, 1280) token grid (IPAdapter.convert_to_grid_features, image_prompt.py:553-564)."""
    clone = clip_image_encoder.structural_copy()
    assert isinstance(clone[-1], fl.Linear) and isinstance(clone[-2], fl.LayerNorm) and isinstance(clone[-3], fl.Lambda)
    for _ in range(3):
        clone.pop()
    layers = clone[-1]
    assert isinstance(layers, fl.Chain) and len(layers) == 32
    layers.pop()
    return clone


