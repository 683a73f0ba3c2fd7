"""SegmentAnything ViT-H image encoder as a Chain tree (BASELINE.json config 5).

Mirrors reference src/refiners/foundationals/segment_anything/image_encoder.py:8-368 (same class names, child order and
parameter names, hence the same state-dict keys -- checked against tests/golden/sam_vit_h_keys.json) and the HQ-SAM encoder
hook `SAMViTAdapter` of segment_anything/hq_sam.py:230-264.  Everything here is the unfused torch path; the MI355X engine
(refiners_amd/engine/sam.py) lowers the same tree.

Shapes for ViT-H: (B, 3, 1024, 1024) -> patch conv 16x16/16 -> (B, 64, 64, 1280) channels-last tokens -> 32 layers
(14x14 windowed attention with the grid padded 64 -> 70, global attention in layers 7, 15, 23, 31, 16 heads of 80,
decomposed relative position bias) -> neck -> (B, 256, 64, 64).
"""
from __future__ import annotations

from typing import Any

import torch
import torch.nn.functional as F
from torch import Tensor, nn

import refiners_amd.fluxion.layers as fl
from refiners_amd.fluxion.adapters import Adapter
from refiners_amd.fluxion.tree import Contexts


class PatchEncoder(fl.Chain):
    """Non-overlapping patch convolution, output channels-last."""

    def __init__(self, in_channels: int, out_channels: int, patch_size: int = 16, use_bias: bool = True, device: Any = None, dtype: Any = None) -> None:
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.patch_size = patch_size
        self.use_bias = use_bias
        super().__init__(
            fl.Conv2d(in_channels, out_channels, kernel_size=(patch_size, patch_size), stride=(patch_size, patch_size), use_bias=use_bias, device=device, dtype=dtype),
            fl.Permute(0, 2, 3, 1),
        )


class PositionalEncoder(fl.Residual):
    """x + learned (H, W, C) position table."""

    def __init__(self, embedding_dim: int, image_embedding_size: tuple[int, int], device: Any = None, dtype: Any = None) -> None:
        self.embedding_dim = embedding_dim
        self.image_embedding_size = image_embedding_size
        super().__init__(fl.Parameter(image_embedding_size[0], image_embedding_size[1], embedding_dim, device=device, dtype=dtype))


class RelativePositionAttention(fl.WeightedModule):
    """softmax(q k^T / sqrt(d) + rel_h + rel_w) v on a packed (B, H, W, 3C) qkv tensor; the bias is the decomposed
    relative position term rel_h[q, kh] = q . Rh[qh - kh], rel_w[q, kw] = q . Rw[qw - kw] of the UNSCALED query."""

    def __init__(self, embedding_dim: int, num_heads: int, spatial_size: tuple[int, int], device: Any = None, dtype: Any = None) -> None:
        super().__init__()
        self.embedding_dim = embedding_dim
        self.num_heads = num_heads
        self.head_dim = embedding_dim // num_heads
        self.spatial_size = spatial_size
        self.horizontal_embedding = nn.Parameter(torch.zeros(2 * spatial_size[0] - 1, self.head_dim, device=device, dtype=dtype))
        self.vertical_embedding = nn.Parameter(torch.zeros(2 * spatial_size[1] - 1, self.head_dim, device=device, dtype=dtype))

    @property
    def device(self) -> torch.device:
        return self.horizontal_embedding.device

    @property
    def dtype(self) -> torch.dtype:
        return self.horizontal_embedding.dtype

    @staticmethod
    def relative_index(size: int) -> Tensor:
        i = torch.arange(size)
        return i[:, None] - i[None, :] + size - 1

    def bias_terms(self, q: Tensor) -> tuple[Tensor, Tensor]:
        """q: (N, L, d) -> (horizontal (N, h, w, 1, w'), vertical (N, h, w, h', 1)).  NB: the reference unpacks
        `width, height = spatial_size` and reshapes q to (N, width, height, d); spatial sizes are square in every SAM
        configuration, which is what makes that consistent."""
        width, height = self.spatial_size
        hor = self.horizontal_embedding[self.relative_index(width)]
        ver = self.vertical_embedding[self.relative_index(height)]
        q4 = q.reshape(q.shape[0], width, height, -1)
        rel_hor = torch.einsum("bhwc,wkc->bhwk", q4, hor).unsqueeze(-2)
        rel_ver = torch.einsum("bhwc,hkc->bhwk", q4, ver).unsqueeze(-1)
        return rel_hor, rel_ver

    def forward(self, x: Tensor) -> Tensor:
        batch, height, width, _ = x.shape
        qkv = x.reshape(batch, width * height, 3, self.num_heads, -1).permute(2, 0, 3, 1, 4).reshape(3, batch * self.num_heads, width * height, -1)
        q, k, v = qkv.unbind(0)
        rel_hor, rel_ver = self.bias_terms(q)
        att = (q * self.head_dim ** -0.5) @ k.transpose(-2, -1)
        att = ((att.reshape(-1, height, width, height, width) + rel_ver) + rel_hor).reshape(att.shape)
        out = att.softmax(dim=-1) @ v
        return out.reshape(batch, self.num_heads, height, width, -1).permute(0, 2, 3, 1, 4).reshape(batch, height, width, -1)


class FusedSelfAttention(fl.Chain):
    """Linear(C -> 3C) -> RelativePositionAttention -> Linear(C -> C)."""

    def __init__(self, embedding_dim: int = 768, spatial_size: tuple[int, int] = (64, 64), num_heads: int = 1, use_bias: bool = True,
                 is_causal: bool = False, device: Any = None, dtype: Any = None) -> None:
        assert embedding_dim % num_heads == 0, f"Embedding dim (embedding_dim={embedding_dim}) must be divisible by num heads (num_heads={num_heads})"
        self.embedding_dim = embedding_dim
        self.num_heads = num_heads
        self.use_bias = use_bias
        self.is_causal = is_causal
        super().__init__(
            fl.Linear(embedding_dim, 3 * embedding_dim, bias=use_bias, device=device, dtype=dtype),
            RelativePositionAttention(embedding_dim, num_heads, spatial_size, device=device, dtype=dtype),
            fl.Linear(embedding_dim, embedding_dim, bias=True, device=device, dtype=dtype),
        )


class FeedForward(fl.Chain):
    def __init__(self, embedding_dim: int, feedforward_dim: int, device: Any = None, dtype: Any = None) -> None:
        self.embedding_dim = embedding_dim
        self.feedforward_dim = feedforward_dim
        super().__init__(
            fl.Linear(embedding_dim, feedforward_dim, bias=True, device=device, dtype=dtype),
            fl.GeLU(),
            fl.Linear(feedforward_dim, embedding_dim, bias=True, device=device, dtype=dtype),
        )


class WindowPartition(fl.ContextModule):
    """(B, H, W, C) -> (B * nH * nW, ws, ws, C), zero padding H and W up to multiples of the window size (kept in context)."""

    def __init__(self) -> None:
        super().__init__()

    def forward(self, x: Tensor) -> Tensor:
        b, h, w, c = x.shape
        ctx = self.use_context("window_partition")
        ws = ctx["window_size"]
        ph, pw = (ws - h % ws) % ws, (ws - w % ws) % ws
        if ph or pw:
            x = F.pad(x, (0, 0, 0, pw, 0, ph))
        ctx.update({"original_height": h, "original_width": w, "padded_height": h + ph, "padded_width": w + pw})
        x = x.view(b, (h + ph) // ws, ws, (w + pw) // ws, ws, c)
        return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, c)


class WindowMerge(fl.ContextModule):
    def __init__(self) -> None:
        super().__init__()

    def forward(self, x: Tensor) -> Tensor:
        ctx = self.use_context("window_partition")
        ws, ph, pw = ctx["window_size"], ctx["padded_height"], ctx["padded_width"]
        h, w = ctx["original_height"], ctx["original_width"]
        b = x.shape[0] // (ph * pw // ws // ws)
        x = x.view(b, ph // ws, pw // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).contiguous().view(b, ph, pw, -1)
        return x[:, :h, :w, :].contiguous() if (ph > h or pw > w) else x


class TransformerLayer(fl.Chain):
    def __init__(self, embedding_dim: int, num_heads: int, feedforward_dim: int, image_embedding_size: tuple[int, int],
                 window_size: int | None = None, layer_norm_eps: float = 1e-6, device: Any = None, dtype: Any = None) -> None:
        self.embedding_dim = embedding_dim
        self.num_heads = num_heads
        self.feedforward_dim = feedforward_dim
        self.window_size = window_size
        self.layer_norm_eps = layer_norm_eps
        self.image_embedding_size = image_embedding_size
        kw = dict(device=device, dtype=dtype)
        windowed = window_size is not None
        super().__init__(
            fl.Residual(
                fl.LayerNorm(embedding_dim, eps=layer_norm_eps, **kw),
                WindowPartition() if windowed else fl.Identity(),
                FusedSelfAttention(embedding_dim=embedding_dim, num_heads=num_heads,
                                   spatial_size=(window_size, window_size) if windowed else image_embedding_size, **kw),
                WindowMerge() if windowed else fl.Reshape(image_embedding_size[0], image_embedding_size[1], embedding_dim),
            ),
            fl.Residual(
                fl.LayerNorm(embedding_dim, eps=layer_norm_eps, **kw),
                FeedForward(embedding_dim=embedding_dim, feedforward_dim=feedforward_dim, **kw),
            ),
        )

    def init_context(self) -> Contexts:
        return {"window_partition": {"window_size": self.window_size}}


class Neck(fl.Chain):
    def __init__(self, in_channels: int = 768, device: Any = None, dtype: Any = None) -> None:
        self.in_channels = in_channels
        kw = dict(device=device, dtype=dtype)
        super().__init__(
            fl.Permute(0, 3, 1, 2),
            fl.Conv2d(in_channels, 256, kernel_size=1, use_bias=False, **kw),
            fl.LayerNorm2d(256, **kw),
            fl.Conv2d(256, 256, kernel_size=3, padding=1, use_bias=False, **kw),
            fl.LayerNorm2d(256, **kw),
        )


class Transformer(fl.Chain):
    pass


class SAMViT(fl.Chain):
    def __init__(self, embedding_dim: int, num_layers: int, num_heads: int, global_attention_indices: tuple[int, ...] | None = None,
                 device: Any = None, dtype: Any = None) -> None:
        self.embedding_dim = embedding_dim
        self.num_layers = num_layers
        self.num_heads = num_heads
        self.image_size = (1024, 1024)
        self.patch_size = 16
        self.window_size = 14
        self.image_embedding_size = (self.image_size[0] // self.patch_size, self.image_size[1] // self.patch_size)
        self.feed_forward_dim = 4 * embedding_dim
        self.global_attention_indices = global_attention_indices or tuple()
        kw = dict(device=device, dtype=dtype)
        super().__init__(
            PatchEncoder(in_channels=3, out_channels=embedding_dim, patch_size=self.patch_size, **kw),
            PositionalEncoder(embedding_dim=embedding_dim, image_embedding_size=self.image_embedding_size, **kw),
            Transformer(
                TransformerLayer(embedding_dim=embedding_dim, num_heads=num_heads, feedforward_dim=self.feed_forward_dim,
                                 window_size=None if i in self.global_attention_indices else self.window_size,
                                 image_embedding_size=self.image_embedding_size, **kw)
                for i in range(num_layers)
            ),
            Neck(in_channels=embedding_dim, **kw),
        )


class SAMViTH(SAMViT):
    def __init__(self, device: Any = None, dtype: Any = None) -> None:
        super().__init__(embedding_dim=1280, num_layers=32, num_heads=16, global_attention_indices=(7, 15, 23, 31), device=device, dtype=dtype)


class SAMViTAdapter(fl.Chain, Adapter[SAMViT]):
    """HQ-SAM's encoder hook: stores the output of the first global-attention layer in context "hq_sam".early_vit_embedding."""

    def __init__(self, target: SAMViT) -> None:
        with self.setup_adapter(target):
            super().__init__(target)
        layer = next((t for t in target.layers(TransformerLayer) if t.window_size is None), None)
        assert layer is not None
        self._transformer_layer = [layer]
        self._set_early_vit_embedding_context = [fl.SetContext("hq_sam", "early_vit_embedding")]

    @property
    def target_transformer_layer(self) -> TransformerLayer:
        return self._transformer_layer[0]

    @property
    def set_early_vit_embedding_context(self) -> fl.SetContext:
        return self._set_early_vit_embedding_context[0]

    def inject(self, parent: fl.Chain | None = None) -> "SAMViTAdapter":
        self.target_transformer_layer.append(self.set_early_vit_embedding_context)
        return super().inject(parent)

    def eject(self) -> None:
        self.target_transformer_layer.remove(self.set_early_vit_embedding_context)
        super().eject()
