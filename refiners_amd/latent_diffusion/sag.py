"""Host-side mirror of refiners' Self-Attention Guidance (`foundationals/latent_diffusion/self_attention_guidance.py:22-105`,
`stable_diffusion_xl/self_attention_guidance.py`, `stable_diffusion_xl/model.py:164-250`) -- SURVEY.md section 8(f) next-4.

SAG adds a SECOND UNet pass per step: the middle block's first self-attention is tapped for its probabilities; keys that
receive more than one unit of attention mass (mean over heads, summed over queries) mark the salient region; there the
model's current data estimate is Gaussian-blurred, re-noised, and the unconditional branch is run again on the result:
    eps <- eps_cfg + sag_scale * (eps_uncond - eps_uncond(degraded latents)).
Class names, child order and context keys follow the reference (`repr` / inject / eject identical); on the MI355X the
step is driven by refiners_amd.engine.compiled.CompiledSDXL (second lowered program, native mask / blur kernels).
"""
import math
from typing import Any, Optional

import torch
import torch.nn.functional as F
from torch import Size, Tensor

import refiners_amd.fluxion.layers as fl
from refiners_amd.fluxion.adapters import Adapter
from refiners_amd.fluxion.leaves import interpolate
from refiners_amd.fluxion.tree import Contexts, bump_epoch

from .blocks import ResidualBlock


def gaussian_blur(tensor: Tensor, kernel_size: int, sigma: Optional[float] = None) -> Tensor:
    """Depthwise Gaussian blur with reflect padding (fluxion/utils.py:65-113), square kernels only."""
    assert torch.is_floating_point(tensor)
    s = sigma if sigma is not None else kernel_size * 0.15 + 0.35
    half = (kernel_size - 1) * 0.5
    x = torch.linspace(-half, half, steps=kernel_size, device=tensor.device, dtype=tensor.dtype)
    pdf = torch.exp(-0.5 * (x / s).pow(2))
    k1 = pdf / pdf.sum()
    kernel = torch.mm(k1[:, None], k1[None, :])
    channels = tensor.shape[-3]
    kernel = kernel.expand(channels, 1, kernel.shape[0], kernel.shape[1])
    p = kernel_size // 2
    return F.conv2d(F.pad(tensor, pad=(p, p, p, p), mode="reflect"), weight=kernel, groups=channels)


class SelfAttentionMap(fl.Passthrough):
    """Sits in front of a SelfAttention's ScaledDotProductAttention: stores softmax(Q K^T / sqrt(d)) per head."""

    def __init__(self, num_heads: int, context_key: str) -> None:
        self.num_heads = num_heads
        self.context_key = context_key
        super().__init__(fl.Lambda(func=self.compute_attention_scores), fl.SetContext(context="self_attention_map", key=context_key))

    def split_to_multi_head(self, x: Tensor) -> Tensor:
        assert len(x.shape) == 3 and x.shape[-1] % self.num_heads == 0
        return x.reshape(x.shape[0], x.shape[1], self.num_heads, x.shape[-1] // self.num_heads).transpose(1, 2)

    def compute_attention_scores(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor) -> torch.Tensor:
        query, key = self.split_to_multi_head(query), self.split_to_multi_head(key)
        dim = query.shape[-1]
        attention = query @ key.permute(0, 1, 3, 2)
        attention = attention / math.sqrt(dim)
        return torch.softmax(input=attention, dim=-1)


class SelfAttentionShape(fl.Passthrough):
    """Records the (H, W) of the feature map the tapped attention runs on."""

    def __init__(self, context_key: str) -> None:
        self.context_key = context_key
        super().__init__(fl.SetContext(context="self_attention_map", key=context_key, callback=self.register_shape))

    def register_shape(self, shapes: list, x: Tensor) -> None:
        assert x.ndim == 4, f"Expected 4D tensor, got {x.ndim}D with shape {x.shape}"
        shapes.append(x.shape[-2:])


class SAGAdapter(fl.Chain, Adapter[fl.Chain]):
    def __init__(self, target: fl.Chain, scale: float = 1.0, kernel_size: int = 9, sigma: float = 1.0) -> None:
        self._scale = scale
        self.kernel_size = kernel_size
        self.sigma = sigma
        with self.setup_adapter(target):
            super().__init__(target)

    @property
    def scale(self) -> float:
        return self._scale

    @scale.setter
    def scale(self, value: float) -> None:
        self._scale = value
        bump_epoch()

    def compute_sag_mask(self, latents: Tensor, classifier_free_guidance: bool = True) -> Tensor:
        attn_map = self.use_context("self_attention_map")["middle_block_attn_map"]
        if classifier_free_guidance:
            attn_map, _ = attn_map.chunk(2)
        attn_shape = self.use_context("self_attention_map")["middle_block_attn_shape"].pop()
        assert len(attn_shape) == 2
        b, c, h, w = latents.shape
        attn_h, attn_w = attn_shape
        attn_mask = attn_map.mean(dim=1, keepdim=False).sum(dim=1, keepdim=False) > 1.0
        attn_mask = attn_mask.reshape(b, attn_h, attn_w).unsqueeze(1).repeat(1, c, 1, 1).type(attn_map.dtype)
        return interpolate(attn_mask, Size((h, w)))

    def compute_degraded_latents(self, solver: Any, latents: Tensor, noise: Tensor, step: int, classifier_free_guidance: bool = True) -> Tensor:
        sag_mask = self.compute_sag_mask(latents=latents, classifier_free_guidance=classifier_free_guidance)
        original = solver.remove_noise(x=latents, noise=noise, step=step)
        degraded = gaussian_blur(original, kernel_size=self.kernel_size, sigma=self.sigma)
        degraded = degraded * sag_mask + original * (1 - sag_mask)
        return solver.add_noise(degraded, noise=noise, step=step)

    def init_context(self) -> Contexts:
        return {"self_attention_map": {"middle_block_attn_map": None, "middle_block_attn_shape": []}}


class _UNetSAGAdapter(SAGAdapter):
    """Shared inject / eject of the SD1.5 and SDXL adapters: SelfAttentionShape after the middle block's first ResidualBlock,
    SelfAttentionMap in front of the SDPA of the middle block's first SelfAttention."""

    def _middle_block(self) -> fl.Chain:
        return next(m for m in self.target.modules() if type(m).__name__ == "MiddleBlock")

    def inject(self, parent: Optional[fl.Chain] = None) -> "_UNetSAGAdapter":
        middle = self._middle_block()
        middle.insert_after_type(ResidualBlock, SelfAttentionShape(context_key="middle_block_attn_shape"))
        self_attn = middle.ensure_find(fl.SelfAttention)
        self_attn.insert_before_type(fl.ScaledDotProductAttention, SelfAttentionMap(num_heads=self_attn.num_heads, context_key="middle_block_attn_map"))
        return super().inject(parent)  # type: ignore[return-value]

    def eject(self) -> None:
        middle = self._middle_block()
        middle.remove(middle.ensure_find(SelfAttentionShape))
        self_attn = middle.ensure_find(fl.SelfAttention)
        self_attn.remove(self_attn.ensure_find(SelfAttentionMap))
        super().eject()


class SDXLSAGAdapter(_UNetSAGAdapter):
    pass


class SD1SAGAdapter(_UNetSAGAdapter):
    pass
