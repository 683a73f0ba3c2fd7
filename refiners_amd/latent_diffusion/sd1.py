"""Stable Diffusion 1.5 denoiser as a Chain tree (reference: latent_diffusion/stable_diffusion_1/unet.py:16-249).

Only needed for BASELINE.json config 1 (the reference's own CPU-runnable case).  Same data-driven construction as
sdxl.py; differences from SDXL: 8 heads everywhere (head dims 40/80/160), text dim 768, Conv2d 1x1 projections around
the transformer, 13 residual slots, a third downsample, and the middle block wrapped in a Sum with residuals[-1].
"""
from __future__ import annotations

from typing import Any

from torch import Tensor

import refiners_amd.fluxion.layers as fl
from refiners_amd.fluxion.tree import Contexts

from .blocks import CrossAttentionBlock2d, RangeEncoder, ResidualBlock
from .sdxl import wire_unet

_DOWN = [
    [("res", 320, 320), ("attn", 320)],
    [("res", 320, 320), ("attn", 320)],
    [("down", 320)],
    [("res", 320, 640), ("attn", 640)],
    [("res", 640, 640), ("attn", 640)],
    [("down", 640)],
    [("res", 640, 1280), ("attn", 1280)],
    [("res", 1280, 1280), ("attn", 1280)],
    [("down", 1280)],
    [("res", 1280, 1280)],
    [("res", 1280, 1280)],
]
_UP = [
    [("res", 2560, 1280)],
    [("res", 2560, 1280)],
    [("res", 2560, 1280), ("up", 1280)],
    [("res", 2560, 1280), ("attn", 1280)],
    [("res", 2560, 1280), ("attn", 1280)],
    [("res", 1920, 1280), ("attn", 1280), ("up", 1280)],
    [("res", 1920, 640), ("attn", 640)],
    [("res", 1280, 640), ("attn", 640)],
    [("res", 960, 640), ("attn", 640), ("up", 640)],
    [("res", 960, 320), ("attn", 320)],
    [("res", 640, 320), ("attn", 320)],
    [("res", 640, 320), ("attn", 320)],
]


class CLIPLCrossAttention(CrossAttentionBlock2d):
    def __init__(self, channels: int, device: Any = None, dtype: Any = None) -> None:
        super().__init__(
            channels=channels, context_embedding_dim=768, context_key="clip_text_embedding", num_attention_heads=8,
            use_bias=False, device=device, dtype=dtype,
        )


def _piece(spec: tuple, kw: dict[str, Any]) -> fl.Module:
    kind = spec[0]
    if kind == "res":
        return ResidualBlock(in_channels=spec[1], out_channels=spec[2], **kw)
    if kind == "attn":
        return CLIPLCrossAttention(channels=spec[1], **kw)
    if kind == "down":
        return fl.Downsample(channels=spec[1], scale_factor=2, padding=1, **kw)
    if kind == "up":
        return fl.Upsample(channels=spec[1], **kw)
    raise ValueError(spec)


class TimestepEncoder(fl.Passthrough):
    def __init__(self, context_key: str = "timestep_embedding", device: Any = None, dtype: Any = None) -> None:
        super().__init__(
            fl.UseContext("diffusion", "timestep"),
            RangeEncoder(320, 1280, device=device, dtype=dtype),
            fl.SetContext("range_adapter", context_key),
        )


class DownBlocks(fl.Chain):
    def __init__(self, in_channels: int, device: Any = None, dtype: Any = None):
        self.in_channels = in_channels
        kw = dict(device=device, dtype=dtype)
        stem = fl.Chain(fl.Conv2d(in_channels, 320, kernel_size=3, padding=1, **kw))
        super().__init__(stem, *(fl.Chain(*(_piece(s, kw) for s in block)) for block in _DOWN))


class UpBlocks(fl.Chain):
    def __init__(self, device: Any = None, dtype: Any = None) -> None:
        kw = dict(device=device, dtype=dtype)
        super().__init__(*(fl.Chain(*(_piece(s, kw) for s in block)) for block in _UP))


class MiddleBlock(fl.Chain):
    def __init__(self, device: Any = None, dtype: Any = None) -> None:
        kw = dict(device=device, dtype=dtype)
        super().__init__(_piece(("res", 1280, 1280), kw), _piece(("attn", 1280), kw), _piece(("res", 1280, 1280), kw))


class SD1UNet(fl.Chain):
    def __init__(self, in_channels: int, device: Any = None, dtype: Any = None) -> None:
        self.in_channels = in_channels
        kw = dict(device=device, dtype=dtype)
        super().__init__(
            TimestepEncoder(**kw),
            DownBlocks(in_channels=in_channels, **kw),
            fl.Sum(
                fl.UseContext(context="unet", key="residuals").compose(lambda residuals: residuals[-1]),
                MiddleBlock(**kw),
            ),
            UpBlocks(**kw),
            fl.Chain(
                fl.GroupNorm(channels=320, num_groups=32, **kw),
                fl.SiLU(),
                fl.Conv2d(320, 4, kernel_size=3, stride=1, padding=1, **kw),
            ),
        )
        wire_unet(self, device, dtype)

    def init_context(self) -> Contexts:
        return {
            "unet": {"residuals": [0.0] * 13},
            "diffusion": {"timestep": None},
            "range_adapter": {"timestep_embedding": None},
            "sampling": {"shapes": []},
        }

    def set_clip_text_embedding(self, clip_text_embedding: Tensor) -> None:
        self.set_context("cross_attention_block", {"clip_text_embedding": clip_text_embedding})

    def set_timestep(self, timestep: Tensor) -> None:
        self.set_context("diffusion", {"timestep": timestep})
