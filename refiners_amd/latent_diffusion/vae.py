"""Latent-diffusion autoencoder as Chain trees (the step right after the sampling loop).

Mirrors reference src/refiners/foundationals/latent_diffusion/auto_encoder.py:83-330 (Resnet, Encoder, Decoder,
LatentDiffusionAutoencoder.encode / decode) and stable_diffusion_xl/model.py:12-19 (SDXLAutoencoder.encoder_scale);
same child order and class names, hence the same state-dict keys (tests/golden/vae_keys.json).  Tensor in, tensor out:
the PIL helpers and the tiled-inference utilities of the reference are not mirrored.  These forwards are the unfused torch
path; refiners_amd.engine.vae.CompiledVAEDecoder lowers the Decoder tree onto the MI355X kernels.
"""
from __future__ import annotations

from typing import Any

from torch import Tensor

import refiners_amd.fluxion.layers as fl
from refiners_amd.fluxion.leaves import Slicing
from refiners_amd.fluxion.tree import Contexts

_WIDTHS = [128, 256, 512, 512, 512]


class Resnet(fl.Sum):
    """Sum( shortcut (1x1 conv iff channels change) , GN -> SiLU -> conv3x3 -> GN -> SiLU -> conv3x3 )."""

    def __init__(self, in_channels: int, out_channels: int, num_groups: int = 32, device: Any = None, dtype: Any = None):
        self.in_channels = in_channels
        self.out_channels = out_channels
        kw = dict(device=device, dtype=dtype)
        shortcut = fl.Conv2d(in_channels, out_channels, kernel_size=1, **kw) if in_channels != out_channels else fl.Identity()
        super().__init__(
            shortcut,
            fl.Chain(
                fl.GroupNorm(channels=in_channels, num_groups=num_groups, **kw),
                fl.SiLU(),
                fl.Conv2d(in_channels, out_channels, kernel_size=3, padding=1, **kw),
                fl.GroupNorm(channels=out_channels, num_groups=num_groups, **kw),
                fl.SiLU(),
                fl.Conv2d(out_channels, out_channels, kernel_size=3, padding=1, **kw),
            ),
        )


def _attention(channels: int, kw: dict[str, Any]) -> fl.Residual:
    return fl.Residual(fl.GroupNorm(channels=channels, num_groups=32, eps=1e-6, **kw), fl.SelfAttention2d(channels=channels, **kw))


class Encoder(fl.Chain):
    def __init__(self, device: Any = None, dtype: Any = None) -> None:
        kw = dict(device=device, dtype=dtype)
        w = _WIDTHS
        stages = [fl.Chain([Resnet(w[i - 1] if i > 0 else w[0], w[i], **kw), Resnet(w[i], w[i], **kw)]) for i in range(len(w))]
        for stage in stages[:3]:
            stage.append(fl.Downsample(channels=stage[-1].out_channels, scale_factor=2, **kw))
        stages[-1].insert_after_type(Resnet, _attention(w[-1], kw))
        super().__init__(
            fl.Conv2d(3, w[0], kernel_size=3, padding=1, **kw),
            fl.Chain(*stages),
            fl.Chain(fl.GroupNorm(channels=w[-1], num_groups=32, eps=1e-6, **kw), fl.SiLU(), fl.Conv2d(w[-1], 8, kernel_size=3, padding=1, **kw)),
            fl.Chain(fl.Conv2d(8, 8, kernel_size=1, **kw), Slicing(dim=1, end=4)),
        )

    def init_context(self) -> Contexts:
        return {"sampling": {"shapes": []}}


class Decoder(fl.Chain):
    def __init__(self, device: Any = None, dtype: Any = None) -> None:
        self.resnet_sizes: list[int] = list(_WIDTHS)
        self.latent_dim: int = 4
        self.output_channels: int = 3
        kw = dict(device=device, dtype=dtype)
        w = _WIDTHS[::-1]
        stages = []
        for i in range(len(w)):
            blocks = [Resnet(w[i - 1] if i > 0 else w[0], w[i], **kw), Resnet(w[i], w[i], **kw)]
            if i > 0:
                blocks.append(Resnet(w[i], w[i], **kw))
            stages.append(fl.Chain(blocks))
        stages[0].insert(1, _attention(w[0], kw))
        for stage in stages[1:4]:
            stage.insert(-1, fl.Upsample(channels=stage.layer(-1, Resnet).out_channels, upsample_factor=2, **kw))
        super().__init__(
            fl.Conv2d(self.latent_dim, self.latent_dim, kernel_size=1, **kw),
            fl.Conv2d(self.latent_dim, w[0], kernel_size=3, padding=1, **kw),
            fl.Chain(*stages),
            fl.Chain(fl.GroupNorm(channels=w[-1], num_groups=32, eps=1e-6, **kw), fl.SiLU(), fl.Conv2d(w[-1], self.output_channels, kernel_size=3, padding=1, **kw)),
        )


class LatentDiffusionAutoencoder(fl.Chain):
    encoder_scale = 0.18125

    def __init__(self, device: Any = None, dtype: Any = None) -> None:
        super().__init__(Encoder(device=device, dtype=dtype), Decoder(device=device, dtype=dtype))

    def encode(self, x: Tensor) -> Tensor:
        return self.encoder_scale * self[0](x)

    def decode(self, x: Tensor) -> Tensor:
        return self[1](x / self.encoder_scale)


class SDXLAutoencoder(LatentDiffusionAutoencoder):
    encoder_scale: float = 0.13025
