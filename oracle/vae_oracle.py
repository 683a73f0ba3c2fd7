"""ORACLE (test infrastructure, never shipped or measured): CPU float32 restatement of the latent-diffusion autoencoder's
Decoder over a flat state dict keyed like the reference's `LatentDiffusionAutoencoder`.
Pinned to the real reference through tests/golden/vae_decode.safetensors (oracle/make_golden_vae.py).
Citations: /root/reference/src/refiners/foundationals/latent_diffusion/auto_encoder.py."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import Tensor

SD = dict[str, Tensor]


def resnet(sd: SD, p: str, x: Tensor) -> Tensor:
    """Resnet = Sum(shortcut, Chain(GN, SiLU, conv3x3, GN, SiLU, conv3x3)) (auto_encoder.py:83-140)."""
    h = F.silu(F.group_norm(x, 32, sd[f"{p}.Chain.GroupNorm_1.weight"], sd[f"{p}.Chain.GroupNorm_1.bias"], 1e-5))
    h = F.conv2d(h, sd[f"{p}.Chain.Conv2d_1.weight"], sd[f"{p}.Chain.Conv2d_1.bias"], padding=1)
    h = F.silu(F.group_norm(h, 32, sd[f"{p}.Chain.GroupNorm_2.weight"], sd[f"{p}.Chain.GroupNorm_2.bias"], 1e-5))
    h = F.conv2d(h, sd[f"{p}.Chain.Conv2d_2.weight"], sd[f"{p}.Chain.Conv2d_2.bias"], padding=1)
    s = F.conv2d(x, sd[f"{p}.Conv2d.weight"], sd[f"{p}.Conv2d.bias"]) if f"{p}.Conv2d.weight" in sd else x
    return s + h


def attention(sd: SD, p: str, x: Tensor) -> Tensor:
    """Residual(GroupNorm(eps 1e-6), SelfAttention2d(1 head)) (auto_encoder.py:221-225; fluxion/layers/attentions.py:388-470)."""
    B, C, H, W = x.shape
    h = F.group_norm(x, 32, sd[f"{p}.GroupNorm.weight"], sd[f"{p}.GroupNorm.bias"], 1e-6)
    t = h.reshape(B, C, H * W).transpose(1, 2)
    a = f"{p}.SelfAttention2d"
    q = F.linear(t, sd[f"{a}.Distribute.Linear_1.weight"], sd[f"{a}.Distribute.Linear_1.bias"])
    k = F.linear(t, sd[f"{a}.Distribute.Linear_2.weight"], sd[f"{a}.Distribute.Linear_2.bias"])
    v = F.linear(t, sd[f"{a}.Distribute.Linear_3.weight"], sd[f"{a}.Distribute.Linear_3.bias"])
    att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(C), dim=-1) @ v
    o = F.linear(att, sd[f"{a}.Linear.weight"], sd[f"{a}.Linear.bias"])
    return x + o.transpose(1, 2).reshape(B, C, H, W)


@torch.no_grad()
def vae_decode(sd: SD, latents: Tensor, encoder_scale: float = 0.13025) -> Tensor:
    """LatentDiffusionAutoencoder.decode (auto_encoder.py:322-325) = Decoder(latents / encoder_scale) (143-207)."""
    p = "Decoder"
    x = latents / encoder_scale
    x = F.conv2d(x, sd[f"{p}.Conv2d_1.weight"], sd[f"{p}.Conv2d_1.bias"])
    x = F.conv2d(x, sd[f"{p}.Conv2d_2.weight"], sd[f"{p}.Conv2d_2.bias"], padding=1)
    for i in range(1, 6):
        s = f"{p}.Chain_1.Chain_{i}"
        x = resnet(sd, f"{s}.Resnet_1", x)
        if i == 1:
            x = attention(sd, f"{s}.Residual", x)
        x = resnet(sd, f"{s}.Resnet_2", x)
        if i > 1:
            x = resnet(sd, f"{s}.Resnet_3", x)
        if f"{s}.Upsample.Conv2d.weight" in sd:  # fl.Upsample(upsample_factor=2): nearest x2, then conv3x3 (sampling.py:112-161)
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            x = F.conv2d(x, sd[f"{s}.Upsample.Conv2d.weight"], sd[f"{s}.Upsample.Conv2d.bias"], padding=1)
    x = F.silu(F.group_norm(x, 32, sd[f"{p}.Chain_2.GroupNorm.weight"], sd[f"{p}.Chain_2.GroupNorm.bias"], 1e-6))
    return F.conv2d(x, sd[f"{p}.Chain_2.Conv2d.weight"], sd[f"{p}.Chain_2.Conv2d.bias"], padding=1)


@torch.no_grad()
def vae_encode(sd: SD, image: Tensor, encoder_scale: float = 0.13025) -> Tensor:
    """LatentDiffusionAutoencoder.encode (auto_encoder.py:317-320) = encoder_scale * Encoder(image) (83-141): image in [-1, 1],
    (B, 3, 8h, 8w) -> latents (B, 4, h, w).  Downsample(padding=0) = F.pad(x, (0, 1, 0, 1)) + stride-2 conv (sampling.py:41-109)."""
    p = "Encoder"
    x = F.conv2d(image, sd[f"{p}.Conv2d.weight"], sd[f"{p}.Conv2d.bias"], padding=1)
    for i in range(1, 6):
        s = f"{p}.Chain_1.Chain_{i}"
        x = resnet(sd, f"{s}.Resnet_1", x)
        if i == 5:
            x = attention(sd, f"{s}.Residual", x)
        x = resnet(sd, f"{s}.Resnet_2", x)
        if f"{s}.Downsample.Conv2d.weight" in sd:
            x = F.conv2d(F.pad(x, (0, 1, 0, 1)), sd[f"{s}.Downsample.Conv2d.weight"], sd[f"{s}.Downsample.Conv2d.bias"], stride=2)
    x = F.silu(F.group_norm(x, 32, sd[f"{p}.Chain_2.GroupNorm.weight"], sd[f"{p}.Chain_2.GroupNorm.bias"], 1e-6))
    x = F.conv2d(x, sd[f"{p}.Chain_2.Conv2d.weight"], sd[f"{p}.Chain_2.Conv2d.bias"], padding=1)
    x = F.conv2d(x, sd[f"{p}.Chain_3.Conv2d.weight"], sd[f"{p}.Chain_3.Conv2d.bias"])
    return encoder_scale * x[:, :4]
