"""ORACLE (test infrastructure, never shipped or measured): CPU float32 restatement of refiners' SegmentAnything ViT image
encoder (BASELINE.json config 5) as plain functions over a flat state dict keyed like the reference's.  Pinned to the
real reference through tests/golden/sam_vit_h.safetensors (written by oracle/make_golden_sam.py from
finegrain-ai/refiners itself).  Citations: /root/reference/src/refiners/foundationals/segment_anything/image_encoder.py.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor

SD = dict[str, Tensor]


def _rel_index(size: int) -> Tensor:
    i = torch.arange(size)
    return i[:, None] - i[None, :] + size - 1  # image_encoder.py:101-103


def rel_pos_attention(qkv: Tensor, hor: Tensor, ver: Tensor, heads: int) -> Tensor:
    """RelativePositionAttention.forward (image_encoder.py:82-127) on (B, H, W, 3C)."""
    B, H, W, C3 = qkv.shape
    d = C3 // 3 // heads
    x = qkv.reshape(B, H * W, 3, heads, d).permute(2, 0, 3, 1, 4).reshape(3, B * heads, H * W, d)
    q, k, v = x.unbind(0)
    q4 = q.reshape(q.shape[0], H, W, d)
    rel_hor = torch.einsum("bhwc,wkc->bhwk", q4, hor[_rel_index(W)]).unsqueeze(-2)
    rel_ver = torch.einsum("bhwc,hkc->bhwk", q4, ver[_rel_index(H)]).unsqueeze(-1)
    att = (q * d ** -0.5) @ k.transpose(-2, -1)
    att = ((att.reshape(-1, H, W, H, W) + rel_ver) + rel_hor).reshape(att.shape).softmax(dim=-1)
    out = att @ v
    return out.reshape(B, heads, H, W, d).permute(0, 2, 3, 1, 4).reshape(B, H, W, heads * d)


def transformer_layer(sd: SD, p: str, x: Tensor, heads: int, window: int | None) -> Tensor:
    """TransformerLayer (image_encoder.py:239-283): windowed or global attention block + MLP, both residual."""
    B, H, W, C = x.shape
    a = f"{p}.Residual_1"
    h = F.layer_norm(x, (C,), sd[f"{a}.LayerNorm.weight"], sd[f"{a}.LayerNorm.bias"], 1e-6)
    if window is not None:  # WindowPartition (image_encoder.py:202-219): zero pad to multiples of the window, split
        ph, pw = (window - H % window) % window, (window - W % window) % window
        h = F.pad(h, (0, 0, 0, pw, 0, ph))
        Hp, Wp = H + ph, W + pw
        h = h.view(B, Hp // window, window, Wp // window, window, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, window, window, C)
    f = f"{a}.FusedSelfAttention"
    qkv = F.linear(h, sd[f"{f}.Linear_1.weight"], sd[f"{f}.Linear_1.bias"])
    o = rel_pos_attention(qkv, sd[f"{f}.RelativePositionAttention.horizontal_embedding"], sd[f"{f}.RelativePositionAttention.vertical_embedding"], heads)
    o = F.linear(o, sd[f"{f}.Linear_2.weight"], sd[f"{f}.Linear_2.bias"])
    if window is not None:  # WindowMerge (image_encoder.py:222-236)
        o = o.view(B, Hp // window, Wp // window, window, window, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)[:, :H, :W]
    x = x + o
    a = f"{p}.Residual_2"
    h = F.layer_norm(x, (C,), sd[f"{a}.LayerNorm.weight"], sd[f"{a}.LayerNorm.bias"], 1e-6)
    h = F.gelu(F.linear(h, sd[f"{a}.FeedForward.Linear_1.weight"], sd[f"{a}.FeedForward.Linear_1.bias"]), approximate="none")
    return x + F.linear(h, sd[f"{a}.FeedForward.Linear_2.weight"], sd[f"{a}.FeedForward.Linear_2.bias"])


def layer_norm_2d(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-6) -> Tensor:
    """fl.LayerNorm2d (fluxion/layers/norm.py:96-140): normalise over the channel dimension of an NCHW tensor."""
    mu = x.mean(1, keepdim=True)
    var = (x - mu).pow(2).mean(1, keepdim=True)
    return w[None, :, None, None] * ((x - mu) / torch.sqrt(var + eps)) + b[None, :, None, None]


@torch.no_grad()
def sam_vit(sd: SD, image: Tensor, num_layers: int = 32, heads: int = 16, global_layers: tuple[int, ...] = (7, 15, 23, 31),
            window: int = 14) -> tuple[Tensor, Tensor]:
    """SAMViT.forward (image_encoder.py:317-368) -> (neck output (B, 256, 64, 64), early ViT embedding = output of the first
    global-attention layer (B, 64, 64, C), what HQ-SAM's SAMViTAdapter stores, hq_sam.py:230-264)."""
    x = F.conv2d(image, sd["PatchEncoder.Conv2d.weight"], sd["PatchEncoder.Conv2d.bias"], stride=16).permute(0, 2, 3, 1)
    x = x + sd["PositionalEncoder.Parameter.weight"]
    early = None
    for i in range(num_layers):
        x = transformer_layer(sd, f"Transformer.TransformerLayer_{i + 1}", x, heads, None if i in global_layers else window)
        if early is None and i in global_layers:
            early = x
    y = x.permute(0, 3, 1, 2)
    y = F.conv2d(y, sd["Neck.Conv2d_1.weight"])
    y = layer_norm_2d(y, sd["Neck.LayerNorm2d_1.weight"], sd["Neck.LayerNorm2d_1.bias"])
    y = F.conv2d(y, sd["Neck.Conv2d_2.weight"], padding=1)
    y = layer_norm_2d(y, sd["Neck.LayerNorm2d_2.weight"], sd["Neck.LayerNorm2d_2.bias"])
    return y, early
