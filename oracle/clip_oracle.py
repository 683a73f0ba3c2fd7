"""ORACLE (test infrastructure, never shipped or measured): CPU float32 restatement of SDXL's prompt encoder (SURVEY.md
section 8(f) next-2) over a flat state dict keyed like the reference's `DoubleTextEncoder`, starting from token ids.
Pinned to the real reference through tests/golden/double_text_encoder.safetensors (oracle/make_golden_clip.py).
Citations: /root/reference/src/refiners/foundationals/clip/text_encoder.py, clip/common.py,
foundationals/latent_diffusion/stable_diffusion_xl/text_encoder.py, fluxion/layers/attentions.py."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import Tensor

SD = dict[str, Tensor]


def causal_self_attention(sd: SD, p: str, x: Tensor, heads: int) -> Tensor:
    """fl.SelfAttention(is_causal=True) (attentions.py:319-385): q, k, v, out projections with bias; key j is visible to
    query i iff j <= i (attentions.py:15-34)."""
    B, L, C = x.shape
    d = C // heads
    q, k, v = (F.linear(x, sd[f"{p}.Distribute.Linear_{i}.weight"], sd[f"{p}.Distribute.Linear_{i}.bias"]).reshape(B, L, heads, d).transpose(1, 2) for i in (1, 2, 3))
    logits = q @ k.transpose(-1, -2) / math.sqrt(d)
    mask = torch.ones(L, L, dtype=torch.bool).tril()
    att = torch.softmax(logits.masked_fill(~mask, float("-inf")), dim=-1) @ v
    return F.linear(att.transpose(1, 2).reshape(B, L, C), sd[f"{p}.Linear.weight"], sd[f"{p}.Linear.bias"])


def transformer_layer(sd: SD, p: str, x: Tensor, heads: int, quick_gelu: bool) -> Tensor:
    """TransformerLayer (clip/text_encoder.py:25-69): two pre-LN residual branches, eps 1e-5."""
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), sd[f"{p}.Residual_1.LayerNorm.weight"], sd[f"{p}.Residual_1.LayerNorm.bias"], 1e-5)
    x = x + causal_self_attention(sd, f"{p}.Residual_1.SelfAttention", h, heads)
    h = F.layer_norm(x, (C,), sd[f"{p}.Residual_2.LayerNorm.weight"], sd[f"{p}.Residual_2.LayerNorm.bias"], 1e-5)
    f = F.linear(h, sd[f"{p}.Residual_2.FeedForward.Linear_1.weight"], sd[f"{p}.Residual_2.FeedForward.Linear_1.bias"])
    f = f * torch.sigmoid(1.702 * f) if quick_gelu else F.gelu(f)  # GeLUApproximation.SIGMOID (activations.py:83-118)
    return x + F.linear(f, sd[f"{p}.Residual_2.FeedForward.Linear_2.weight"], sd[f"{p}.Residual_2.FeedForward.Linear_2.bias"])


def embed(sd: SD, p: str, tokens: Tensor) -> Tensor:
    """Sum(TokenEncoder, PositionalEncoder) (clip/text_encoder.py:122-135, common.py:7-31)."""
    return sd[f"{p}.Sum.TokenEncoder.weight"][tokens.long()] + sd[f"{p}.Sum.PositionalEncoder.Embedding.weight"][: tokens.shape[1]][None]


@torch.no_grad()
def double_text_encoder(sd: SD, tokens_l: Tensor, tokens_g: Tensor, end_of_text_token_id: int = 49407) -> tuple[Tensor, Tensor]:
    """DoubleTextEncoder (xl/text_encoder.py:61-101): CLIP-L without its last layer and final LN (`text_encoder_l[:-2]`),
    CLIP-G's first 31 layers (`target[1:-2]`) for the embedding; CLIP-G's last layer + final LN + bias-free projection,
    read at the first end-of-text token, for the pooled embedding (TextEncoderWithPooling, :14-58)."""
    pl = "Parallel.CLIPTextEncoderL"
    x = embed(sd, pl, tokens_l)
    for i in range(1, 12):
        x = transformer_layer(sd, f"{pl}.TransformerLayer_{i}", x, 12, quick_gelu=True)
    pg = "Parallel.TextEncoderWithPooling.CLIPTextEncoderG"
    y = embed(sd, pg, tokens_g)
    for i in range(1, 32):
        y = transformer_layer(sd, f"{pg}.TransformerLayer_{i}", y, 20, quick_gelu=False)
    pc = "Parallel.TextEncoderWithPooling.Parallel.Chain"
    z = transformer_layer(sd, f"{pc}.CLIPTextEncoderG.TransformerLayer", y, 20, quick_gelu=False)
    z = F.layer_norm(z, (z.shape[-1],), sd[f"{pc}.CLIPTextEncoderG.LayerNorm.weight"], sd[f"{pc}.CLIPTextEncoderG.LayerNorm.bias"], 1e-5)
    z = F.linear(z, sd[f"{pc}.Linear.weight"])
    eot = [int((row == end_of_text_token_id).nonzero()[0]) for row in tokens_g]
    pooled = torch.stack([z[i, e] for i, e in enumerate(eot)])
    return torch.cat((x, y), dim=-1), pooled


# ------------------------------------------------------------------------------------------------ image prompt side
def self_attention(sd: SD, p: str, x: Tensor, heads: int) -> Tensor:
    """fl.SelfAttention, bidirectional (attentions.py:319-385)."""
    B, L, C = x.shape
    d = C // heads
    q, k, v = (F.linear(x, sd[f"{p}.Distribute.Linear_{i}.weight"], sd[f"{p}.Distribute.Linear_{i}.bias"]).reshape(B, L, heads, d).transpose(1, 2) for i in (1, 2, 3))
    att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), dim=-1) @ v
    return F.linear(att.transpose(1, 2).reshape(B, L, C), sd[f"{p}.Linear.weight"], sd[f"{p}.Linear.bias"])


@torch.no_grad()
def clip_image_encoder(sd: SD, image: Tensor, heads: int = 16, layers: int = 32, patch: int = 14) -> Tensor:
    """CLIPImageEncoder (clip/image_encoder.py:134-197): patch conv (no bias) -> [cls ; patches] + positions -> LN ->
    pre-LN transformer layers (erf GELU) -> class token -> LN -> bias-free projection."""
    B = image.shape[0]
    w = sd["ViTEmbeddings.Concatenate.Chain.PatchEncoder.Conv2d.weight"]
    C = w.shape[0]
    patches = F.conv2d(image, w, stride=patch).permute(0, 2, 3, 1).reshape(B, -1, C)
    cls = sd["ViTEmbeddings.Concatenate.ClassToken.Parameter.weight"].expand(B, 1, C)
    x = torch.cat((cls, patches), dim=1)
    x = x + sd["ViTEmbeddings.Residual.PositionalEncoder.Embedding.weight"][: x.shape[1]][None]
    x = F.layer_norm(x, (C,), sd["LayerNorm_1.weight"], sd["LayerNorm_1.bias"], 1e-5)
    for i in range(1, layers + 1):
        p = f"Chain.TransformerLayer_{i}"
        h = F.layer_norm(x, (C,), sd[f"{p}.Residual_1.LayerNorm.weight"], sd[f"{p}.Residual_1.LayerNorm.bias"], 1e-5)
        x = x + self_attention(sd, f"{p}.Residual_1.SelfAttention", h, heads)
        h = F.layer_norm(x, (C,), sd[f"{p}.Residual_2.LayerNorm.weight"], sd[f"{p}.Residual_2.LayerNorm.bias"], 1e-5)
        f = F.gelu(F.linear(h, sd[f"{p}.Residual_2.FeedForward.Linear_1.weight"], sd[f"{p}.Residual_2.FeedForward.Linear_1.bias"]))
        x = x + F.linear(f, sd[f"{p}.Residual_2.FeedForward.Linear_2.weight"], sd[f"{p}.Residual_2.FeedForward.Linear_2.bias"])
    y = F.layer_norm(x[:, 0], (C,), sd["LayerNorm_2.weight"], sd["LayerNorm_2.bias"], 1e-5)
    return F.linear(y, sd["Linear.weight"])


@torch.no_grad()
def image_prompt_tokens(psd: SD, embedding: Tensor, num_tokens: int = 4) -> Tensor:
    """cat(image_proj(0), image_proj(embedding)) -- ImageProjection = Linear -> Reshape(num_tokens, C) -> LayerNorm
    (image_prompt.py:24-45), negative / conditional as in IPAdapter.compute_clip_image_embedding (:457-510)."""
    def proj(e: Tensor) -> Tensor:
        t = F.linear(e, psd["Linear.weight"], psd["Linear.bias"]).reshape(e.shape[0], num_tokens, -1)
        return F.layer_norm(t, (t.shape[-1],), psd["LayerNorm.weight"], psd["LayerNorm.bias"], 1e-5)

    return torch.cat((proj(torch.zeros_like(embedding)), proj(embedding)))
