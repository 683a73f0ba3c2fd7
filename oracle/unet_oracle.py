"""ORACLE (test infrastructure, never shipped or measured): CPU float32 restatement of refiners' SDXL / SD1.5 UNet
forward, adapters and sampling step as plain functions over a flat state dict.

It deliberately shares NO code with refiners_amd/ (no Chain tree, no context store): it is the independent checker the
HIP path and the host mirror are compared with.  It is itself pinned to the real reference: oracle/make_golden.py
loads the same synthetic weights into finegrain-ai/refiners (imported from /root/reference in the build container),
runs the reference's own Chain forward on CPU and stores the outputs under tests/golden/; tests/test_oracle_golden.py
requires this file to reproduce them.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.

Weights are keyed exactly like the reference's bare-model `state_dict()` (e.g.
`DownBlocks.Chain_5.SDXLCrossAttention.Chain_2.CrossAttentionBlock_1.Residual_2.Attention.Distribute.Linear_2.weight`).
Adapters are passed as data:
  loras   = [{"scale": s, "pairs": {<bare Linear/Conv2d key prefix>: (down, up)}}, ...]
  ip      = {"scale": s, "tokens": (B, T, 2048), "kv": {<"...Residual_2.Attention" prefix>: (Wk', Wv')}}
  control = [{"name":..., "scale": s, "condition": (B,3,8H,8W), "encoder": {ConditionEncoder keys}, "zero": [(w, b)] * 10,
              "loras": [...same format as loras, applied inside the control branch only...]}, ...]

Every function cites the reference lines it restates (paths relative to /root/reference/src/refiners).
"""
from __future__ import annotations

import math
from typing import Any

import torch
import torch.nn.functional as F
from torch import Tensor

SD = dict[str, Tensor]

# (kind, ...) per block, identical to the reference constructors
SDXL_DOWN = [  # foundationals/latent_diffusion/stable_diffusion_xl/unet.py:115-170
    ["conv_in"], ["res"], ["res"], ["down"], ["res", "attn"], ["res", "attn"], ["down"], ["res", "attn"], ["res", "attn"],
]
SDXL_UP = [  # xl/unet.py:173-235
    ["res", "attn"], ["res", "attn"], ["res", "attn", "up"], ["res", "attn"], ["res", "attn"], ["res", "attn", "up"], ["res"], ["res"], ["res"],
]
SD1_DOWN = [  # stable_diffusion_1/unet.py:48-99
    ["conv_in"], ["res", "attn"], ["res", "attn"], ["down"], ["res", "attn"], ["res", "attn"], ["down"], ["res", "attn"], ["res", "attn"],
    ["down"], ["res"], ["res"],
]
SD1_UP = [  # sd1/unet.py:102-155
    ["res"], ["res"], ["res", "up"], ["res", "attn"], ["res", "attn"], ["res", "attn", "up"], ["res", "attn"], ["res", "attn"],
    ["res", "attn", "up"], ["res", "attn"], ["res", "attn"], ["res", "attn"],
]


class _Net:
    """Weight lookup + LoRA application for one branch (main UNet or a ControlLora copy)."""

    def __init__(self, sd: SD, loras: list[dict[str, Any]] | None) -> None:
        self.sd = sd
        self.loras = loras or []

    def has(self, prefix: str) -> bool:
        return f"{prefix}.weight" in self.sd

    def linear(self, prefix: str, x: Tensor) -> Tensor:
        """fl.Linear (fluxion/layers/linear.py:9-56) wrapped by LoraAdapter = Sum(target, loras...)
        (fluxion/adapters/lora.py:383-397); lora = Multiply(scale)(up(down(x))) (lora.py:51-54, basics.py:404-405)."""
        y = F.linear(x, self.sd[f"{prefix}.weight"], self.sd.get(f"{prefix}.bias"))
        for lr in self.loras:
            pair = lr["pairs"].get(prefix)
            if pair is not None:
                y = y + (lr["scale"] * F.linear(F.linear(x, pair[0]), pair[1]) + 0.0)
        return y

    def conv(self, prefix: str, x: Tensor, stride: int = 1, padding: int = 0) -> Tensor:
        """fl.Conv2d (fluxion/layers/conv.py:6-61) + Conv2dLora (lora.py:269-380: down k x k with the target's stride,
        up 1x1 or 3x3)."""
        y = F.conv2d(x, self.sd[f"{prefix}.weight"], self.sd.get(f"{prefix}.bias"), stride=stride, padding=padding)
        for lr in self.loras:
            pair = lr["pairs"].get(prefix)
            if pair is not None:
                d = F.conv2d(x, pair[0], None, stride=stride, padding=1 if pair[0].shape[2] == 3 else 0)
                u = F.conv2d(d, pair[1], None, stride=1, padding=1 if pair[1].shape[2] == 3 else 0)
                y = y + (lr["scale"] * u + 0.0)
        return y

    def group_norm(self, prefix: str, x: Tensor, eps: float) -> Tensor:
        return F.group_norm(x, 32, self.sd[f"{prefix}.weight"], self.sd[f"{prefix}.bias"], eps)  # layers/norm.py:49-93

    def layer_norm(self, prefix: str, x: Tensor) -> Tensor:
        w = self.sd[f"{prefix}.weight"]
        return F.layer_norm(x, (w.shape[0],), w, self.sd[f"{prefix}.bias"], 1e-5)  # layers/norm.py:13-46


def sinusoidal_embedding(x: Tensor, dim: int) -> Tensor:
    """foundationals/latent_diffusion/range_adapter.py:11-22."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half
    angle = x.unsqueeze(1).float() * torch.exp(exponent).unsqueeze(0)
    return torch.cat([torch.cos(angle), torch.sin(angle)], dim=-1)


def sdpa(q: Tensor, k: Tensor, v: Tensor, heads: int) -> Tensor:
    """ScaledDotProductAttention: head split, softmax(QK^T / sqrt(d)) V, head merge (layers/attentions.py:15-34, 157-202).
    Written out with explicit matmul/softmax (the `non_optimized` form of attentions.py:37-57) so that it does not depend
    on which fused SDPA backend torch picks."""
    B, Lq, C = q.shape
    d = C // heads
    qh = q.reshape(B, Lq, heads, d).transpose(1, 2)
    kh = k.reshape(B, k.shape[1], heads, d).transpose(1, 2)
    vh = v.reshape(B, v.shape[1], heads, d).transpose(1, 2)
    att = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(d), dim=-1)
    return (att @ vh).transpose(1, 2).reshape(B, Lq, C)


def residual_block(net: _Net, p: str, x: Tensor, temb: Tensor) -> Tensor:
    """ResidualBlock with the RangeAdapter2d injected on its first conv
    (foundationals/latent_diffusion/unet.py:6-51, range_adapter.py:47-86, xl/unet.py:286-295)."""
    h = F.silu(net.group_norm(f"{p}.Chain.GroupNorm_1", x, 1e-5))
    h = net.conv(f"{p}.Chain.RangeAdapter2d.Conv2d", h, padding=1)
    t = net.linear(f"{p}.Chain.RangeAdapter2d.Chain.Linear", F.silu(temb))
    h = h + t.reshape(t.shape[0], -1, 1, 1)
    h = F.silu(net.group_norm(f"{p}.Chain.GroupNorm_2", h, 1e-5))
    h = net.conv(f"{p}.Chain.Conv2d", h, padding=1)
    shortcut = net.conv(f"{p}.Conv2d", x) if net.has(f"{p}.Conv2d") else x
    return h + shortcut


def cross_attention_block(net: _Net, p: str, x: Tensor, text: Tensor, heads: int, ip: dict[str, Any] | None) -> Tensor:
    """CrossAttentionBlock (foundationals/latent_diffusion/cross_attention.py:25-73); with an IP-Adapter the text SDPA
    becomes Sum(SDPA, scale * SDPA(q, Wk' img, Wv' img)) (image_prompt.py:237-309)."""
    a = f"{p}.Residual_1"
    h = net.layer_norm(f"{a}.LayerNorm", x)
    q = net.linear(f"{a}.SelfAttention.Distribute.Linear_1", h)
    k = net.linear(f"{a}.SelfAttention.Distribute.Linear_2", h)
    v = net.linear(f"{a}.SelfAttention.Distribute.Linear_3", h)
    x = x + net.linear(f"{a}.SelfAttention.Linear", sdpa(q, k, v, heads))
    a = f"{p}.Residual_2"
    h = net.layer_norm(f"{a}.LayerNorm", x)
    q = net.linear(f"{a}.Attention.Distribute.Linear_1", h)
    k = net.linear(f"{a}.Attention.Distribute.Linear_2", text)
    v = net.linear(f"{a}.Attention.Distribute.Linear_3", text)
    o = sdpa(q, k, v, heads)
    if ip is not None:
        wk, wv = ip["kv"][f"{a}.Attention"]
        tok = ip["tokens"]
        o = o + (ip["scale"] * sdpa(q, F.linear(tok, wk), F.linear(tok, wv), heads) + 0.0)
    x = x + net.linear(f"{a}.Attention.Linear", o)
    a = f"{p}.Residual_3"
    h = net.layer_norm(f"{a}.LayerNorm", x)
    h = net.linear(f"{a}.Linear_1", h)
    val, gate = h.chunk(2, dim=-1)
    h = val * F.gelu(gate, approximate="none")  # fl.GLU(fl.GeLU()), layers/activations.py:83-160
    return x + net.linear(f"{a}.Linear_2", h)


def cross_attention_2d(net: _Net, p: str, x: Tensor, text: Tensor, heads: int, linear_proj: bool, ip: dict[str, Any] | None) -> Tensor:
    """CrossAttentionBlock2d (cross_attention.py:92-175): GN(eps 1e-6), NCHW -> tokens, proj, blocks, proj, back, + x."""
    B, C, H, W = x.shape
    h = net.group_norm(f"{p}.Chain_1.GroupNorm", x, 1e-6)
    if linear_proj:
        h = net.linear(f"{p}.Chain_1.Linear", h.flatten(2).transpose(1, 2))
    else:
        h = net.conv(f"{p}.Chain_1.Conv2d", h).flatten(2).transpose(1, 2)
    if net.has(f"{p}.Chain_2.CrossAttentionBlock.Residual_1.LayerNorm"):  # a single layer keeps the bare class name
        names = ["CrossAttentionBlock"]
    else:
        names, i = [], 1
        while net.has(f"{p}.Chain_2.CrossAttentionBlock_{i}.Residual_1.LayerNorm"):
            names.append(f"CrossAttentionBlock_{i}")
            i += 1
    for name in names:
        h = cross_attention_block(net, f"{p}.Chain_2.{name}", h, text, heads, ip)
    if linear_proj:
        h = net.linear(f"{p}.Chain_3.Linear", h).transpose(1, 2).reshape(B, C, H, W)
    else:
        h = net.conv(f"{p}.Chain_3.Conv2d", h.transpose(1, 2).reshape(B, C, H, W))
    return h + x


def _heads(channels: int, family: str) -> int:
    if family == "sd1":
        return 8  # sd1/unet.py:41
    return 10 if channels == 640 else 20  # xl/unet.py:127-166


def _stage(net: _Net, p: str, kinds: list[str], family: str, x: Tensor, temb: Tensor, text: Tensor, ip, shapes: list) -> Tensor:
    attn_name = "SDXLCrossAttention" if family == "sdxl" else "CLIPLCrossAttention"
    for kind in kinds:
        if kind == "conv_in":
            x = net.conv(f"{p}.Conv2d", x, padding=1)
        elif kind == "res":
            x = residual_block(net, f"{p}.ResidualBlock", x, temb)
        elif kind == "attn":
            x = cross_attention_2d(net, f"{p}.{attn_name}", x, text, _heads(x.shape[1], family), family == "sdxl", ip)
        elif kind == "down":  # fl.Downsample(padding=1): records H x W, stride-2 3x3 conv (layers/sampling.py:41-109)
            shapes.append(x.shape[2:])
            x = net.conv(f"{p}.Downsample.Conv2d", x, stride=2, padding=1)
        elif kind == "up":  # fl.Upsample: nearest interpolate to the popped shape, 3x3 conv (sampling.py:112-161)
            x = F.interpolate(x, size=shapes.pop(), mode="nearest")
            x = net.conv(f"{p}.Upsample.Conv2d", x, padding=1)
    return x


def timestep_embedding_sdxl(net: _Net, timestep: Tensor, pooled: Tensor, time_ids: Tensor) -> Tensor:
    """TimestepEncoder = Sum(RangeEncoder(timestep), TextTimeEmbedding) (xl/unet.py:20-90)."""
    p = "TimestepEncoder.Sum"
    t = sinusoidal_embedding(timestep, 320)
    t = net.linear(f"{p}.Chain.RangeEncoder.Linear_2", F.silu(net.linear(f"{p}.Chain.RangeEncoder.Linear_1", t)))
    ids = sinusoidal_embedding(time_ids.unsqueeze(-1), 256).reshape(time_ids.shape[0], -1)
    tt = torch.cat([pooled, ids], dim=1)
    tt = net.linear(f"{p}.TextTimeEmbedding.Linear_2", F.silu(net.linear(f"{p}.TextTimeEmbedding.Linear_1", tt)))
    return t + tt


def condition_encoder(enc: SD, x: Tensor) -> Tensor:
    """ConditionEncoder (xl/control_lora.py:14-87)."""
    x = F.silu(F.conv2d(x, enc["Chain_1.Conv2d.weight"], enc["Chain_1.Conv2d.bias"], padding=1))
    for i in (2, 3, 4):
        x = F.silu(F.conv2d(x, enc[f"Chain_{i}.Conv2d_1.weight"], enc[f"Chain_{i}.Conv2d_1.bias"], padding=1))
        x = F.silu(F.conv2d(x, enc[f"Chain_{i}.Conv2d_2.weight"], enc[f"Chain_{i}.Conv2d_2.bias"], stride=2, padding=1))
    return F.conv2d(x, enc["Conv2d.weight"], enc["Conv2d.bias"], padding=1)


def control_lora_residuals(sd: SD, ctl: dict[str, Any], x: Tensor, timestep, text, pooled, time_ids) -> list[Tensor]:
    """ControlLora (xl/control_lora.py:144-248): the encoder half run a second time on shared weights (+ its own
    LoRAs), condition added after the first conv, every block output -> 1x1 ZeroConvolution * scale.  (An IP-Adapter
    cannot coexist inside the copy: the reference refuses to structural_copy Chain adapters, adapter.py:106-108.)"""
    net = _Net(sd, ctl.get("loras"))
    ip = None
    temb = timestep_embedding_sdxl(net, timestep, pooled, time_ids)
    out: list[Tensor] = []
    shapes: list = []
    for n, kinds in enumerate(SDXL_DOWN):
        x = _stage(net, f"DownBlocks.Chain_{n + 1}", kinds, "sdxl", x, temb, text, ip, shapes)
        w, b = ctl["zero"][n]
        out.append(ctl["scale"] * F.conv2d(x, w, b) + 0.0)
        if n == 0:  # appended AFTER the first stage's accumulator slot (control_lora.py:190-202), so zero-conv 0 does not see it
            x = x + condition_encoder(ctl["encoder"], ctl["condition"])
    x = residual_block(net, "MiddleBlock.ResidualBlock_1", x, temb)
    x = cross_attention_2d(net, "MiddleBlock.SDXLCrossAttention", x, text, 20, True, ip)
    x = residual_block(net, "MiddleBlock.ResidualBlock_2", x, temb)
    w, b = ctl["zero"][9]
    out.append(ctl["scale"] * F.conv2d(x, w, b) + 0.0)
    return out


T2I_SDXL_BLOCKS = (3, 5, 8)  # SDXLT2IAdapter.residual_indices


@torch.no_grad()
def t2i_condition_encoder_xl(sd: SD, picture: Tensor) -> list[Tensor]:
    """ConditionEncoderXL (latent_diffusion/t2i_adapter.py:132-163): PixelUnshuffle(16) -> conv3x3 -> four stages of
    [avg-pool 2 (third stage only)] -> [1x1 shortcut when the width changes] -> 2 x (x + conv1x1(relu(conv3x3(x))))."""
    x = F.conv2d(F.pixel_unshuffle(picture, 16), sd["Conv2d.weight"], sd["Conv2d.bias"], padding=1)
    feats = []
    for i in range(1, 5):
        p = f"StatefulResidualBlocks_{i}.ResidualBlocks"
        if i == 3:
            x = F.avg_pool2d(x, 2, 2)
        if f"{p}.Conv2d.weight" in sd:
            x = F.conv2d(x, sd[f"{p}.Conv2d.weight"], sd[f"{p}.Conv2d.bias"])
        for j in (1, 2):
            q = f"{p}.Chain.ResidualBlock_{j}"
            h = F.relu(F.conv2d(x, sd[f"{q}.Conv2d_1.weight"], sd[f"{q}.Conv2d_1.bias"], padding=1))
            x = x + F.conv2d(h, sd[f"{q}.Conv2d_2.weight"], sd[f"{q}.Conv2d_2.bias"])
        feats.append(x)
    return feats


@torch.no_grad()
def sdxl_unet(sd: SD, x: Tensor, timestep: Tensor, text: Tensor, pooled: Tensor, time_ids: Tensor,
              loras: list | None = None, ip: dict | None = None, control: list | None = None, t2i: dict | None = None) -> Tensor:
    """SDXLUNet.forward (xl/unet.py:258-351).  x (B,4,H,W); timestep (1,) or (B,); text (B,77,2048); pooled (B,1280);
    time_ids (B,6).  Skip handling = ResidualAccumulator / ResidualConcatenator (latent_diffusion/unet.py:54-79)."""
    residuals: list[Any] = [0.0] * 10
    for ctl in control or []:  # each ControlLora sits at unet[0] and pre-populates the residual slots
        extra = control_lora_residuals(sd, ctl, x, timestep, text, pooled, time_ids)
        residuals = [r + e for r, e in zip(residuals, extra)]
    net = _Net(sd, loras)
    temb = timestep_embedding_sdxl(net, timestep, pooled, time_ids)
    shapes: list = []
    for n, kinds in enumerate(SDXL_DOWN):
        x = _stage(net, f"DownBlocks.Chain_{n + 1}", kinds, "sdxl", x, temb, text, ip, shapes)
        if t2i is not None and n in T2I_SDXL_BLOCKS:  # T2IFeatures sits in front of the ResidualAccumulator (xl/t2i_adapter.py:27-40)
            x = x + t2i["scale"] * t2i["features"][T2I_SDXL_BLOCKS.index(n)]
        residuals[n] = x + residuals[n]
    x = residual_block(net, "MiddleBlock.ResidualBlock_1", x, temb)
    x = cross_attention_2d(net, "MiddleBlock.SDXLCrossAttention", x, text, 20, True, ip)
    x = residual_block(net, "MiddleBlock.ResidualBlock_2", x, temb)
    if t2i is not None:  # the fourth feature map is appended to the MiddleBlock (xl/t2i_adapter.py:42-44)
        x = x + t2i["scale"] * t2i["features"][3]
    x = x + residuals[-1]  # fl.Residual(UseContext residuals[-1]), xl/unet.py:282 (slot 9 is ControlLora's middle output or 0.0)
    for n, kinds in enumerate(SDXL_UP):
        x = torch.cat([x, residuals[-n - 2]], dim=1)
        x = _stage(net, f"UpBlocks.Chain_{n + 1}", kinds, "sdxl", x, temb, text, ip, shapes)
    x = F.silu(net.group_norm("OutputBlock.GroupNorm", x, 1e-5))
    return net.conv("OutputBlock.Conv2d", x, padding=1)


@torch.no_grad()
def sd1_controlnet_residuals(cn: SD, x: Tensor, timestep: Tensor, text: Tensor, condition: Tensor, scale: float, scale_decay: float) -> list[Tensor]:
    """Controlnet (stable_diffusion_1/controlnet.py:66-150): its own TimestepEncoder / DownBlocks / MiddleBlock weights; the
    ConditionEncoder output is added right after conv_in (BEFORE the first 1x1 conv, unlike ControlLora); every block output goes
    through a 1x1 conv and is scaled by scale * scale_decay^(12 - n)."""
    net = _Net(cn, None)
    t = sinusoidal_embedding(timestep, 320)
    temb = net.linear("TimestepEncoder.RangeEncoder.Linear_2", F.silu(net.linear("TimestepEncoder.RangeEncoder.Linear_1", t)))
    x = x[:, :4]
    enc = {k.removeprefix("DownBlocks.Chain_1.Residual.ConditionEncoder."): v for k, v in cn.items() if "ConditionEncoder." in k}
    out: list[Tensor] = []
    shapes: list = []
    for n, kinds in enumerate(SD1_DOWN):
        x = _stage(net, f"DownBlocks.Chain_{n + 1}", kinds, "sd1", x, temb, text, None, shapes)
        if n == 0:
            x = x + condition_encoder(enc, condition)
        out.append(net.conv(f"DownBlocks.Chain_{n + 1}.Passthrough.Conv2d", x) * scale * scale_decay ** float(12 - n))
    x = residual_block(net, "MiddleBlock.ResidualBlock_1", x, temb)
    x = cross_attention_2d(net, "MiddleBlock.CLIPLCrossAttention", x, text, 8, False, None)
    x = residual_block(net, "MiddleBlock.ResidualBlock_2", x, temb)
    out.append(net.conv("MiddleBlock.Passthrough.Conv2d", x) * scale)
    return out


@torch.no_grad()
def sd1_unet(sd: SD, x: Tensor, timestep: Tensor, text: Tensor, controlnets: list | None = None) -> Tensor:
    """SD1UNet.forward (stable_diffusion_1/unet.py:165-249): 13 residual slots, middle block summed with residuals[-1].
    `controlnets`: dicts {weights, condition, scale, scale_decay}; each runs first and pre-populates the slots."""
    net = _Net(sd, None)
    t = sinusoidal_embedding(timestep, 320)
    temb = net.linear("TimestepEncoder.RangeEncoder.Linear_2", F.silu(net.linear("TimestepEncoder.RangeEncoder.Linear_1", t)))
    residuals: list[Any] = [0.0] * 13
    for c in controlnets or []:
        extra = sd1_controlnet_residuals(c["weights"], x, timestep, text, c["condition"], c["scale"], c["scale_decay"])
        residuals = [r + e for r, e in zip(residuals, extra)]
    shapes: list = []
    for n, kinds in enumerate(SD1_DOWN):
        x = _stage(net, f"DownBlocks.Chain_{n + 1}", kinds, "sd1", x, temb, text, None, shapes)
        residuals[n] = x + residuals[n]
    m = residual_block(net, "Sum.MiddleBlock.ResidualBlock_1", x, temb)
    m = cross_attention_2d(net, "Sum.MiddleBlock.CLIPLCrossAttention", m, text, 8, False, None)
    m = residual_block(net, "Sum.MiddleBlock.ResidualBlock_2", m, temb)
    x = residuals[-1] + m
    for n, kinds in enumerate(SD1_UP):
        x = torch.cat([x, residuals[-n - 2]], dim=1)
        x = _stage(net, f"UpBlocks.Chain_{n + 1}", kinds, "sd1", x, temb, text, None, shapes)
    x = F.silu(net.group_norm("Chain.GroupNorm", x, 1e-5))
    return net.conv("Chain.Conv2d", x, padding=1)


# ------------------------------------------------------------------------------------------------ sampling step
def ddim_tables(num_inference_steps: int, num_train: int = 1000) -> tuple[Tensor, Tensor]:
    """(timesteps, sqrt(alpha_bar)): quadratic schedule 8.5e-4..1.2e-2 (solvers/solver.py:151-180, 386-416),
    LEADING spacing with offset 1 (solvers/ddim.py:20-24, solver.py:226-228)."""
    betas = torch.linspace(8.5e-4 ** 0.5, 1.2e-2 ** 0.5, num_train) ** 2
    csf = torch.sqrt((1 - betas).cumprod(dim=0))
    ts = (torch.arange(0, num_inference_steps) * (num_train // num_inference_steps) + 1).flip(0)
    return ts, csf


def ddim_step(x: Tensor, noise: Tensor, step: int, num_inference_steps: int) -> Tensor:
    """DDIM.__call__ (solvers/ddim.py:56-95)."""
    ts, csf = ddim_tables(num_inference_steps)
    t = int(ts[step])
    prev_t = int(ts[step + 1]) if step < num_inference_steps - 1 else 0
    cur = csf[t]
    prev = csf[prev_t] if prev_t > 0 else csf[0]
    x0 = (x - torch.sqrt(1 - cur ** 2) * noise) / cur
    nf = torch.sqrt(1 - prev ** 2) if step != num_inference_steps - 1 else 0
    return prev * x0 + nf * noise


@torch.no_grad()
def sdxl_cfg_step(sd: SD, x: Tensor, step: int, num_inference_steps: int, text: Tensor, pooled: Tensor, time_ids: Tensor,
                  condition_scale: float = 5.0, **adapters: Any) -> Tensor:
    """LatentDiffusionModel.forward with classifier-free guidance (latent_diffusion/model.py:128-159): x (N,4,H,W) ->
    UNet on cat(x, x) with [negative ; conditional] embeddings -> u + s (c - u) -> DDIM."""
    ts, _ = ddim_tables(num_inference_steps)
    out = sdxl_unet(sd, torch.cat((x, x)), ts[step].unsqueeze(0), text, pooled, time_ids, **adapters)
    u, c = out.chunk(2)
    return ddim_step(x, u + condition_scale * (c - u), step, num_inference_steps)
