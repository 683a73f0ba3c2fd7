"""Recipes of the golden cases: everything needed to rebuild weights, adapters and inputs from seeds alone.

Shared by oracle/make_golden.py (which runs the real reference on them) and by the tests that check the oracle, the
host mirror and the HIP path against the stored reference outputs.  Latents are 32x32 (the size the reference's own
UNet parity test uses, tests/foundationals/latent_diffusion/test_sdxl_unet.py:17-54) so that a case costs seconds.
"""
from __future__ import annotations

from typing import Any, Mapping, Sequence

from refiners_amd import synth

CASES: dict[str, dict[str, Any]] = {
    # BASELINE.json config 2 at reduced latent size: bare SDXL, one CFG pair, DDIM step 10 of 50
    "sdxl_bare": dict(family="sdxl", weight_seed=0, input_seed=1, images=1, latent_hw=(32, 32), num_steps=50, step=10, condition_scale=5.0,
                      adapters=[]),
    # config 3: IP-Adapter + two rank-16 LoRAs (scales 1.0 / 0.8) on all 722 Linears under SDXLCrossAttention
    "sdxl_lora_ip": dict(family="sdxl", weight_seed=0, input_seed=2, images=1, latent_hw=(32, 32), num_steps=50, step=25, condition_scale=5.0,
                         adapters=["ip", "lora:l1:1.0", "lora:l2:0.8"]),
    # config 4's adapter: ControlLora (with its own rank-8 LoRA on the copied encoder) at scale 0.9, non-square latent
    "sdxl_control": dict(family="sdxl", weight_seed=0, input_seed=3, images=1, latent_hw=(32, 24), num_steps=30, step=29, condition_scale=7.5,
                         adapters=["control:canny:0.9"]),
    # two STACKED ControlLoras at 0.55 each, as the reference's own end-to-end test does (tests/e2e/test_diffusion.py:299-310,
    # xl/control_lora.py:251-411): distinct names / contexts / scales / control pictures, the first with its own LoRAs, the second without
    "sdxl_control2": dict(family="sdxl", weight_seed=0, input_seed=6, images=1, latent_hw=(24, 16), num_steps=30, step=7, condition_scale=6.0,
                          adapters=["control:canny:0.55", "control:depth:0.45:nolora"]),
    # Conv2d LoRAs on ResidualBlock / Downsample / Upsample convolutions (reference tests/e2e/test_diffusion.py:1655-1663)
    "sdxl_conv_lora": dict(family="sdxl", weight_seed=0, input_seed=4, images=1, latent_hw=(16, 16), num_steps=50, step=0, condition_scale=5.0,
                           adapters=["convlora:c1:0.7"]),
    # config 1 at reduced latent size: SD1.5, single forward, no CFG
    "sd1_bare": dict(family="sd1", weight_seed=0, input_seed=5, latent_hw=(32, 32), timestep=500),
}


def conv_lora_targets(shapes: Mapping[str, Sequence[int]]) -> list[str]:
    out = []
    for k, s in shapes.items():
        if k.endswith(".weight") and len(s) == 4 and any(t in k for t in ("ResidualBlock", "Downsample", "Upsample")):
            if "MiddleBlock" in k or "Chain_4" in k or "Chain_3.Upsample" in k or "UpBlocks.Chain_9" in k:
                out.append(k[: -len(".weight")])
    return out


def control_lora_targets(shapes: Mapping[str, Sequence[int]]) -> list[str]:
    """A handful of encoder-half Linears and convs (real control-lora files adapt all of them at rank 128)."""
    picks = []
    for k in shapes:
        if not k.endswith(".weight") or not (k.startswith("DownBlocks") or k.startswith("MiddleBlock")):
            continue
        if "Chain_5.SDXLCrossAttention.Chain_2.CrossAttentionBlock_1" in k and "Linear" in k.split(".")[-2]:
            picks.append(k[: -len(".weight")])
        elif "Chain_2.ResidualBlock.Chain.Conv2d" in k or "MiddleBlock.ResidualBlock_1.Chain.RangeAdapter2d.Conv2d" in k:
            picks.append(k[: -len(".weight")])
    return picks


def build_specs(cfg: Mapping[str, Any], shapes: Mapping[str, Sequence[int]]) -> dict[str, Any]:
    """Adapter specs (refiners_amd.synth formats) of a case: kwargs for synth.apply_adapters and for the oracle."""
    seed = cfg["weight_seed"] + 100
    batch = cfg["images"] * 2
    loras, ip, control = [], None, []
    for item in cfg["adapters"]:
        kind, *rest = item.split(":")
        if kind == "lora":
            loras.append(synth.lora_spec(shapes, rest[0], float(rest[1]), rank=16, seed=seed))
        elif kind == "convlora":
            loras.append(synth.lora_spec(shapes, rest[0], float(rest[1]), rank=8, seed=seed, targets=conv_lora_targets(shapes)))
        elif kind == "ip":
            ip = synth.ip_spec(shapes, scale=0.6, batch=batch, seed=seed)
        elif kind == "control":
            own = [] if "nolora" in rest[2:] else [synth.lora_spec(shapes, f"ctl_{rest[0]}", 1.0, rank=8, seed=seed + 1, targets=control_lora_targets(shapes))]
            control.append(synth.control_spec(rest[0], float(rest[1]), batch, cfg["latent_hw"], seed=seed, loras=own))
    return {"loras": loras, "ip": ip, "control": control}


# ------------------------------------------------------------------------------------------------ config 5: SAM ViT-H
SAM_CASE = dict(weight_seed=0, input_seed=9)


def sam_sample(neck_out, early):
    """Strided samples + statistics of the image encoder's two outputs (what tests/golden/sam_vit_h.safetensors stores)."""
    import torch

    stats = torch.tensor([neck_out.mean(), neck_out.abs().mean(), neck_out.std(), early.mean(), early.abs().mean(), early.std()], dtype=torch.float64)
    return {"neck": neck_out[:, :, ::4, ::4].float().clone(), "early": early[:, ::8, ::8, ::8].float().clone(), "stats": stats.float()}


# ------------------------------------------------------------------------------------------------ next-1: VAE decode
VAE_CASE = dict(weight_seed=0, input_seed=13, latent_hw=(16, 24), latent_std=0.13)


# ------------------------------------------------------------------------------------------------ next-2: prompt encoder
CLIP_CASE = dict(weight_seed=0, prompts=("a photograph of an astronaut riding a horse on mars, highly detailed, 4k, dramatic lighting", ""))
CLIP_IMAGE_CASE = dict(weight_seed=0, input_seed=21)


# ------------------------------------------------------------------------------------------------ next-3: LoRA wire format
def kohya_sdxl_lora_keys(rank: int = 8):
    """(key, shape) list of a CivitAI-style SDXL LoRA file restricted to the UNet's attention stacks, in FILE order
    (diffusers module names joined by '_', `lora_down` / `lora_up` / `alpha` per module): 722 modules."""
    out = []

    def mod(name, fin, fout):
        out.append((f"lora_unet_{name}.alpha", ()))
        out.append((f"lora_unet_{name}.lora_down.weight", (rank, fin)))
        out.append((f"lora_unet_{name}.lora_up.weight", (fout, rank)))

    stacks = [("down_blocks_1", 640, 2, 2), ("down_blocks_2", 1280, 2, 10), ("mid_block", 1280, 1, 10), ("up_blocks_0", 1280, 3, 10), ("up_blocks_1", 640, 3, 2)]
    for block, C, n_att, depth in stacks:
        for a in range(n_att):
            base = f"{block}_attentions_{a}"
            mod(f"{base}_proj_in", C, C)
            mod(f"{base}_proj_out", C, C)
            for t in range(depth):
                tb = f"{base}_transformer_blocks_{t}"
                for proj in ("to_k", "to_out_0", "to_q", "to_v"):  # alphabetical, as safetensors files store them
                    mod(f"{tb}_attn1_{proj}", C, C)
                for proj, fin in (("to_k", 2048), ("to_out_0", C), ("to_q", C), ("to_v", 2048)):
                    mod(f"{tb}_attn2_{proj}", fin, C)
                mod(f"{tb}_ff_net_0_proj", C, 8 * C)
                mod(f"{tb}_ff_net_2", 4 * C, C)
    return out


# ------------------------------------------------------------------------------------------------ next-4: T2I-Adapter
T2I_CASE = dict(weight_seed=0, input_seed=31, latent_hw=(32, 32), num_steps=50, step=20, scale=0.8)
CONTROLNET_CASE = dict(weight_seed=0, input_seed=41, latent_hw=(16, 16), timestep=601, scale=0.8, scale_decay=0.9)


# ------------------------------------------------------------------------------------------------ next-3: adapter checkpoint formats
def tagged(keys, device="cpu"):
    """{key: tensor of the given shape whose every element equals the key's index + 1} as stride-0 views (no memory)."""
    import torch

    return {k: torch.tensor(float(i + 1), device=device).expand(tuple(shape)) if shape else torch.tensor(float(i + 1), device=device) for i, (k, shape) in enumerate(keys)}


def ip_adapter_file():
    """(key, shape) list of an SDXL IP-Adapter checkpoint in refiners' converted format, in FILE order: `image_proj.*` then, per
    text cross-attention NNN (70 of them, UNet walk order), `ip_adapter.NNN.to_k_ip.weight` and `ip_adapter.NNN.to_v_ip.weight`."""
    out = [("image_proj.Linear.weight", (4 * 2048, 1024)), ("image_proj.Linear.bias", (4 * 2048,)), ("image_proj.LayerNorm.weight", (2048,)), ("image_proj.LayerNorm.bias", (2048,))]
    widths = [640] * 4 + [1280] * 20 + [1280] * 10 + [1280] * 30 + [640] * 6  # down 2x2 + 2x10, middle 10, up 3x10 + 3x2 transformer blocks
    for i, c in enumerate(widths):
        # on the meta device nothing checks these shapes, so the second dimension doubles as a TAG (2048 + 2 i for the key tensor,
        # + 1 for the value tensor): reading the shapes back from the loaded adapter tells which file tensor landed where
        out.append((f"ip_adapter.{i:03d}.to_k_ip.weight", (c, 2048 + 2 * i)))
        out.append((f"ip_adapter.{i:03d}.to_v_ip.weight", (c, 2048 + 2 * i + 1)))
    return out


def control_lora_file(rank: int = 8):
    """(key, shape) list of a ControlLora checkpoint in refiners' converted format: LoRA pairs for a few encoder-half layers
    (real files adapt all of them at rank 128), the ten zero convolutions, the condition encoder."""
    from refiners_amd import synth

    out = []
    sites = [("DownBlocks.Chain_5.SDXLCrossAttention.Chain_2.CrossAttentionBlock_1.Residual_1.SelfAttention.Distribute.Linear_1", (640, 640)),
             ("DownBlocks.Chain_5.SDXLCrossAttention.Chain_2.CrossAttentionBlock_1.Residual_3.Linear_1", (5120, 640)),
             ("DownBlocks.Chain_2.ResidualBlock.Chain.Conv2d", (320, 320, 3, 3)),
             ("MiddleBlock.ResidualBlock_1.Chain.RangeAdapter2d.Conv2d", (1280, 1280, 3, 3))]
    for path, w in sites:
        if len(w) == 2:
            out += [(f"ControlLora.{path}.down", (rank, w[1])), (f"ControlLora.{path}.up", (w[0], rank))]
        else:
            out += [(f"ControlLora.{path}.down", (rank, w[1], w[2], w[3])), (f"ControlLora.{path}.up", (w[0], rank, 1, 1))]
    for i, c in enumerate(synth.CONTROL_SLOT_CHANNELS):
        out += [(f"ZeroConvolution_{i + 1:02d}.Conv2d.weight", (c, c, 1, 1)), (f"ZeroConvolution_{i + 1:02d}.Conv2d.bias", (c,))]
    for k, (co, ci) in synth.CONTROL_ENCODER_SHAPES.items():
        out += [(f"ConditionEncoder.{k}.weight", (co, ci, 3, 3)), (f"ConditionEncoder.{k}.bias", (co,))]
    return out


# ------------------------------------------------------------------------------------------------ next-4: Self-Attention Guidance
SAG_CASE = dict(weight_seed=0, input_seed=51, latent_hw=(32, 32), num_steps=30, step=11, condition_scale=5.0, sag_scale=0.75)
