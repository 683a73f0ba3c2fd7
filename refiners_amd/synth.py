"""Synthetic (seeded) weights, adapters and inputs for the SDXL hot path.

There are no pretrained checkpoints in this environment (no network), so every parity test, golden vector and bench
run uses weights drawn here.  The draw is per state-dict KEY (generator seeded with `seed ^ crc32(key)`), so it does not
depend on construction order and can be reproduced by any process that knows the key list: the reference model
(oracle/make_golden.py), the CPU oracle, the host mirror and the HIP path all load the very same tensors.

Adapters are described as plain data ("specs") that both the oracle and `apply_adapters` understand:
  lora    : {"name", "scale", "pairs": {<Linear/Conv2d key prefix in the BARE UNet>: (down, up)}}
  ip      : {"scale", "tokens": (B, T, 2048), "kv": {<"...Residual_2.Attention" prefix>: (Wk', Wv')}}
  control : {"name", "scale", "condition": (B, 3, 8H, 8W), "encoder": {...}, "zero": [(w, b)] * 10, "loras": [...]}
following SURVEY.md section 8(d) (LoRA `up` is NOT zero here, norm affine parameters are not 1/0, zero-convs are not 0,
so that none of those paths is a numerical no-op).
"""
from __future__ import annotations

import zlib
from typing import Any, Mapping, Sequence

import torch
from torch import Tensor

CONTROL_ENCODER_SHAPES = {  # ConditionEncoder(3 -> 16 -> 16/32 -> 32/96 -> 96/256 -> 320), xl/control_lora.py:14-87
    "Chain_1.Conv2d": (16, 3), "Chain_2.Conv2d_1": (16, 16), "Chain_2.Conv2d_2": (32, 16), "Chain_3.Conv2d_1": (32, 32),
    "Chain_3.Conv2d_2": (96, 32), "Chain_4.Conv2d_1": (96, 96), "Chain_4.Conv2d_2": (256, 96), "Conv2d": (320, 256),
}
CONTROL_SLOT_CHANNELS = (320, 320, 320, 320, 640, 640, 640, 1280, 1280, 1280)


def _gen(key: str, seed: int) -> torch.Generator:
    return torch.Generator(device="cpu").manual_seed((seed * 1_000_003) ^ zlib.crc32(key.encode()))


def synth_tensor(key: str, shape: Sequence[int], seed: int = 0, gain: float = 1.0) -> Tensor:
    """float32 CPU tensor for state-dict entry `key`: norm gains 1 + 0.1 n, biases 0.1 n, matrices n / sqrt(fan_in)."""
    g = _gen(key, seed)
    shape = tuple(shape)
    leaf, kind = key.split(".")[-2:] if "." in key else ("", key)
    n = torch.randn(shape, generator=g, dtype=torch.float32)
    if "Norm" in leaf:
        return 1 + 0.1 * n if kind == "weight" else 0.1 * n
    if kind == "bias" or len(shape) < 2:
        return 0.1 * n
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    return n * (gain / fan_in ** 0.5)


def synth_state_dict(shapes: Mapping[str, Sequence[int]], seed: int = 0) -> dict[str, Tensor]:
    return {k: synth_tensor(k, s, seed) for k, s in shapes.items()}


def model_shapes(model: torch.nn.Module) -> dict[str, tuple[int, ...]]:
    return {k: tuple(v.shape) for k, v in model.state_dict().items()}


# ------------------------------------------------------------------------------------------------ adapter specs
def linear_targets(shapes: Mapping[str, Sequence[int]], ancestor: str = "SDXLCrossAttention") -> list[str]:
    """Key prefixes of every fl.Linear below an `ancestor` node of the bare UNet (722 for SDXL), in walk order."""
    out = []
    for k, s in shapes.items():
        parts = k.split(".")
        if parts[-1] == "weight" and parts[-2].split("_")[0] == "Linear" and any(p.split("_")[0] == ancestor for p in parts):
            out.append(k[: -len(".weight")])
    return out


def lora_spec(shapes: Mapping[str, Sequence[int]], name: str, scale: float, rank: int = 16, seed: int = 0,
              targets: Sequence[str] | None = None) -> dict[str, Any]:
    pairs = {}
    for p in (linear_targets(shapes) if targets is None else targets):
        w = shapes[f"{p}.weight"]
        if len(w) == 2:
            down = synth_tensor(f"lora.{name}.{p}.down.weight", (rank, w[1]), seed)
            up = synth_tensor(f"lora.{name}.{p}.up.weight", (w[0], rank), seed, gain=0.25)
        else:  # Conv2dLora: down k x k (target's kernel), up 1 x 1
            down = synth_tensor(f"lora.{name}.{p}.down.weight", (rank, w[1], w[2], w[3]), seed)
            up = synth_tensor(f"lora.{name}.{p}.up.weight", (w[0], rank, 1, 1), seed, gain=0.25)
        pairs[p] = (down, up)
    return {"name": name, "scale": scale, "pairs": pairs}


def ip_targets(shapes: Mapping[str, Sequence[int]]) -> list[str]:
    """Prefixes of the text cross-attentions ("...Residual_2.Attention"), in walk order (70 for SDXL)."""
    return [k[: -len(".Linear.weight")] for k in shapes if k.endswith("Residual_2.Attention.Linear.weight")]


def ip_spec(shapes: Mapping[str, Sequence[int]], scale: float, batch: int, num_tokens: int = 4, seed: int = 0) -> dict[str, Any]:
    kv = {}
    for p in ip_targets(shapes):
        inner, width = shapes[f"{p}.Distribute.Linear_2.weight"]
        kv[p] = (synth_tensor(f"ip.{p}.k.weight", (inner, width), seed), synth_tensor(f"ip.{p}.v.weight", (inner, width), seed))
    tokens = torch.randn((batch, num_tokens, 2048), generator=_gen("ip.tokens", seed), dtype=torch.float32)
    return {"scale": scale, "tokens": tokens, "kv": kv}


def control_spec(name: str, scale: float, batch: int, latent_hw: tuple[int, int], seed: int = 0,
                 loras: list[dict[str, Any]] | None = None) -> dict[str, Any]:
    enc = {}
    for k, (co, ci) in CONTROL_ENCODER_SHAPES.items():
        enc[f"{k}.weight"] = synth_tensor(f"control.{name}.enc.{k}.weight", (co, ci, 3, 3), seed)
        enc[f"{k}.bias"] = synth_tensor(f"control.{name}.enc.{k}.bias", (co,), seed)
    zero = [
        (synth_tensor(f"control.{name}.zero{i}.weight", (c, c, 1, 1), seed, gain=0.5), synth_tensor(f"control.{name}.zero{i}.bias", (c,), seed))
        for i, c in enumerate(CONTROL_SLOT_CHANNELS)
    ]
    h, w = latent_hw
    cond = torch.rand((batch, 3, 8 * h, 8 * w), generator=_gen(f"control.{name}.condition", seed), dtype=torch.float32)
    return {"name": name, "scale": scale, "condition": cond, "encoder": enc, "zero": zero, "loras": loras or []}


def sdxl_inputs(images: int, latent_hw: tuple[int, int], seed: int = 1, cfg: bool = True) -> dict[str, Tensor]:
    """Latents for `images` prompts plus [negative ; conditional] embeddings for the CFG batch."""
    b = images * (2 if cfg else 1)
    h, w = latent_hw
    return {
        "x": torch.randn((images, 4, h, w), generator=_gen("in.x", seed)),
        "text": torch.randn((b, 77, 2048), generator=_gen("in.text", seed)),
        "pooled": torch.randn((b, 1280), generator=_gen("in.pooled", seed)),
        "time_ids": torch.tensor([[1024, 1024, 0, 0, 1024, 1024]]).repeat(b, 1),
    }


# ------------------------------------------------------------------------------------------------ specs -> tree
def apply_adapters(unet: Any, api: Any, *, loras: Sequence[dict[str, Any]] = (), ip: dict[str, Any] | None = None,
                   control: Sequence[dict[str, Any]] = (), device: Any = None, dtype: Any = None) -> dict[str, Any]:
    """Inject the adapters described by the specs into a (reference or mirror) UNet tree through its own public API.

    `api` provides `fl`, `LinearLora`, `Conv2dLora`, `LoraAdapter`, `SDXLIPAdapter`, `ControlLoraAdapter` and works
    unchanged for `refiners` (oracle/make_golden.py) and for `refiners_amd` (api = refiners_amd.namespace()).
    Order: ControlLoras (they structural_copy the bare encoder), then their own LoRAs, then the IP-Adapter, then the
    UNet LoRAs (resolved on the bare tree first, so that key prefixes refer to the bare model).
    """
    fl = api.fl
    conv = lambda t: t.to(device=device, dtype=dtype)

    def resolve(root: Any, prefix: str) -> tuple[Any, Any]:
        path = prefix.split(".")
        return root.layer(path, fl.WeightedModule), root.layer(path[:-1], fl.Chain)

    def make_lora(spec: dict[str, Any], down: Tensor, up: Tensor) -> Any:
        cls = api.LinearLora if down.ndim == 2 else api.Conv2dLora
        lora = cls.from_weights(spec["name"], down=conv(down), up=conv(up))
        lora.scale = spec["scale"]
        return lora

    sites: dict[str, tuple[Any, Any]] = {}
    for spec in loras:
        for prefix in spec["pairs"]:
            if prefix not in sites:
                sites[prefix] = resolve(unet, prefix)
    handles: dict[str, Any] = {"control": [], "ip": None, "loras": []}
    for ctl in control:
        ad = api.ControlLoraAdapter(name=ctl["name"], target=unet, scale=ctl["scale"])
        cl = ad.control_lora
        cl.ensure_find(api.ConditionEncoder).load_state_dict({k: conv(v) for k, v in ctl["encoder"].items()})
        for zc, (w, b) in zip(cl.layers(api.ZeroConvolution), ctl["zero"]):
            zc.load_state_dict({"Conv2d.weight": conv(w), "Conv2d.bias": conv(b)})
        ctl_sites = {}
        for spec in ctl["loras"]:
            for prefix in spec["pairs"]:
                ctl_sites.setdefault(prefix, resolve(cl, prefix.removeprefix("")))
        for prefix, (leaf, parent) in ctl_sites.items():
            parts = [make_lora(s, *s["pairs"][prefix]) for s in ctl["loras"] if prefix in s["pairs"]]
            for lr in parts:
                assert lr.is_compatible(leaf)
            api.LoraAdapter(leaf, *parts).inject(parent)
        ad.inject()
        ad.set_condition(conv(ctl["condition"]))
        handles["control"].append(ad)
    if ip is not None:
        ad = api.SDXLIPAdapter(target=unet, clip_image_encoder=_NoImageEncoder(), scale=ip["scale"])
        assert len(ad.sub_adapters) == len(ip["kv"]), (len(ad.sub_adapters), len(ip["kv"]))
        for sub, (wk, wv) in zip(ad.sub_adapters, ip["kv"].values()):
            sub.image_key_projection.weight = torch.nn.Parameter(conv(wk))
            sub.image_value_projection.weight = torch.nn.Parameter(conv(wv))
        ad.inject()
        ad.set_clip_image_embedding(conv(ip["tokens"]))
        handles["ip"] = ad
    for prefix, (leaf, parent) in sites.items():
        parts = [make_lora(s, *s["pairs"][prefix]) for s in loras if prefix in s["pairs"]]
        for lr in parts:
            assert lr.is_compatible(leaf)
        handles["loras"].append(api.LoraAdapter(leaf, *parts).inject(parent))
    return handles


class _NoImageEncoder:
    """Stand-in for CLIPImageEncoderH (out of scope): only its `output_dim` is consulted by SDXLIPAdapter."""

    output_dim = 1024
    embedding_dim = 1280
