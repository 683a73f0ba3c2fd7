"""Shared helpers of the parity tests: golden fixtures, synthetic weights, oracle adapters, error metrics."""
from __future__ import annotations

import json
import sys
from functools import lru_cache
from pathlib import Path
from typing import Any

import torch
from safetensors.torch import load_file

ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden"
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from refiners_amd import synth  # noqa: E402
from tests.golden_cases import CASES, build_specs  # noqa: E402


@lru_cache(maxsize=None)
def key_shapes(family: str) -> dict[str, tuple[int, ...]]:
    """The reference's bare-model state-dict keys and shapes (written by oracle/make_golden.py)."""
    return {k: tuple(v) for k, v in json.loads((GOLD / f"{family}_unet_keys.json").read_text()).items()}


@lru_cache(maxsize=2)
def weights(family: str, seed: int = 0) -> dict[str, torch.Tensor]:
    """Synthetic float32 CPU weights of the bare UNet (cached: SDXL is 2.57 G parameters, ~30 s to draw)."""
    return synth.synth_state_dict(key_shapes(family), seed)


def golden(case: str) -> dict[str, torch.Tensor]:
    return load_file(str(GOLD / f"{case}.safetensors"))


def manifest() -> dict[str, Any]:
    return json.loads((GOLD / "manifest.json").read_text())


def rel_err(got: torch.Tensor, ref: torch.Tensor) -> tuple[float, float]:
    """(||got - ref|| / ||ref||, max|got - ref| / max|ref|) in float64: the two figures SURVEY.md 8(d) asks for."""
    g, r = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((g - r).norm() / r.norm()), float((g - r).abs().max() / r.abs().max())


def oracle_adapters(specs: dict[str, Any]) -> dict[str, Any]:
    """synth specs -> keyword arguments of oracle.unet_oracle.sdxl_unet."""
    return {"loras": specs["loras"] or None, "ip": specs["ip"], "control": specs["control"] or None}


def load_mirror_weights(unet: Any, sd: dict[str, torch.Tensor], device: Any = None, dtype: Any = None) -> None:
    unet.load_state_dict({k: v.to(device=device, dtype=dtype) for k, v in sd.items()}, assign=True)


# ---- full-size oracle steps ---------------------------------------------------------------------------------------------------------------
# The GPU tests at the benchmarked geometry (128 x 128 latents, CFG pair) compare against ONE float32 step of the CPU oracle each.  That step
# costs a minute or more of host time per recipe -- on a GPU box, where every minute is GPU budget.  oracle/make_golden_full_size.py runs the
# three recipes below in the build container and commits the results (tests/golden/full_size_oracle.safetensors, 0.9 MB, recipe echoed in its
# metadata); full_size_oracle() returns the committed tensor when its recipe still matches and computes it on the spot otherwise.
FULL_SIZE = {
    "bare_step0": dict(weight_seed=0, input_seed=7, images=1, step=0, num_steps=50, condition_scale=5.0, adapters="none"),
    "lora_ip_step7": dict(weight_seed=0, input_seed=8, images=1, step=7, num_steps=50, condition_scale=5.0, adapters="2 loras (1.0, 0.8; seed 5) + ip 0.6 (seed 5)"),
    "control_single_step12": dict(weight_seed=0, input_seed=9, images=4, pick=[0, 4], step=12, num_steps=30, condition_scale=7.5, adapters="control canny 0.9 (seed 6, rows 0 and 4 of 8)"),
}


def full_size_inputs(name: str) -> tuple[dict[str, Any], dict[str, torch.Tensor]]:
    """(adapter specs, CPU inputs) of a FULL_SIZE recipe -- what both the engine under test and the oracle are fed."""
    r = FULL_SIZE[name]
    shapes = key_shapes("sdxl")
    specs: dict[str, Any] = {"loras": [], "ip": None, "control": []}
    if name == "lora_ip_step7":
        specs = {"loras": [synth.lora_spec(shapes, "l1", 1.0, seed=5), synth.lora_spec(shapes, "l2", 0.8, seed=5)], "ip": synth.ip_spec(shapes, 0.6, batch=2, seed=5), "control": []}
    inp = synth.sdxl_inputs(r["images"], (128, 128), seed=r["input_seed"])
    if name == "control_single_step12":
        n = r["images"]
        ctl = synth.control_spec("canny", 0.9, 2 * n, (128, 128), seed=6)
        specs["control_batch"] = ctl  # the whole batch's adapter (the engine test runs all four images)
        pick = torch.tensor(r["pick"])
        one = synth.control_spec("canny", 0.9, 2, (128, 128), seed=6)
        one["condition"] = ctl["condition"][pick]
        specs["control"] = [one]
    return specs, inp


def compute_full_size_oracle(name: str) -> torch.Tensor:
    from oracle import unet_oracle as O

    r = FULL_SIZE[name]
    specs, inp = full_size_inputs(name)
    sd = weights("sdxl", r["weight_seed"])
    if name == "control_single_step12":
        pick = torch.tensor(r["pick"])
        return O.sdxl_cfg_step(sd, inp["x"][:1], r["step"], r["num_steps"], inp["text"][pick], inp["pooled"][pick], inp["time_ids"][pick], condition_scale=r["condition_scale"],
                               control=specs["control"])
    return O.sdxl_cfg_step(sd, inp["x"], r["step"], r["num_steps"], inp["text"], inp["pooled"], inp["time_ids"], condition_scale=r["condition_scale"],
                           loras=specs["loras"] or None, ip=specs["ip"])


def oracle_sources_digest() -> str:
    """sha256 over the sources a FULL_SIZE tensor depends on besides its recipe: the CPU oracle (oracle/*.py without the fixture writers) and the synthetic
    weights / inputs (refiners_amd/synth.py).  Stored in the fixture's metadata; an edit to any of them makes full_size_oracle() recompute instead of
    serving a stale reference (round-4 advisor)."""
    import hashlib

    root = GOLD.parent.parent
    files = sorted(p for p in (root / "oracle").glob("*.py") if not p.name.startswith("make_golden")) + [root / "refiners_amd" / "synth.py"]
    h = hashlib.sha256()
    for p in files:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def synth_digest() -> str:
    """sha256 of refiners_amd/synth.py: what a reference-written full-size tensor depends on besides its recipe (weights, adapters and inputs are drawn there)."""
    import hashlib

    return hashlib.sha256((GOLD.parent.parent / "refiners_amd" / "synth.py").read_bytes()).hexdigest()


def full_size_reference(name: str) -> "torch.Tensor | None":
    """x_next of a FULL_SIZE recipe as the REAL reference computed it (tests/golden/full_size_reference.safetensors, written in the build container by
    oracle/make_golden_full_size_reference.py from refiners' own classes), or None when the committed file does not hold the recipe as it stands now (an edited
    recipe or an edited synth.py): callers then fall back to the oracle, which tests/test_oracle_golden.py pins to the reference."""
    from safetensors import safe_open

    path = GOLD / "full_size_reference.safetensors"
    if not path.exists():
        return None
    with safe_open(str(path), framework="pt") as f:
        meta = f.metadata() or {}
        if name in f.keys() and json.loads(meta.get(name, "null")) == FULL_SIZE[name] and meta.get("synth") == synth_digest():
            return f.get_tensor(name)
    return None


def full_size_golden(name: str) -> tuple[torch.Tensor, str]:
    """(x_next, who computed it): the REAL reference's full-size step where the committed file holds the recipe ("reference"), else the CPU oracle's ("oracle")."""
    ref = full_size_reference(name)
    return (ref, "reference") if ref is not None else (full_size_oracle(name), "oracle")


def full_size_oracle(name: str) -> torch.Tensor:
    from safetensors import safe_open

    path = GOLD / "full_size_oracle.safetensors"
    if path.exists():
        with safe_open(str(path), framework="pt") as f:
            meta = f.metadata() or {}
            if name in f.keys() and json.loads(meta.get(name, "null")) == FULL_SIZE[name] and meta.get("sources") == oracle_sources_digest():
                return f.get_tensor(name)
    return compute_full_size_oracle(name)
