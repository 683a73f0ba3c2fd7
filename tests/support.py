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
