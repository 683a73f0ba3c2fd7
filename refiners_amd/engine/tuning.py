"""Measured tile choices for mi355x_gemm on MI355X (gfx950).

The library's own heuristic only knows the output size; which tile / LDS depth wins also depends on K, on whether the
launch's weights arrive cold, and on what the neighbouring launches leave in the caches.  `tools/autotune.py` therefore
measures every candidate IN PLACE (the whole recorded SDXL step replayed with one shape class switched at a time) on the
GPU and writes `tuning_gfx950.json`: {signature: [tile, stages]} (signature = native.gemm_signature).  Shapes the table
does not know fall back to the heuristic in gemm_kernel.cuh (pick_tile / pick_stages).  Set REFINERS_AMD_TUNING=0 to
ignore the table (A/B runs, and the tuner itself)."""
from __future__ import annotations

import json
import os
from pathlib import Path
from typing import Optional

TABLE_PATH = Path(__file__).resolve().parent / "tuning_gfx950.json"
_table: Optional[dict] = None
enabled = os.environ.get("REFINERS_AMD_TUNING", "1") != "0"
lora_g8 = os.environ.get("REFINERS_AMD_LORA_G8", "1") != "0"  # A/B: 0 = LoRA launches never inherit the 8-wave loop from their un-adapted shape class (round-4 behaviour)


def table_path() -> Path:
    """REFINERS_AMD_TUNING_TABLE names another table (a file name beside this module, or a path): A/B runs of a candidate entry (tools/ab_step.py)."""
    name = os.environ.get("REFINERS_AMD_TUNING_TABLE")
    return TABLE_PATH if not name else (Path(name) if os.sep in name else TABLE_PATH.parent / name)


def table() -> dict:
    global _table
    if _table is None:
        try:
            _table = {k: tuple(v) for k, v in json.loads(table_path().read_text())["choices"].items()}
        except (OSError, ValueError, KeyError):
            _table = {}
    return _table


def lookup(signature: str, stages: int = 0) -> tuple[int, int]:
    """(tile, stages) for a launch whose caller did not choose: the table's entry, else (0, stages) = library heuristic."""
    if enabled:
        got = table().get(signature)
        if got is None and signature.endswith("lora"):  # the LoRA producers leave the tiles' K loop untouched: same choice as the un-adapted launch
            got = table().get(signature[: -len("lora")])
            if got is not None and got[0] in (7, 8, 9, 10):
                # the 8-wave loop takes the LoRAs of a plain one-segment GEMM with one column group (whole tiles: no stream-K); a launch it cannot
                # take -- several groups, a transposed group, a convolution -- gets the 128 x 128 tile of the 4-wave kernel
                kind, _, _, seg, flags = signature.split(":")
                got = (9 if got[0] == 9 else 7, 0) if lora_g8 and kind == "gemm" and seg == "s1" and "T" not in flags else (1, 2)
            elif got is not None and (got[0] > 4 or got[1] != 2):  # the 4-wave LoRA kernels exist for tiles 1 .. 4 with two LDS stages
                got = (got[0] if got[0] <= 4 else 1, 2)
        if got is not None:
            return int(got[0]), int(got[1])
    return 0, stages


def summary() -> dict:
    return {"table": table_path().name if table() else None, "entries": len(table()), "enabled": enabled}
