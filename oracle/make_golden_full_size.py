"""Test infrastructure (never imported by refiners_amd/ or bench.py): run the CPU oracle ONCE at the benchmarked geometry for the recipes of
tests/support.py::FULL_SIZE and commit the results, so that the GPU tests at full size do not spend GPU-box minutes on a host computation.

    python oracle/make_golden_full_size.py [name ...]        # ~2-5 min per recipe on 8 cores, ~25 GB RAM

Writes tests/golden/full_size_oracle.safetensors: one float32 x_next per recipe (1 x 4 x 128 x 128), the recipe as JSON in the file's metadata
(tests/support.py::full_size_oracle ignores an entry whose recipe no longer matches, or a file whose `sources` digest -- sha256 over oracle/*.py and
refiners_amd/synth.py -- differs from the tree's, and computes the step on the spot instead).  The oracle
itself is pinned against the reference's own outputs by tests/test_oracle_golden.py (tests/golden/sdxl_*.safetensors, oracle/make_golden.py)."""
from __future__ import annotations

import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402
from safetensors import safe_open  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

from tests import support as S  # noqa: E402


def main() -> None:
    path = S.GOLD / "full_size_oracle.safetensors"
    tensors, meta = {}, {}
    if path.exists():
        with safe_open(str(path), framework="pt") as f:
            meta = dict(f.metadata() or {})
            tensors = {k: f.get_tensor(k) for k in f.keys()}
    for name in sys.argv[1:] or list(S.FULL_SIZE):
        t0 = time.time()
        with torch.no_grad():
            tensors[name] = S.compute_full_size_oracle(name).float().contiguous()
        meta[name] = json.dumps(S.FULL_SIZE[name])
        print(name, tuple(tensors[name].shape), f"abs mean {float(tensors[name].abs().mean()):.4f}", f"{time.time() - t0:.0f} s", flush=True)
        save_file(tensors, str(path), metadata={**meta, "torch": torch.__version__, "sources": S.oracle_sources_digest()})


if __name__ == "__main__":
    main()
