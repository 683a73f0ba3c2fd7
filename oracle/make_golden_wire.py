"""Golden data for the weight wire format: where the REAL reference's SDLoraManager attaches a
CivitAI-style SDXL LoRA file (kohya key names, down/up pairs, alphas) in the UNet.  Runs on the meta device (no numbers,
only structure): tests/golden/lora_wire_sdxl.json = the file's keys and shapes + the reference's (key -> adapter path) map.
Run in the build container only:  python oracle/make_golden_wire.py"""
from __future__ import annotations

import hashlib
import re
import json
import sys
from pathlib import Path
from types import SimpleNamespace

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "oracle" / "shim"), "/root/reference/src", str(ROOT)]

import torch  # noqa: E402

import refiners.fluxion.layers as fl  # noqa: E402
from refiners.fluxion.adapters.lora import LoraAdapter  # noqa: E402
from refiners.foundationals.latent_diffusion.lora import SDLoraManager  # noqa: E402
from refiners.foundationals.latent_diffusion.stable_diffusion_xl.unet import SDXLUNet  # noqa: E402

from tests.golden_cases import kohya_sdxl_lora_keys  # noqa: E402

GOLD = ROOT / "tests" / "golden"


def main() -> None:
    keys = kohya_sdxl_lora_keys(rank=8)
    tensors = {k: torch.empty(shape, device="meta") for k, shape in keys}
    unet = SDXLUNet(4, device="meta")
    manager = SDLoraManager(SimpleNamespace(unet=unet, clip_text_encoder=fl.Chain(fl.Identity()), device=torch.device("meta"), dtype=torch.float32))  # type: ignore[arg-type]
    manager.add_loras("style", tensors=tensors, scale=0.75)
    attached = [[a.get_path(), [list(lr.down.weight.shape) for lr in a.loras.values()]] for a in unet.layers(LoraAdapter)]
    # the same through the two-step internal route, to record which file key landed where
    from refiners.fluxion.adapters.lora import Lora  # noqa: E402

    unet2 = SDXLUNet(4, device="meta")
    manager2 = SDLoraManager(SimpleNamespace(unet=unet2, clip_text_encoder=fl.Chain(fl.Identity()), device=torch.device("meta"), dtype=torch.float32))  # type: ignore[arg-type]
    loras = Lora.from_dict("style", state_dict=tensors)
    loras = {k: loras[k] for k in sorted(loras.keys(), key=SDLoraManager.sort_keys)}
    key_map: list[tuple[str, str]] = []
    manager2.add_loras_to_unet(loras, debug_map=key_map)
    manager2.set_scale("style", 0.75)
    assert repr(unet2) == repr(unet)
    out = {"keys": [[k, list(s)] for k, s in keys], "attached": attached, "key_map": [list(kv) for kv in key_map], "repr_sha256": hashlib.sha256(re.sub(r"Lambda\(.*\)", "Lambda", repr(unet)).encode()).hexdigest(),
           "scales": manager.scales, "names": manager.names}
    (GOLD / "lora_wire_sdxl.json").write_text(json.dumps(out))
    print(len(keys), "tensors ->", len(attached), "adapters;", attached[0], attached[-1])


if __name__ == "__main__":
    main()
