"""Golden data for the adapter checkpoint formats: where the REAL reference puts the tensors of
  * an IP-Adapter file   -- `image_proj.<key>` + `ip_adapter.NNN.<anything>` pairs (image_prompt.py:395-410): the two tensors of
    index NNN become the key / value projection of the NNN-th text cross-attention, IN FILE ORDER;
  * a ControlLora file   -- `ControlLora.<path>.{down,up}`, `ZeroConvolution_NN.*`, `ConditionEncoder.*` (xl/control_lora.py:345-411).
Everything runs on the meta device; the IP-Adapter's key / value tensors carry a distinct TAG in their second dimension (see
tests/golden_cases.ip_adapter_file), so "which file tensor ended up in which parameter" can be read back from the loaded modules.  Run in the build container only:  python oracle/make_golden_adapter_wire.py"""
from __future__ import annotations

import hashlib
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "oracle" / "shim"), "/root/reference/src", str(ROOT)]

import torch  # noqa: E402

from refiners.fluxion.adapters.lora import LoraAdapter  # noqa: E402
from refiners.foundationals.latent_diffusion.stable_diffusion_xl.control_lora import ConditionEncoder, ControlLoraAdapter, ZeroConvolution  # noqa: E402
from refiners.foundationals.latent_diffusion.stable_diffusion_xl.image_prompt import SDXLIPAdapter  # noqa: E402
from refiners.foundationals.latent_diffusion.stable_diffusion_xl.unet import SDXLUNet  # noqa: E402

from tests.golden_cases import control_lora_file, ip_adapter_file, tagged  # noqa: E402

GOLD = ROOT / "tests" / "golden"


class _Enc:  # SDXLIPAdapter only consults these two attributes of the image encoder
    output_dim, embedding_dim = 1024, 1280


def main() -> None:
    out = {}
    # ---- IP-Adapter ---------------------------------------------------------------------------------------------------------
    unet = SDXLUNet(4, device="meta")
    keys = ip_adapter_file()
    weights = tagged(keys, device="meta")
    ad = SDXLIPAdapter(target=unet, clip_image_encoder=_Enc(), weights=weights)  # type: ignore[arg-type]
    landed = [[i, list(sub.image_key_projection.weight.shape), list(sub.image_value_projection.weight.shape)] for i, sub in enumerate(ad.sub_adapters)]
    out["ip_adapter"] = {"keys": [[k, list(s)] for k, s in keys], "image_proj_keys": [[k, list(v.shape)] for k, v in ad.image_proj.state_dict().items()],
                         "landed": landed}
    # ---- ControlLora ----------------------------------------------------------------------------------------------------------
    unet = SDXLUNet(4, device="meta")
    keys = control_lora_file()
    weights = tagged(keys, device="meta")
    cad = ControlLoraAdapter(name="canny", target=unet, scale=0.7, weights=weights)
    cl = cad.control_lora
    attached = [[a.get_path(), [list(lr.down.weight.shape) for lr in a.loras.values()], list(a.names)] for a in cl.layers(LoraAdapter)]
    zero = [[list(z.state_dict()), [list(v.shape) for v in z.state_dict().values()]] for z in cl.layers(ZeroConvolution)]
    enc = [[k, list(v.shape)] for k, v in cl.ensure_find(ConditionEncoder).state_dict().items()]
    out["control_lora"] = {"keys": [[k, list(s)] for k, s in keys], "attached": attached, "zero": zero, "encoder": enc,
                           "repr_sha256": hashlib.sha256(re.sub(r"Lambda\(.*\)", "Lambda", repr(cl)).encode()).hexdigest()}
    (GOLD / "adapter_wire_sdxl.json").write_text(json.dumps(out))
    print(len(out["ip_adapter"]["keys"]), "IP tensors ->", len(landed), "cross-attentions;", len(out["control_lora"]["keys"]), "ControlLora tensors ->", len(attached), "LoRA sites")


if __name__ == "__main__":
    main()
