"""Only in the build container (skipped where /root/reference is absent): the engine's lowering accepts trees built from
refiners' OWN classes (it matches on class names), and produces the same program as for the mirror."""
import sys
from collections import Counter
from pathlib import Path
from types import SimpleNamespace

import pytest
import torch

REF = Path("/root/reference/src")
pytestmark = pytest.mark.skipif(not REF.exists(), reason="the reference checkout is only present in the build container")


def test_lowering_accepts_the_real_refiners_tree():
    root = Path(__file__).resolve().parent.parent
    sys.path[:0] = [str(root / "oracle" / "shim"), str(REF)]
    import refiners.fluxion.layers as rfl
    from refiners.fluxion.adapters.lora import Conv2dLora, LinearLora, LoraAdapter
    from refiners.foundationals.latent_diffusion.stable_diffusion_xl.control_lora import ConditionEncoder, ControlLoraAdapter, ZeroConvolution
    from refiners.foundationals.latent_diffusion.stable_diffusion_xl.image_prompt import SDXLIPAdapter
    from refiners.foundationals.latent_diffusion.stable_diffusion_xl.unet import SDXLUNet as RefUNet

    import refiners_amd
    from refiners_amd import synth
    from refiners_amd.latent_diffusion.sdxl import SDXLUNet
    from tests import support as S
    from tests.test_lowering_cpu import _dry

    ref_api = SimpleNamespace(fl=rfl, LinearLora=LinearLora, Conv2dLora=Conv2dLora, LoraAdapter=LoraAdapter, SDXLIPAdapter=SDXLIPAdapter,
                              ControlLoraAdapter=ControlLoraAdapter, ConditionEncoder=ConditionEncoder, ZeroConvolution=ZeroConvolution)
    for case in ("sdxl_lora_ip", "sdxl_control"):
        cfg = S.CASES[case]
        specs = S.build_specs(cfg, S.key_shapes("sdxl"))
        tokens = {("cross_attention_block", "clip_text_embedding"): (77, 2048)}
        if specs["ip"] is not None:
            tokens[("ip_adapter", "clip_image_embedding")] = (4, 2048)
        conds = [f"control_lora_{c['name']}" for c in specs["control"]]
        programs = []
        for cls, api in ((RefUNet, ref_api), (SDXLUNet, refiners_amd.namespace())):
            unet = cls(4, device="meta", dtype=torch.bfloat16)
            synth.apply_adapters(unet, api, device="meta", dtype=torch.bfloat16, **specs)
            low = _dry(unet, 2, *cfg["latent_hw"], torch.bfloat16, tokens, conditions=conds)
            programs.append((Counter(e[2] for e in low.step), Counter(e[2] for e in low.prologue), low.stats["lora_sites"], low.stats["ip_sites"], low.stats["fallback_nodes"]))
        assert programs[0] == programs[1], case
