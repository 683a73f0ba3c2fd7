"""The host mirror (refiners_amd.fluxion Chain trees, unfused torch path, CPU float32) against the reference outputs
in tests/golden/: same state-dict keys, same adapter API, same numbers."""
import re

import pytest
import torch

import refiners_amd
from refiners_amd.latent_diffusion.sampling import DDIM, SDXLDenoiser
from refiners_amd.latent_diffusion.sd1 import SD1UNet
from refiners_amd.latent_diffusion.sdxl import SDXLUNet
from tests import support as S

TOL = 2e-4


def test_state_dict_keys_match_reference():
    for family, cls in (("sdxl", SDXLUNet), ("sd1", SD1UNet)):
        mine = {k: tuple(v.shape) for k, v in cls(4, device="meta").state_dict().items()}
        ref = S.key_shapes(family)
        assert list(mine) == list(ref) and mine == ref


@pytest.mark.parametrize("case", [c for c, cfg in S.CASES.items() if cfg["family"] == "sdxl"])
def test_sdxl_mirror_matches_reference(case):
    cfg = S.CASES[case]
    gold = S.golden(case)
    with torch.no_grad():
        unet = SDXLUNet(4, device="meta")
        S.load_mirror_weights(unet, S.weights("sdxl", cfg["weight_seed"]))
        before = repr(unet)
        handles = S.synth.apply_adapters(unet, refiners_amd.namespace(), **S.build_specs(cfg, S.key_shapes("sdxl")))
        inp = S.synth.sdxl_inputs(cfg["images"], cfg["latent_hw"], cfg["input_seed"])
        sd = SDXLDenoiser(unet, DDIM(cfg["num_steps"]))
        x_next = sd(inp["x"], cfg["step"], clip_text_embedding=inp["text"], pooled_text_embedding=inp["pooled"], time_ids=inp["time_ids"],
                    condition_scale=cfg["condition_scale"])
    l2, mx = S.rel_err(x_next, gold["x_next"])
    assert l2 < TOL and mx < TOL, (case, l2, mx)
    # inject -> eject leaves the tree as it was (reference tests/adapters/test_ip_adapter.py:27-41, test_control_lora.py:9-17)
    for a in handles["loras"]:
        a.eject()
    if handles["ip"] is not None:
        handles["ip"].eject()
    for a in reversed(handles["control"]):
        a.eject()
    assert repr(unet) == before


def test_sd1_mirror_matches_reference_and_repeats_bit_exactly():
    cfg = S.CASES["sd1_bare"]
    gold = S.golden("sd1_bare")
    with torch.no_grad():
        unet = SD1UNet(4, device="meta")
        S.load_mirror_weights(unet, S.weights("sd1", cfg["weight_seed"]))
        x = torch.randn((1, 4, *cfg["latent_hw"]), generator=S.synth._gen("in.x", cfg["input_seed"]))
        text = torch.randn((1, 77, 768), generator=S.synth._gen("in.text", cfg["input_seed"]))
        unet.set_clip_text_embedding(text)
        unet.set_timestep(torch.tensor([cfg["timestep"]]))
        y1 = unet(x)
        unet.set_timestep(torch.tensor([cfg["timestep"]]))  # the text context persists (test_sd15_unet.py:21-37)
        y2 = unet(x)
    assert torch.equal(y1, y2)
    l2, mx = S.rel_err(y1, gold["unet_out"])
    assert l2 < TOL and mx < TOL, (l2, mx)
