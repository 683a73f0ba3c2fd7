"""SD1.5 ControlNet: CPU oracle and host mirror vs the real reference's SD1UNet +
SD1ControlnetAdapter; the engine's lowering of the tree is checked here as a dry run (meta device) and on the GPU in
tests/test_engine_gpu.py::test_sd1_controlnet_on_the_engine."""
import json

import pytest
import torch

from oracle import unet_oracle
from refiners_amd import synth
from refiners_amd.latent_diffusion.controlnet import SD1ControlnetAdapter
from refiners_amd.latent_diffusion.sd1 import SD1UNet
from tests import support as S
from tests.golden_cases import CONTROLNET_CASE as CFG

TOL = 2e-4


@pytest.fixture(scope="module")
def cn_inputs():
    cshapes = {k: tuple(v) for k, v in json.loads((S.GOLD / "sd1_controlnet_keys.json").read_text()).items()}
    csd = synth.synth_state_dict(cshapes, CFG["weight_seed"] + 11)
    h, w = CFG["latent_hw"]
    x = torch.randn((1, 4, h, w), generator=synth._gen("in.x", CFG["input_seed"]))
    text = torch.randn((1, 77, 768), generator=synth._gen("in.text", CFG["input_seed"]))
    picture = torch.rand((1, 3, 8 * h, 8 * w), generator=synth._gen("controlnet.condition", CFG["input_seed"]))
    return cshapes, csd, x, text, picture, S.golden("sd1_controlnet")


def test_controlnet_oracle_matches_reference(cn_inputs):
    _, csd, x, text, picture, gold = cn_inputs
    sd = S.weights("sd1", CFG["weight_seed"])
    y = unet_oracle.sd1_unet(sd, x, torch.tensor([CFG["timestep"]]), text,
                             controlnets=[dict(weights=csd, condition=picture, scale=CFG["scale"], scale_decay=CFG["scale_decay"])])
    l2, mx = S.rel_err(y, gold["unet_out"])
    assert l2 < TOL and mx < TOL, (l2, mx)
    bare = unet_oracle.sd1_unet(sd, x, torch.tensor([CFG["timestep"]]), text)
    assert S.rel_err(bare, gold["unet_out"])[0] > 1e-2  # the branch matters in this fixture


def test_controlnet_mirror_matches_reference(cn_inputs):
    cshapes, csd, x, text, picture, gold = cn_inputs
    unet = SD1UNet(4, device="meta")
    S.load_mirror_weights(unet, S.weights("sd1", CFG["weight_seed"]))
    adapter = SD1ControlnetAdapter(unet, name="canny", scale=CFG["scale"], scale_decay=CFG["scale_decay"])
    assert {k: tuple(v.shape) for k, v in adapter.controlnet.state_dict().items()} == cshapes and list(adapter.controlnet.state_dict()) == list(cshapes)
    adapter.controlnet.load_state_dict(csd, assign=True)
    adapter.inject()
    with torch.no_grad():
        adapter.set_controlnet_condition(picture)
        unet.set_timestep(torch.tensor([CFG["timestep"]]))
        unet.set_clip_text_embedding(text)
        y = unet(x)
    l2, mx = S.rel_err(y, gold["unet_out"])
    assert l2 < TOL and mx < TOL, (l2, mx)
    # dry lowering (meta device): the branch becomes launches -- 13 residual taps -- and nothing falls back; without its
    # condition image the engine refuses loudly rather than skip the branch
    from refiners_amd.engine.packing import Unsupported, launches
    from refiners_amd.engine.unet_lowering import UNetIO, UNetLowering

    dev = torch.device("meta")

    def io_for(with_condition: bool) -> UNetIO:
        io = UNetIO(x=torch.empty(1, 4, 16, 16, device=dev), timestep=torch.empty(1, device=dev), out=torch.empty(1, 4, 16, 16, device=dev))
        io.tokens[("cross_attention_block", "clip_text_embedding")] = (torch.zeros(128, 768, device=dev), 77)
        if with_condition:
            io.conditions["controlnet.condition_canny"] = torch.empty(1, 3, 128, 128, device=dev)
        return io

    meta_unet = SD1UNet(4, device="meta")
    bare = UNetLowering(dev, torch.float32, None, "merged")
    bare.lower(meta_unet, io_for(False))
    SD1ControlnetAdapter(meta_unet, name="canny").inject()
    low = UNetLowering(dev, torch.float32, None, "merged")
    low.lower(meta_unet, io_for(True))
    assert low.stats["controlnets"] == 1 and low.stats["fallback_nodes"] == []
    assert launches(low.step) > launches(bare.step) + 13 and launches(low.prologue) >= 8  # the condition encoder runs once per image
    with pytest.raises(Unsupported):
        UNetLowering(dev, torch.float32, None, "merged").lower(meta_unet, io_for(False))
    adapter.eject()
