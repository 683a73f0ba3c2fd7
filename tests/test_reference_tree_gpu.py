"""pytest -m gpu, only where a checkout of finegrain-ai/refiners is reachable (REFINERS_SRC=<.../src>, default /root/reference/src;
the GPU boxes of this project have none, so the test skips there): a UNet built from refiners' OWN classes, adapters injected
through refiners' own API, runs through CompiledUNet and lands on the same golden output as the mirror.  What this covers
beyond the dry-lowering equality of tests/test_reference_tree_cpu.py: the context plumbing (inputs read where refiners'
UseContext nodes read them, context reset after the call) and the per-call tree signature of epoch-less trees."""
import os
import sys
from pathlib import Path
from types import SimpleNamespace

import pytest
import torch

REF = Path(os.environ.get("REFINERS_SRC", "/root/reference/src"))
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not (REF / "refiners").exists(), reason="no refiners checkout (set REFINERS_SRC)")]


@pytest.mark.parametrize("case", ["sdxl_bare", "sdxl_lora_ip"])
def test_compiled_unet_on_the_real_refiners_tree(gpu_device, case):
    root = Path(__file__).resolve().parent.parent
    sys.path[:0] = [str(root / "oracle" / "shim"), str(REF)]
    import refiners.fluxion.layers as rfl
    from refiners.fluxion.adapters.lora import Conv2dLora, LinearLora, LoraAdapter
    from refiners.foundationals.latent_diffusion.solvers import DDIM
    from refiners.foundationals.latent_diffusion.stable_diffusion_xl.control_lora import ConditionEncoder, ControlLoraAdapter, ZeroConvolution
    from refiners.foundationals.latent_diffusion.stable_diffusion_xl.image_prompt import SDXLIPAdapter
    from refiners.foundationals.latent_diffusion.stable_diffusion_xl.unet import SDXLUNet

    from refiners_amd import native, synth
    from refiners_amd.engine.compiled import CompiledUNet
    from tests import support as S

    native.load()
    api = SimpleNamespace(fl=rfl, LinearLora=LinearLora, Conv2dLora=Conv2dLora, LoraAdapter=LoraAdapter, SDXLIPAdapter=SDXLIPAdapter,
                          ControlLoraAdapter=ControlLoraAdapter, ConditionEncoder=ConditionEncoder, ZeroConvolution=ZeroConvolution)
    cfg = S.CASES[case]
    unet = SDXLUNet(4, device="meta")
    unet.load_state_dict({k: v.cuda() for k, v in S.weights("sdxl", cfg["weight_seed"]).items()}, assign=True)
    specs = S.build_specs(cfg, S.key_shapes("sdxl"))
    synth.apply_adapters(unet, api, device="cuda", dtype=torch.float32, **specs)
    inp = {k: v.cuda() for k, v in synth.sdxl_inputs(cfg["images"], cfg["latent_hw"], cfg["input_seed"]).items()}
    fast = CompiledUNet(unet)
    xx = torch.cat((inp["x"], inp["x"]))

    def run():
        unet.set_timestep(DDIM(cfg["num_steps"]).timesteps[cfg["step"]].unsqueeze(0).cuda())
        unet.set_clip_text_embedding(inp["text"])
        unet.set_pooled_text_embedding(inp["pooled"])
        unet.set_time_ids(inp["time_ids"])
        return fast(xx)

    y = run()
    l2, mx = S.rel_err(y, S.golden(case)["unet_out"])
    print(f"{case} on refiners' own tree: l2 {l2:.2e} max {mx:.2e} ops {fast.stats['step_ops']} fallbacks {fast.stats['fallback_nodes']}")
    assert l2 < 1e-3 and mx < 1e-3 and fast.stats["fallback_nodes"] == []
    assert torch.equal(y, run())  # second call: same program (the tree signature is stable), context re-read
    with pytest.raises(Exception):
        fast(xx)  # the context was reset after the call, exactly like Chain.forward leaves it: a forward without set_timestep fails
