"""pytest -m gpu: UNets built from refiners' OWN classes (finegrain-ai/refiners), adapters injected through refiners' own API, run
through CompiledUNet on the MI355X and land on the golden outputs the reference itself produced on CPU.

The reference package is found under REFINERS_SRC, else oracle/_ref/src (staged by __graft_entry__.build() in the build container;
git-ignored, it travels to the GPU box with the snapshot like the built .so), else /root/reference/src.  It is test infrastructure:
nothing under refiners_amd/ or bench.py imports it.  What this covers beyond the dry-lowering equality of
tests/test_reference_tree_cpu.py: the context plumbing against refiners' real UseContext / SetContext nodes
(fluxion/layers/chain.py:645-720: inputs read where the nodes read them, context reset after the call), the per-call tree signature
of epoch-less trees (engine/compiled.py), and the three-line binding INTEGRATION.md proposes for
foundationals/latent_diffusion/model.py:128-159, executed on refiners' own StableDiffusion_XL."""
import os
import sys
from pathlib import Path
from types import SimpleNamespace

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


def _find_reference() -> Path:
    for cand in (os.environ.get("REFINERS_SRC"), ROOT / "oracle" / "_ref" / "src", "/root/reference/src"):
        if cand and (Path(cand) / "refiners").exists():
            return Path(cand)
    return Path("/nonexistent")


REF = _find_reference()
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not (REF / "refiners").exists(), reason="no refiners checkout (REFINERS_SRC / oracle/_ref / /root/reference)")]


def _api():
    sys.path[:0] = [p for p in (str(ROOT / "oracle" / "shim"), str(REF)) if p not in sys.path]
    import refiners.fluxion.layers as rfl
    from refiners.fluxion.adapters.lora import Conv2dLora, LinearLora, LoraAdapter
    from refiners.foundationals.latent_diffusion.stable_diffusion_xl.control_lora import ConditionEncoder, ControlLoraAdapter, ZeroConvolution
    from refiners.foundationals.latent_diffusion.stable_diffusion_xl.image_prompt import SDXLIPAdapter

    assert Path(rfl.__file__).resolve().is_relative_to(REF.resolve()), "refiners was imported from somewhere else"
    return SimpleNamespace(fl=rfl, LinearLora=LinearLora, Conv2dLora=Conv2dLora, LoraAdapter=LoraAdapter, SDXLIPAdapter=SDXLIPAdapter,
                           ControlLoraAdapter=ControlLoraAdapter, ConditionEncoder=ConditionEncoder, ZeroConvolution=ZeroConvolution)


def _build(case):
    from refiners.foundationals.latent_diffusion.stable_diffusion_xl.unet import SDXLUNet

    from refiners_amd import native, synth
    from tests import support as S

    native.load()
    api = _api()
    cfg = S.CASES[case]
    unet = SDXLUNet(4, device="meta")
    unet.load_state_dict({k: v.cuda() for k, v in S.weights("sdxl", cfg["weight_seed"]).items()}, assign=True)
    specs = S.build_specs(cfg, S.key_shapes("sdxl"))
    synth.apply_adapters(unet, api, device="cuda", dtype=torch.float32, **specs)
    inp = {k: v.cuda() for k, v in synth.sdxl_inputs(cfg["images"], cfg["latent_hw"], cfg["input_seed"]).items()}
    return cfg, unet, specs, inp


@pytest.mark.parametrize("case", ["sdxl_bare", "sdxl_lora_ip", "sdxl_control", "sdxl_control2", "sdxl_conv_lora"])
def test_compiled_unet_on_the_real_refiners_tree(gpu_device, case):
    _api()
    from refiners.foundationals.latent_diffusion.solvers import DDIM

    from refiners_amd.engine.compiled import CompiledUNet
    from tests import support as S

    cfg, unet, specs, inp = _build(case)
    assert type(unet).__module__.startswith("refiners.")  # refiners' own class, not the mirror
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


def test_compiled_sd1_unet_on_the_real_refiners_tree(gpu_device):
    _api()
    from refiners.foundationals.latent_diffusion.stable_diffusion_1.unet import SD1UNet

    from refiners_amd import native, synth
    from refiners_amd.engine.compiled import CompiledUNet
    from tests import support as S

    native.load()
    cfg = S.CASES["sd1_bare"]
    unet = SD1UNet(4, device="meta")
    unet.load_state_dict({k: v.cuda() for k, v in S.weights("sd1", cfg["weight_seed"]).items()}, assign=True)
    x = torch.randn((1, 4, *cfg["latent_hw"]), generator=synth._gen("in.x", cfg["input_seed"])).cuda()
    text = torch.randn((1, 77, 768), generator=synth._gen("in.text", cfg["input_seed"])).cuda()
    fast = CompiledUNet(unet)

    def run():
        unet.set_timestep(torch.tensor([cfg["timestep"]], device="cuda"))
        unet.set_clip_text_embedding(text)
        return fast(x)

    y = run()
    l2, mx = S.rel_err(y, S.golden("sd1_bare")["unet_out"])
    print(f"sd1_bare on refiners' own tree: l2 {l2:.2e} max {mx:.2e} ops {fast.stats['step_ops']} fallbacks {fast.stats['fallback_nodes']}")
    assert l2 < 1e-3 and mx < 1e-3 and fast.stats["fallback_nodes"] == []
    assert torch.equal(y, run())  # reference contract: repeated calls are bit-identical (tests/foundationals/latent_diffusion/test_sd15_unet.py:21-37)


class _Bound:
    """What `unet = self._fast_unet or self.unet` of the proposed patch evaluates to: calls go to CompiledUNet, everything else
    (set_timestep, set_clip_text_embedding, ...) to the Chain tree."""

    def __init__(self, unet, fast):
        self._unet, self._fast = unet, fast

    def __call__(self, x):
        return self._fast(x)

    def __getattr__(self, name):
        return getattr(self._unet, name)


@pytest.mark.parametrize("case", ["sdxl_bare", "sdxl_lora_ip"])
def test_the_proposed_binding_on_refiners_own_stable_diffusion_xl(gpu_device, case):
    """INTEGRATION.md section 1 executed: refiners' LatentDiffusionModel.forward (model.py:128-159: set_unet_context, cat(x, x),
    scale_model_input, unet(latents).chunk(2), CFG combine, solver step) runs unchanged with its `self.unet(latents)` bound to
    CompiledUNet; the result is the reference's own x_next."""
    api = _api()
    from refiners.foundationals.latent_diffusion.solvers import DDIM
    from refiners.foundationals.latent_diffusion.stable_diffusion_xl.model import StableDiffusion_XL

    from refiners_amd.engine.compiled import CompiledUNet
    from tests import support as S

    cfg, unet, specs, inp = _build(case)
    sdxl = StableDiffusion_XL(unet=unet, lda=api.fl.Identity(), clip_text_encoder=api.fl.Identity(), solver=DDIM(num_inference_steps=cfg["num_steps"]),
                              device="cuda", dtype=torch.float32)
    fast = CompiledUNet(sdxl.unet)
    sdxl.__dict__["unet"] = _Bound(sdxl.unet, fast)  # the patch's three lines, without editing the checkout
    with torch.no_grad():
        x1 = sdxl(inp["x"], step=cfg["step"], clip_text_embedding=inp["text"], pooled_text_embedding=inp["pooled"], time_ids=inp["time_ids"],
                  condition_scale=cfg["condition_scale"])
    l2, mx = S.rel_err(x1, S.golden(case)["x_next"])
    print(f"{case}: refiners' StableDiffusion_XL.forward over CompiledUNet: x_next l2 {l2:.2e} max {mx:.2e}")
    assert l2 < 1e-3 and mx < 1e-3 and fast.stats["fallback_nodes"] == []
    del sdxl.__dict__["unet"]
    with torch.no_grad():
        x_ref = sdxl(inp["x"], step=cfg["step"], clip_text_embedding=inp["text"], pooled_text_embedding=inp["pooled"], time_ids=inp["time_ids"],
                     condition_scale=cfg["condition_scale"])  # the unpatched reference on the same GPU (stock ATen kernels)
    l2, mx = S.rel_err(x1, x_ref)
    assert l2 < 1e-3 and mx < 1e-3


def test_out_of_scope_adapters_keep_working_on_the_real_refiners_tree(gpu_device):
    """adapters outside the lowered set "must keep working unfused".  refiners' own `SDFreeUAdapter`
    (latent_diffusion/freeu.py:75-104) swaps the ResidualConcatenators of the first UpBlocks for nodes that read `unet.residuals` from the
    context store: the lowering refuses the tree, CompiledUNet warns and runs refiners' stock Chain forward (bit-identical to calling the
    UNet); after eject() the lowered path is back.  A context-free foreign layer (refiners' own fl.Multiply appended to a stage) runs as a
    torch node INSIDE the lowered program."""
    api = _api()
    from refiners.foundationals.latent_diffusion.freeu import SDFreeUAdapter
    from refiners.foundationals.latent_diffusion.solvers import DDIM

    from refiners_amd.engine.compiled import CompiledUNet
    from tests import support as S

    cfg, unet, specs, inp = _build("sdxl_bare")
    xx = torch.cat((inp["x"], inp["x"]))
    fast = CompiledUNet(unet)

    def ctx():
        unet.set_timestep(DDIM(cfg["num_steps"]).timesteps[cfg["step"]].unsqueeze(0).cuda())
        unet.set_clip_text_embedding(inp["text"])
        unet.set_pooled_text_embedding(inp["pooled"])
        unet.set_time_ids(inp["time_ids"])

    freeu = SDFreeUAdapter(unet, backbone_scales=[1.2, 1.2], skip_scales=[0.9, 0.9]).inject()
    ctx()
    with pytest.warns(RuntimeWarning, match="stock Chain forward"):
        y = fast(xx)
    assert "FreeUResidualConcatenator" in fast.stats["whole_fallback"] and fast.stats["fallback_nodes"] == ["<whole UNet>"]
    ctx()
    with torch.no_grad():
        l2, mx = S.rel_err(y, unet(xx))  # the same stock forward (torch's own kernels are not bit-reproducible between calls)
    assert l2 < 1e-5 and mx < 1e-4, (l2, mx)
    l2, _ = S.rel_err(y, S.golden("sdxl_bare")["unet_out"])
    assert l2 > 1e-3  # FreeU does change the output
    freeu.eject()
    ctx()
    y0 = fast(xx)
    l2, mx = S.rel_err(y0, S.golden("sdxl_bare")["unet_out"])
    assert l2 < 1e-3 and mx < 1e-3 and fast.stats["fallback_nodes"] == [] and "whole_fallback" not in fast.stats
    # node-level: a context-free layer of refiners' own
    unet.layer(("DownBlocks", 4), api.fl.Chain).append(api.fl.Multiply(scale=0.8))
    ctx()
    y1 = fast(xx)
    assert fast.stats["fallback_nodes"] == ["Multiply"]
    ctx()
    l2, mx = S.rel_err(y1, unet(xx))
    assert l2 < 1e-3 and mx < 1e-3, (l2, mx)
