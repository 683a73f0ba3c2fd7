"""Self-Attention Guidance: the host mirror (SDXLSAGAdapter + SDXLDenoiser's second UNet pass) against
the REAL reference's StableDiffusion_XL step (tests/golden/sdxl_sag.safetensors, oracle/make_golden_sag.py), CPU float32; the engine's
version of the same step is checked on the GPU in tests/test_engine_gpu.py::test_self_attention_guidance_on_the_engine."""
import pytest
import torch

import refiners_amd
from refiners_amd import synth
from refiners_amd.latent_diffusion.sag import SDXLSAGAdapter
from refiners_amd.latent_diffusion.sampling import DDIM, SDXLDenoiser
from refiners_amd.latent_diffusion.sdxl import SDXLUNet
from tests import support as S
from tests.golden_cases import SAG_CASE as CFG

TOL = 2e-4


@pytest.mark.parametrize("tag", ["plain", "ip"])
def test_sag_mirror_matches_reference(tag):
    gold = S.golden("sdxl_sag")
    unet = SDXLUNet(4, device="meta")
    S.load_mirror_weights(unet, S.weights("sdxl", CFG["weight_seed"]))
    bare = repr(unet)
    if tag == "ip":
        ip = synth.ip_spec(S.key_shapes("sdxl"), scale=0.6, batch=2, seed=CFG["weight_seed"] + 100)
        synth.apply_adapters(unet, refiners_amd.namespace(), loras=[], ip=ip, control=[])
    sd = SDXLDenoiser(unet, DDIM(CFG["num_steps"]))
    sd.set_self_attention_guidance(True, CFG["sag_scale"])
    assert sd.has_self_attention_guidance() and isinstance(sd._find_sag_adapter(), SDXLSAGAdapter)
    inp = synth.sdxl_inputs(1, CFG["latent_hw"], CFG["input_seed"])
    kw = dict(clip_text_embedding=inp["text"], pooled_text_embedding=inp["pooled"], time_ids=inp["time_ids"], condition_scale=CFG["condition_scale"])
    with torch.no_grad():
        x1 = sd(inp["x"], CFG["step"], **kw)
    l2, mx = S.rel_err(x1, gold[f"x_next_{tag}"])
    assert l2 < TOL and mx < TOL, (tag, l2, mx)
    assert S.rel_err(x1, gold[f"x_next_{tag}_without_sag"])[0] > 5e-3  # the guidance matters in this fixture
    sd.set_self_attention_guidance(False)
    assert not sd.has_self_attention_guidance()
    if tag == "plain":
        assert repr(unet) == bare  # eject restores the tree


def test_sag_mirror_through_dpm_and_lcm_matches_reference():
    """The guidance degrades the latents through Solver.remove_noise / add_noise (self_attention_guidance.py:86-95), so it works with every
    solver whose two maps the reference can evaluate: DPM-Solver++ (step-indexed tables, dpm.py:171-202; two consecutive steps = first- and
    second-order update) and LCMSolver (timestep-indexed, solver.py:244-319; the re-noising draw comes from the seeded global generator).
    The reference itself raises IndexError with Euler (float timesteps as table indices): so does the mirror."""
    from refiners_amd.latent_diffusion.solvers import DPMSolver, Euler, LCMSolver

    gold = S.golden("sdxl_sag_solvers")
    unet = SDXLUNet(4, device="meta")
    S.load_mirror_weights(unet, S.weights("sdxl", CFG["weight_seed"]))
    inp = synth.sdxl_inputs(1, CFG["latent_hw"], CFG["input_seed"])
    kw = dict(clip_text_embedding=inp["text"], pooled_text_embedding=inp["pooled"], time_ids=inp["time_ids"], condition_scale=CFG["condition_scale"])
    sd = SDXLDenoiser(unet, DPMSolver(CFG["num_steps"]))
    sd.set_self_attention_guidance(True, CFG["sag_scale"])
    with torch.no_grad():
        x1 = sd(inp["x"], 0, **kw)
        x2 = sd(x1, 1, **kw)
    for got, key in ((x1, "dpm_x1"), (x2, "dpm_x2")):
        l2, mx = S.rel_err(got, gold[key])
        assert l2 < TOL and mx < TOL, (key, l2, mx)
    sd.solver = LCMSolver(4)
    torch.manual_seed(CFG["input_seed"] + 7)
    with torch.no_grad():
        y1 = sd(inp["x"], 0, **dict(kw, condition_scale=1.5))
    l2, mx = S.rel_err(y1, gold["lcm_x1"])
    assert l2 < TOL and mx < TOL, ("lcm", l2, mx)
    assert S.rel_err(y1, gold["lcm_x1_without_sag"])[0] > 1e-3  # the guidance matters in this fixture
    sd.solver = Euler(CFG["num_steps"])
    with pytest.raises(IndexError), torch.no_grad():
        sd(inp["x"], 0, **kw)


@pytest.mark.parametrize("tag", ["control", "t2i"])
def test_sag_mirror_with_spatial_conditions_matches_reference(tag):
    """ControlLora (own rank-8 LoRA) / SDXLT2IAdapter stay injected for the guidance's second pass (xl/model.py:186-246 swaps embeddings only):
    batch-1 conditions broadcast into the 2n-row CFG pass and into the n-row degraded pass (tests/golden/sdxl_sag_conditions.safetensors,
    oracle/make_golden_sag.py --conditions); a 2n-row control picture fails in the reference's Sum and in the mirror's alike."""
    import json

    from tests.golden_cases import T2I_CASE, control_lora_targets

    gold = S.golden("sdxl_sag_conditions")
    unet = SDXLUNet(4, device="meta")
    S.load_mirror_weights(unet, S.weights("sdxl", CFG["weight_seed"]))
    shapes = S.key_shapes("sdxl")
    if tag == "control":
        own = synth.lora_spec(shapes, "ctl_canny", 1.0, rank=8, seed=CFG["weight_seed"] + 101, targets=control_lora_targets(shapes))
        ctl = synth.control_spec("canny", 0.9, 1, CFG["latent_hw"], seed=CFG["weight_seed"] + 100, loras=[own])
        handles = synth.apply_adapters(unet, refiners_amd.namespace(), loras=[], ip=None, control=[ctl])
    else:
        from refiners_amd.latent_diffusion.t2i import SDXLT2IAdapter

        adapter = SDXLT2IAdapter(unet, name="depth", scale=T2I_CASE["scale"]).inject()
        eshapes = {k: tuple(v) for k, v in json.loads((S.GOLD / "t2i_keys.json").read_text()).items()}
        adapter.condition_encoder.load_state_dict(synth.synth_state_dict(eshapes, T2I_CASE["weight_seed"] + 7), assign=True)
        picture = torch.rand((1, 3, 8 * CFG["latent_hw"][0], 8 * CFG["latent_hw"][1]), generator=synth._gen("t2i.condition", CFG["input_seed"]))
        with torch.no_grad():
            adapter.set_condition_features(adapter.compute_condition_features(picture))
    sd = SDXLDenoiser(unet, DDIM(CFG["num_steps"]))
    sd.set_self_attention_guidance(True, CFG["sag_scale"])
    inp = synth.sdxl_inputs(1, CFG["latent_hw"], CFG["input_seed"])
    kw = dict(clip_text_embedding=inp["text"], pooled_text_embedding=inp["pooled"], time_ids=inp["time_ids"], condition_scale=CFG["condition_scale"])
    with torch.no_grad():
        x1 = sd(inp["x"], CFG["step"], **kw)
    l2, mx = S.rel_err(x1, gold[f"{tag}_x1"])
    assert l2 < TOL and mx < TOL, (tag, l2, mx)
    assert S.rel_err(x1, gold[f"{tag}_x1_without_sag"])[0] > 5e-3  # the guidance matters in this fixture
    if tag == "control":
        handles["control"][0].set_condition(torch.cat([ctl["condition"]] * 2))
        with pytest.raises(RuntimeError), torch.no_grad():
            sd(inp["x"], CFG["step"], **kw)
