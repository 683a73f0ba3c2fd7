"""Golden vector for Self-Attention Guidance: the REAL reference's StableDiffusion_XL.forward with
`set_self_attention_guidance(True, scale)` -- CFG pass, attention-map mask, blurred / re-noised latents, second UNet pass,
guidance, DDIM update -- on synthetic weights, CPU float32, one step at 32x32 latents (the tapped attention sees 8x8 = 64 tokens).
With and without the IP-Adapter (the reference halves the image embedding for the second pass, xl/model.py:240-246).
Run in the build container only:  python oracle/make_golden_sag.py"""
from __future__ import annotations

import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "oracle" / "shim"), "/root/reference/src", str(ROOT)]

import torch  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

import refiners.fluxion.layers as rfl  # noqa: E402
from refiners.foundationals.latent_diffusion.solvers import DDIM  # noqa: E402
from refiners.foundationals.latent_diffusion.stable_diffusion_xl.model import StableDiffusion_XL  # noqa: E402
from refiners.foundationals.latent_diffusion.stable_diffusion_xl.unet import SDXLUNet  # noqa: E402

from oracle.make_golden import REF_API, reference_model  # noqa: E402
from refiners_amd import synth  # noqa: E402
from tests.golden_cases import SAG_CASE as CFG  # noqa: E402

GOLD = ROOT / "tests" / "golden"


def main() -> None:
    shapes = synth.model_shapes(SDXLUNet(4, device="meta"))
    out = {}
    with torch.no_grad():
        for tag, with_ip in (("plain", False), ("ip", True)):
            unet = reference_model(SDXLUNet, shapes, CFG["weight_seed"])
            kw = {}
            if with_ip:
                ip = synth.ip_spec(shapes, scale=0.6, batch=2, seed=CFG["weight_seed"] + 100)
                synth.apply_adapters(unet, REF_API, loras=[], ip=ip, control=[])
            sd = StableDiffusion_XL(unet=unet, lda=rfl.Chain(rfl.Identity()), clip_text_encoder=rfl.Chain(rfl.Identity()),  # type: ignore[arg-type]
                                    solver=DDIM(num_inference_steps=CFG["num_steps"]))
            sd.set_self_attention_guidance(enable=True, scale=CFG["sag_scale"])
            inp = synth.sdxl_inputs(1, CFG["latent_hw"], CFG["input_seed"])
            x1 = sd(inp["x"], step=CFG["step"], clip_text_embedding=inp["text"], pooled_text_embedding=inp["pooled"], time_ids=inp["time_ids"],
                    condition_scale=CFG["condition_scale"], **kw)
            sd.set_self_attention_guidance(enable=False)
            x0 = sd(inp["x"], step=CFG["step"], clip_text_embedding=inp["text"], pooled_text_embedding=inp["pooled"], time_ids=inp["time_ids"],
                    condition_scale=CFG["condition_scale"])
            out[f"x_next_{tag}"] = x1.contiguous()
            out[f"x_next_{tag}_without_sag"] = x0.contiguous()
            print(tag, float((x1 - x0).abs().mean()), float(x1.abs().mean()), flush=True)
    save_file(out, str(GOLD / "sdxl_sag.safetensors"))


def main_solvers() -> None:
    """The same guidance through the solvers whose add_noise / remove_noise the reference can evaluate besides DDIM (self_attention_guidance.py:
    86-95 calls solver.remove_noise / add_noise): DPM-Solver++ (two consecutive steps: first- then second-order update) and LCMSolver (one
    step; its re-noising draw comes from the global CPU generator, seeded here).  Euler is not in the file: refiners raises IndexError there
    (float timesteps used as table indices).  -> tests/golden/sdxl_sag_solvers.safetensors"""
    from refiners.foundationals.latent_diffusion.solvers import DPMSolver, LCMSolver

    shapes = synth.model_shapes(SDXLUNet(4, device="meta"))
    out = {}
    with torch.no_grad():
        unet = reference_model(SDXLUNet, shapes, CFG["weight_seed"])
        inp = synth.sdxl_inputs(1, CFG["latent_hw"], CFG["input_seed"])
        kw = dict(clip_text_embedding=inp["text"], pooled_text_embedding=inp["pooled"], time_ids=inp["time_ids"], condition_scale=CFG["condition_scale"])
        sd = StableDiffusion_XL(unet=unet, lda=rfl.Chain(rfl.Identity()), clip_text_encoder=rfl.Chain(rfl.Identity()),  # type: ignore[arg-type]
                                solver=DPMSolver(num_inference_steps=CFG["num_steps"]))
        sd.set_self_attention_guidance(enable=True, scale=CFG["sag_scale"])
        x1 = sd(inp["x"], step=0, **kw)
        x2 = sd(x1, step=1, **kw)
        out["dpm_x1"], out["dpm_x2"] = x1.contiguous(), x2.contiguous()
        print("dpm", float(x1.abs().mean()), float(x2.abs().mean()), flush=True)
        # (a fresh model per solver: re-injecting the adapter into a tree it was ejected from leaves the reference's attention-map context unset)
        unet = reference_model(SDXLUNet, shapes, CFG["weight_seed"])
        sd = StableDiffusion_XL(unet=unet, lda=rfl.Chain(rfl.Identity()), clip_text_encoder=rfl.Chain(rfl.Identity()),  # type: ignore[arg-type]
                                solver=LCMSolver(num_inference_steps=4))
        sd.set_self_attention_guidance(enable=True, scale=CFG["sag_scale"])
        torch.manual_seed(CFG["input_seed"] + 7)
        out["lcm_x1"] = sd(inp["x"], step=0, **dict(kw, condition_scale=1.5)).contiguous()
        sd.set_self_attention_guidance(enable=True, scale=0.0)  # the guidance term times zero
        torch.manual_seed(CFG["input_seed"] + 7)
        out["lcm_x1_without_sag"] = sd(inp["x"], step=0, **dict(kw, condition_scale=1.5)).contiguous()
        print("lcm", float((out["lcm_x1"] - out["lcm_x1_without_sag"]).abs().mean()), flush=True)
        unet = reference_model(SDXLUNet, shapes, CFG["weight_seed"])
        sd = StableDiffusion_XL(unet=unet, lda=rfl.Chain(rfl.Identity()), clip_text_encoder=rfl.Chain(rfl.Identity()))  # type: ignore[arg-type]
        try:
            from refiners.foundationals.latent_diffusion.solvers import Euler

            sd.solver = Euler(num_inference_steps=CFG["num_steps"])
            sd.set_self_attention_guidance(enable=True, scale=CFG["sag_scale"])
            sd(inp["x"], step=0, **kw)
            raise SystemExit("the reference evaluated SAG with Euler: the mirror's refusal is wrong")
        except IndexError as e:
            print("euler: the reference raises IndexError:", str(e)[:80])
    save_file(out, str(GOLD / "sdxl_sag_solvers.safetensors"))


def main_conditions() -> None:
    """The guidance together with spatial conditions.  The second (degraded) UNet pass runs on n rows while the CFG pass ran on 2n, and the
    ControlLora / T2I-Adapter stay injected with their contexts set (xl/model.py:186-246 only swaps the text / pooled / time-id / image
    embeddings): a batch-1 control picture / batch-1 T2I features broadcast into both passes, a 2n-row control picture cannot be added to an
    n-row batch.  Stored: one DDIM step each for ControlLora (own rank-8 LoRA, scale 0.9) and SDXLT2IAdapter with batch-1 conditions, with
    the guidance and with its scale at zero; the 2n-row ControlLora case is run to confirm that the reference raises.
    -> tests/golden/sdxl_sag_conditions.safetensors"""
    from refiners.foundationals.latent_diffusion.stable_diffusion_xl.t2i_adapter import SDXLT2IAdapter

    from tests.golden_cases import T2I_CASE, control_lora_targets

    shapes = synth.model_shapes(SDXLUNet(4, device="meta"))
    out = {}
    inp = synth.sdxl_inputs(1, CFG["latent_hw"], CFG["input_seed"])
    kw = dict(clip_text_embedding=inp["text"], pooled_text_embedding=inp["pooled"], time_ids=inp["time_ids"], condition_scale=CFG["condition_scale"])

    def pipeline(unet):  # noqa: ANN001, ANN202
        return StableDiffusion_XL(unet=unet, lda=rfl.Chain(rfl.Identity()), clip_text_encoder=rfl.Chain(rfl.Identity()),  # type: ignore[arg-type]
                                  solver=DDIM(num_inference_steps=CFG["num_steps"]))

    with torch.no_grad():
        for batch in (1, 2):
            unet = reference_model(SDXLUNet, shapes, CFG["weight_seed"])
            own = synth.lora_spec(shapes, "ctl_canny", 1.0, rank=8, seed=CFG["weight_seed"] + 101, targets=control_lora_targets(shapes))
            ctl = synth.control_spec("canny", 0.9, batch, CFG["latent_hw"], seed=CFG["weight_seed"] + 100, loras=[own])
            synth.apply_adapters(unet, REF_API, loras=[], ip=None, control=[ctl])
            sd = pipeline(unet)
            sd.set_self_attention_guidance(enable=True, scale=CFG["sag_scale"])
            if batch == 2:
                try:
                    sd(inp["x"], step=CFG["step"], **kw)
                    raise SystemExit("the reference evaluated SAG with a 2n-row control picture: the engine's refusal is wrong")
                except RuntimeError as e:
                    print("control picture with 2n rows: the reference raises RuntimeError:", str(e)[:120])
                continue
            out["control_x1"] = sd(inp["x"], step=CFG["step"], **kw).contiguous()
            sd.set_self_attention_guidance(enable=True, scale=0.0)
            out["control_x1_without_sag"] = sd(inp["x"], step=CFG["step"], **kw).contiguous()
            print("control", float((out["control_x1"] - out["control_x1_without_sag"]).abs().mean()), float(out["control_x1"].abs().mean()), flush=True)
        unet = reference_model(SDXLUNet, shapes, CFG["weight_seed"])
        adapter = SDXLT2IAdapter(unet, name="depth", scale=T2I_CASE["scale"]).inject()
        eshapes = synth.model_shapes(adapter.condition_encoder)
        adapter.condition_encoder.load_state_dict(synth.synth_state_dict(eshapes, T2I_CASE["weight_seed"] + 7), assign=True)
        picture = torch.rand((1, 3, 8 * CFG["latent_hw"][0], 8 * CFG["latent_hw"][1]), generator=synth._gen("t2i.condition", CFG["input_seed"]))
        adapter.set_condition_features(adapter.compute_condition_features(picture))
        sd = pipeline(unet)
        sd.set_self_attention_guidance(enable=True, scale=CFG["sag_scale"])
        out["t2i_x1"] = sd(inp["x"], step=CFG["step"], **kw).contiguous()
        sd.set_self_attention_guidance(enable=True, scale=0.0)
        out["t2i_x1_without_sag"] = sd(inp["x"], step=CFG["step"], **kw).contiguous()
        print("t2i", float((out["t2i_x1"] - out["t2i_x1_without_sag"]).abs().mean()), float(out["t2i_x1"].abs().mean()), flush=True)
    save_file(out, str(GOLD / "sdxl_sag_conditions.safetensors"))


if __name__ == "__main__":
    if "--conditions" in sys.argv:
        main_conditions()
    elif "--solvers" in sys.argv:
        main_solvers()
    else:
        main()
