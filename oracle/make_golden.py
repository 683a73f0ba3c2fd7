"""Generate tests/golden/* by running the REAL reference (finegrain-ai/refiners, imported read-only from
/root/reference/src) on CPU float32 with the synthetic weights of refiners_amd/synth.py.

Run in the build container only (the GPU box has no /root/reference):
    python oracle/make_golden.py            # ~10 min, needs ~25 GB RAM
    python oracle/make_golden.py --only sdxl_control2     # (re)generate the named cases only; the manifest keeps the others' entries
Outputs (all small, committed):
    tests/golden/sdxl_unet_keys.json, sd1_unet_keys.json   state-dict key -> shape of the reference's bare models
    tests/golden/<case>.safetensors                         reference outputs for the cases in CASES
    tests/golden/manifest.json                              seeds / sizes / adapter recipe of every case + torch version
The oracle (oracle/unet_oracle.py) and the host mirror must reproduce these tensors from the recipe alone
(tests/test_oracle_golden.py, tests/test_mirror_golden.py).
"""
from __future__ import annotations

import json
import sys
import time
from pathlib import Path
from types import SimpleNamespace

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "oracle" / "shim"), "/root/reference/src", str(ROOT)]

import torch  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

import refiners.fluxion.layers as rfl  # noqa: E402
from refiners.fluxion.adapters.lora import Conv2dLora, LinearLora, LoraAdapter  # noqa: E402
from refiners.foundationals.latent_diffusion.solvers.ddim import DDIM  # noqa: E402
from refiners.foundationals.latent_diffusion.stable_diffusion_1.unet import SD1UNet  # noqa: E402
from refiners.foundationals.latent_diffusion.stable_diffusion_xl.control_lora import ConditionEncoder, ControlLoraAdapter, ZeroConvolution  # noqa: E402
from refiners.foundationals.latent_diffusion.stable_diffusion_xl.image_prompt import SDXLIPAdapter  # noqa: E402
from refiners.foundationals.latent_diffusion.stable_diffusion_xl.unet import SDXLUNet  # noqa: E402

from refiners_amd import synth  # noqa: E402
from tests.golden_cases import CASES, build_specs  # noqa: E402

GOLD = ROOT / "tests" / "golden"
REF_API = SimpleNamespace(fl=rfl, LinearLora=LinearLora, Conv2dLora=Conv2dLora, LoraAdapter=LoraAdapter, SDXLIPAdapter=SDXLIPAdapter,
                          ControlLoraAdapter=ControlLoraAdapter, ConditionEncoder=ConditionEncoder, ZeroConvolution=ZeroConvolution)


def reference_model(cls, shapes, seed):
    model = cls(4, device="meta")
    sd = synth.synth_state_dict(shapes, seed)
    model.load_state_dict(sd, assign=True)
    return model


def main() -> None:
    torch.manual_seed(0)
    only = sys.argv[sys.argv.index("--only") + 1 :] if "--only" in sys.argv else None
    manifest = {"torch": torch.__version__, "threads": torch.get_num_threads(), "cases": {}}
    if only and (GOLD / "manifest.json").exists():
        manifest["cases"] = json.loads((GOLD / "manifest.json").read_text())["cases"]
    shapes = {}
    for name, cls in (("sdxl", SDXLUNet), ("sd1", SD1UNet)):
        shapes[name] = synth.model_shapes(cls(4, device="meta"))
        (GOLD / f"{name}_unet_keys.json").write_text(json.dumps({k: list(v) for k, v in shapes[name].items()}))
    models = {}
    with torch.no_grad():
        for case, cfg in CASES.items():
            if only and case not in only:
                continue
            t0 = time.time()
            fam = cfg["family"]
            unet = reference_model(SDXLUNet if fam == "sdxl" else SD1UNet, shapes[fam], cfg["weight_seed"])
            out = {}
            if fam == "sd1":
                x = torch.randn((1, 4, *cfg["latent_hw"]), generator=synth._gen("in.x", cfg["input_seed"]))
                text = torch.randn((1, 77, 768), generator=synth._gen("in.text", cfg["input_seed"]))
                unet.set_timestep(torch.tensor([cfg["timestep"]]))
                unet.set_clip_text_embedding(text)
                out["unet_out"] = unet(x)
                unet.set_timestep(torch.tensor([cfg["timestep"]]))  # context persists: second call must be bit-identical
                out["unet_out_again"] = unet(x)
            else:
                inp = synth.sdxl_inputs(cfg["images"], cfg["latent_hw"], cfg["input_seed"])
                specs = build_specs(cfg, shapes[fam])
                synth.apply_adapters(unet, REF_API, **specs)
                solver = DDIM(num_inference_steps=cfg["num_steps"])
                step = cfg["step"]
                timestep = solver.timesteps[step].unsqueeze(0)
                unet.set_timestep(timestep)
                unet.set_clip_text_embedding(inp["text"])
                unet.set_pooled_text_embedding(inp["pooled"])
                unet.set_time_ids(inp["time_ids"])
                y = unet(torch.cat((inp["x"], inp["x"])))
                u, c = y.chunk(2)
                noise = u + cfg["condition_scale"] * (c - u)
                out["unet_out"] = y
                out["x_next"] = solver(inp["x"], predicted_noise=noise, step=step)
                out["timestep"] = timestep.float()
            save_file({k: v.contiguous() for k, v in out.items()}, str(GOLD / f"{case}.safetensors"))
            manifest["cases"][case] = {**cfg, "abs_mean": float(out["unet_out"].abs().mean()), "seconds": round(time.time() - t0, 1)}
            print(case, manifest["cases"][case], flush=True)
            del unet
    (GOLD / "manifest.json").write_text(json.dumps(manifest, indent=1))


if __name__ == "__main__":
    main()
