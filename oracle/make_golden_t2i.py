"""Golden vectors for the T2I-Adapter: the REAL reference's SDXLUNet + SDXLT2IAdapter on CPU float32,
synthetic per-key weights, one 256x256 conditioning picture -> four feature maps -> one CFG UNet forward at 32x32 latents.
Run in the build container only:  python oracle/make_golden_t2i.py"""
from __future__ import annotations

import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "oracle" / "shim"), "/root/reference/src", str(ROOT)]

import torch  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

from refiners.foundationals.latent_diffusion.solvers.ddim import DDIM  # noqa: E402
from refiners.foundationals.latent_diffusion.stable_diffusion_xl.t2i_adapter import SDXLT2IAdapter  # noqa: E402
from refiners.foundationals.latent_diffusion.stable_diffusion_xl.unet import SDXLUNet  # noqa: E402

from refiners_amd import synth  # noqa: E402
from tests.golden_cases import T2I_CASE  # noqa: E402

GOLD = ROOT / "tests" / "golden"


def main() -> None:
    cfg = T2I_CASE
    shapes = {k: tuple(v) for k, v in json.loads((GOLD / "sdxl_unet_keys.json").read_text()).items()}
    unet = SDXLUNet(4, device="meta")
    unet.load_state_dict(synth.synth_state_dict(shapes, cfg["weight_seed"]), assign=True)
    adapter = SDXLT2IAdapter(unet, name="depth", scale=cfg["scale"]).inject()
    eshapes = synth.model_shapes(adapter.condition_encoder)
    (GOLD / "t2i_keys.json").write_text(json.dumps({k: list(v) for k, v in eshapes.items()}))
    # the encoder was built on the meta device by the adapter: give it real synthetic weights
    adapter.condition_encoder.load_state_dict(synth.synth_state_dict(eshapes, cfg["weight_seed"] + 7), assign=True)
    inp = synth.sdxl_inputs(1, cfg["latent_hw"], cfg["input_seed"])
    picture = torch.rand((1, 3, 8 * cfg["latent_hw"][0], 8 * cfg["latent_hw"][1]), generator=synth._gen("t2i.condition", cfg["input_seed"]))
    with torch.no_grad():
        features = adapter.compute_condition_features(picture)
        adapter.set_condition_features(features)
        unet.set_timestep(DDIM(num_inference_steps=cfg["num_steps"]).timesteps[cfg["step"]].unsqueeze(0))
        unet.set_clip_text_embedding(inp["text"])
        unet.set_pooled_text_embedding(inp["pooled"])
        unet.set_time_ids(inp["time_ids"])
        y = unet(torch.cat((inp["x"], inp["x"])))
    out = {"unet_out": y.contiguous()}
    for i, f in enumerate(features):  # strided samples + moments keep the fixture small
        out[f"feature_{i}"] = f[:, ::4, ::2, ::2].contiguous()
        out[f"feature_{i}_stats"] = torch.stack([f.mean(), f.std(), f.abs().max()])
    save_file(out, str(GOLD / "sdxl_t2i.safetensors"))
    print([tuple(f.shape) for f in features], float(y.abs().mean()), [float(f.std()) for f in features])


if __name__ == "__main__":
    main()
