"""Test infrastructure (never imported by refiners_amd/ or bench.py): run the REAL reference (finegrain-ai/refiners, imported read-only from
/root/reference/src) ONCE at the benchmarked geometry -- 128 x 128 latents, the recipes of tests/support.py::FULL_SIZE -- and commit its outputs.

    python oracle/make_golden_full_size_reference.py [name ...]      # build container only; ~1-2 min per recipe on 8 cores, ~30 GB RAM

Writes tests/golden/full_size_reference.safetensors: one float32 x_next per recipe (1 x 4 x 128 x 128) computed by refiners' OWN classes
(SDXLUNet Chain forward on the CFG pair, adapters injected through its own API, its own DDIM), the recipe as JSON in the file's metadata.
Until round 6 the full-size GPU tests compared the engine with the CPU oracle only (tests/golden/full_size_oracle.safetensors), and the
reference-written goldens stopped at 32 x 32 latents; with this file `tests/test_oracle_golden.py` pins the oracle to the reference AT FULL SIZE
(two committed tensors compared, no computation) and the `-m gpu` full-size tests compare the HIP path with the reference itself."""
from __future__ import annotations

import json
import sys
import time
from pathlib import Path
from types import SimpleNamespace

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "oracle" / "shim"), "/root/reference/src", str(ROOT)]

import torch  # noqa: E402
from safetensors import safe_open  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

import refiners.fluxion.layers as rfl  # noqa: E402
from refiners.fluxion.adapters.lora import Conv2dLora, LinearLora, LoraAdapter  # noqa: E402
from refiners.foundationals.latent_diffusion.solvers.ddim import DDIM  # noqa: E402
from refiners.foundationals.latent_diffusion.stable_diffusion_xl.control_lora import ConditionEncoder, ControlLoraAdapter, ZeroConvolution  # noqa: E402
from refiners.foundationals.latent_diffusion.stable_diffusion_xl.image_prompt import SDXLIPAdapter  # noqa: E402
from refiners.foundationals.latent_diffusion.stable_diffusion_xl.unet import SDXLUNet  # noqa: E402

from refiners_amd import synth  # noqa: E402
from tests import support as S  # noqa: E402

REF_API = SimpleNamespace(fl=rfl, LinearLora=LinearLora, Conv2dLora=Conv2dLora, LoraAdapter=LoraAdapter, SDXLIPAdapter=SDXLIPAdapter,
                          ControlLoraAdapter=ControlLoraAdapter, ConditionEncoder=ConditionEncoder, ZeroConvolution=ZeroConvolution)


def reference_step(name: str) -> torch.Tensor:
    """x_next of one CFG + DDIM step the way LatentDiffusionModel.forward runs it (latent_diffusion/model.py:128-159), by refiners itself."""
    assert Path(rfl.__file__).resolve().is_relative_to(Path("/root/reference/src").resolve())
    r = S.FULL_SIZE[name]
    specs, inp = S.full_size_inputs(name)
    unet = SDXLUNet(4, device="meta")
    unet.load_state_dict(S.weights("sdxl", r["weight_seed"]), assign=True)
    x, text, pooled, ids = inp["x"], inp["text"], inp["pooled"], inp["time_ids"]
    if "pick" in r:  # one image of the batch: rows `pick` of the [negative ; conditional] stacks
        pick = torch.tensor(r["pick"])
        x, text, pooled, ids = x[:1], text[pick], pooled[pick], ids[pick]
    synth.apply_adapters(unet, REF_API, loras=specs["loras"], ip=specs["ip"], control=specs["control"])
    solver = DDIM(num_inference_steps=r["num_steps"])
    unet.set_timestep(solver.timesteps[r["step"]].unsqueeze(0))
    unet.set_clip_text_embedding(text)
    unet.set_pooled_text_embedding(pooled)
    unet.set_time_ids(ids)
    u, c = unet(torch.cat((x, x))).chunk(2)
    return solver(x, predicted_noise=u + r["condition_scale"] * (c - u), step=r["step"])


def main() -> None:
    path = S.GOLD / "full_size_reference.safetensors"
    tensors, meta = {}, {}
    if path.exists():
        with safe_open(str(path), framework="pt") as f:
            meta = {k: v for k, v in (f.metadata() or {}).items() if k in S.FULL_SIZE}
            tensors = {k: f.get_tensor(k) for k in f.keys()}
    for name in sys.argv[1:] or list(S.FULL_SIZE):
        t0 = time.time()
        with torch.no_grad():
            tensors[name] = reference_step(name).float().contiguous()
        meta[name] = json.dumps(S.FULL_SIZE[name])
        print(name, tuple(tensors[name].shape), f"abs mean {float(tensors[name].abs().mean()):.4f}", f"{time.time() - t0:.0f} s", flush=True)
        save_file(tensors, str(path), metadata={**meta, "torch": torch.__version__, "synth": S.synth_digest(),
                                                "written_by": "finegrain-ai/refiners (its own SDXLUNet / adapters / DDIM), CPU float32: oracle/make_golden_full_size_reference.py"})


if __name__ == "__main__":
    main()
