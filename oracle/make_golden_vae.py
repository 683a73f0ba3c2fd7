"""Golden vectors for the VAE decode step: the REAL reference's SDXLAutoencoder.decode on CPU
float32 with the synthetic per-key weights of refiners_amd/synth.py, 16x24 latents -> (1, 3, 128, 192).
Run in the build container only:  python oracle/make_golden_vae.py"""
from __future__ import annotations

import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "oracle" / "shim"), "/root/reference/src", str(ROOT)]

import torch  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

from refiners.foundationals.latent_diffusion.stable_diffusion_xl.model import SDXLAutoencoder  # noqa: E402

from refiners_amd import synth  # noqa: E402
from tests.golden_cases import VAE_CASE  # noqa: E402

GOLD = ROOT / "tests" / "golden"


def main() -> None:
    vae = SDXLAutoencoder(device="meta")
    shapes = synth.model_shapes(vae)
    (GOLD / "vae_keys.json").write_text(json.dumps({k: list(v) for k, v in shapes.items()}))
    vae.load_state_dict(synth.synth_state_dict(shapes, VAE_CASE["weight_seed"]), assign=True)
    z = torch.randn((1, 4, *VAE_CASE["latent_hw"]), generator=synth._gen("vae.latents", VAE_CASE["input_seed"])) * VAE_CASE["latent_std"]
    with torch.no_grad():
        img = vae.decode(z)
    save_file({"image": img.contiguous()}, str(GOLD / "vae_decode.safetensors"))
    print(tuple(img.shape), float(img.abs().mean()), float(img.std()))
    pic = torch.rand((1, 3, 8 * VAE_CASE["latent_hw"][0], 8 * VAE_CASE["latent_hw"][1]), generator=synth._gen("vae.image", VAE_CASE["input_seed"])) * 2 - 1
    with torch.no_grad():
        lat = vae.encode(pic)
    save_file({"latents": lat.contiguous()}, str(GOLD / "vae_encode.safetensors"))
    print(tuple(lat.shape), float(lat.abs().mean()), float(lat.std()))


if __name__ == "__main__":
    main()
