"""VAE decode: CPU oracle and host mirror vs the real reference's output."""
import json

import pytest
import torch

from oracle import vae_oracle
from refiners_amd import synth
from refiners_amd.latent_diffusion.vae import SDXLAutoencoder
from tests import support as S
from tests.golden_cases import VAE_CASE

TOL = 2e-4


@pytest.fixture(scope="module")
def vae_inputs():
    shapes = {k: tuple(v) for k, v in json.loads((S.GOLD / "vae_keys.json").read_text()).items()}
    sd = synth.synth_state_dict(shapes, VAE_CASE["weight_seed"])
    z = torch.randn((1, 4, *VAE_CASE["latent_hw"]), generator=synth._gen("vae.latents", VAE_CASE["input_seed"])) * VAE_CASE["latent_std"]
    return shapes, sd, z


def test_vae_oracle_matches_reference(vae_inputs):
    _, sd, z = vae_inputs
    l2, mx = S.rel_err(vae_oracle.vae_decode(sd, z), S.golden("vae_decode")["image"])
    assert l2 < TOL and mx < TOL, (l2, mx)


def test_vae_mirror_matches_reference(vae_inputs):
    shapes, sd, z = vae_inputs
    vae = SDXLAutoencoder(device="meta")
    assert {k: tuple(v.shape) for k, v in vae.state_dict().items()} == shapes and list(vae.state_dict()) == list(shapes)
    vae.load_state_dict(sd, assign=True)
    with torch.no_grad():
        img = vae.decode(z)
    l2, mx = S.rel_err(img, S.golden("vae_decode")["image"])
    assert l2 < TOL and mx < TOL, (l2, mx)


def test_vae_encode_oracle_and_mirror_match_reference(vae_inputs):
    _, sd, _ = vae_inputs
    pic = torch.rand((1, 3, 8 * VAE_CASE["latent_hw"][0], 8 * VAE_CASE["latent_hw"][1]), generator=synth._gen("vae.image", VAE_CASE["input_seed"])) * 2 - 1
    gold = S.golden("vae_encode")["latents"]
    l2, mx = S.rel_err(vae_oracle.vae_encode(sd, pic), gold)
    assert l2 < TOL and mx < TOL, (l2, mx)
    vae = SDXLAutoencoder(device="meta")
    vae.load_state_dict(sd, assign=True)
    with torch.no_grad():
        lat = vae.encode(pic)
    l2, mx = S.rel_err(lat, gold)
    assert l2 < TOL and mx < TOL, (l2, mx)
