"""pytest -m gpu: text -> image on the engine, every stage in HBM: prompt tokens -> CompiledDoubleTextEncoder ->
CompiledSDXL (CFG + DDIM loop, HIP-graph replay) -> CompiledVAEDecoder, against the same three trees run unfused by torch
on the same GPU in float32 (what refiners' StableDiffusion_XL does: xl/model.py:62-192).  Synthetic weights; the prompt's
token ids are the reference tokenizer's (tests/golden/double_text_encoder.safetensors)."""
import json

import pytest
import torch

from refiners_amd import native
from refiners_amd.clip import CLIPTokenizer
from refiners_amd.engine.compiled import CompiledSDXL
from refiners_amd.engine.text import CompiledDoubleTextEncoder
from refiners_amd.engine.vae import CompiledVAEDecoder
from refiners_amd.latent_diffusion.sampling import DDIM, SDXLDenoiser
from refiners_amd.latent_diffusion.sdxl import SDXLUNet
from refiners_amd.latent_diffusion.prompt import DoubleTextEncoder
from refiners_amd.latent_diffusion.vae import SDXLAutoencoder
from tests import support as S
from tests.golden_cases import CLIP_CASE, VAE_CASE

pytestmark = pytest.mark.gpu


def test_text_to_image_pipeline_matches_the_unfused_trees(gpu_device):
    native.load()
    dtype, steps, hw = torch.float32, 4, (32, 32)
    gold = S.golden("double_text_encoder")
    # [negative ; conditional] order (model.py:134-141): the golden file holds (long prompt, empty prompt)
    tok_l, tok_g = gold["tokens_l"].flip(0).cuda(), gold["tokens_g"].flip(0).cuda()

    enc = DoubleTextEncoder(device="meta")
    shapes = {k: tuple(v) for k, v in json.loads((S.GOLD / "double_text_encoder_keys.json").read_text()).items()}
    enc.load_state_dict({k: v.to("cuda", dtype) for k, v in S.synth.synth_state_dict(shapes, CLIP_CASE["weight_seed"]).items()}, assign=True)
    unet = SDXLUNet(4, device="meta")
    S.load_mirror_weights(unet, S.weights("sdxl", 0), device="cuda", dtype=dtype)
    vae = SDXLAutoencoder(device="meta")
    vshapes = {k: tuple(v) for k, v in json.loads((S.GOLD / "vae_keys.json").read_text()).items()}
    vae.load_state_dict({k: v.to("cuda", dtype) for k, v in S.synth.synth_state_dict(vshapes, VAE_CASE["weight_seed"]).items()}, assign=True)
    x0 = torch.randn((1, 4, *hw), generator=S.synth._gen("e2e.latents", 3)).cuda()
    time_ids = torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * 2, device="cuda")

    # ---- engine: nothing leaves HBM between the stages ----
    emb, pooled = CompiledDoubleTextEncoder(enc)(tokens=(tok_l, tok_g))
    sd = CompiledSDXL(unet, num_inference_steps=steps, condition_scale=5.0)
    sd.set_inputs(x0, clip_text_embedding=emb, pooled_text_embedding=pooled, time_ids=time_ids)
    latents = sd.sample().clone()
    image = CompiledVAEDecoder(vae)(latents * 0.05)  # synthetic UNet weights do not denoise: keep the decoder input in a sane range

    # ---- the same trees, unfused torch forward ----
    for tk in [m for m in enc.modules() if isinstance(m, CLIPTokenizer)]:
        tk.forward = (lambda t: (lambda _text: (tok_g if t.pad_token_id == 0 else tok_l).long()))(tk)  # type: ignore[method-assign]
    with torch.no_grad():
        emb_ref, pooled_ref = enc(["", CLIP_CASE["prompts"][0]])
        ref = SDXLDenoiser(unet, DDIM(steps, device="cuda"))
        x = x0.clone()
        for s in range(steps):
            x = ref(x, s, clip_text_embedding=emb_ref, pooled_text_embedding=pooled_ref, time_ids=time_ids, condition_scale=5.0)
        image_ref = vae.decode(x * 0.05)
    for name, got, want in (("text embedding", emb, emb_ref), ("pooled", pooled, pooled_ref), ("latents", latents, x), ("image", image, image_ref)):
        l2, mx = S.rel_err(got, want)
        print(f"end to end f32 {name}: l2 {l2:.2e} max {mx:.2e}")
        assert l2 < 1e-3 and mx < 1e-3, (name, l2, mx)
    assert image.shape == (1, 3, 8 * hw[0], 8 * hw[1]) and torch.isfinite(image).all()
