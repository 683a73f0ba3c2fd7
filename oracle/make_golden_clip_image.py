"""Golden vectors for the IP-Adapter image-prompt encoder: the REAL reference's
CLIPImageEncoderH + ImageProjection (SDXL widths) on CPU float32, synthetic per-key weights, one 224x224 image.
Run in the build container only:  python oracle/make_golden_clip_image.py"""
from __future__ import annotations

import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "oracle" / "shim"), "/root/reference/src", str(ROOT)]

import torch  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

from refiners.foundationals.clip.image_encoder import CLIPImageEncoderH  # noqa: E402
from refiners.foundationals.latent_diffusion.image_prompt import ImageProjection, IPAdapter, PerceiverResampler  # noqa: E402

from refiners_amd import synth  # noqa: E402
from tests.golden_cases import CLIP_IMAGE_CASE  # noqa: E402

GOLD = ROOT / "tests" / "golden"


def main() -> None:
    enc = CLIPImageEncoderH(device="meta")
    proj = ImageProjection(clip_image_embedding_dim=1024, clip_text_embedding_dim=2048, num_tokens=4, device="meta")
    shapes, pshapes = synth.model_shapes(enc), synth.model_shapes(proj)
    (GOLD / "clip_image_h_keys.json").write_text(json.dumps({"encoder": {k: list(v) for k, v in shapes.items()}, "image_proj": {k: list(v) for k, v in pshapes.items()}}))
    enc.load_state_dict(synth.synth_state_dict(shapes, CLIP_IMAGE_CASE["weight_seed"]), assign=True)
    proj.load_state_dict(synth.synth_state_dict(pshapes, CLIP_IMAGE_CASE["weight_seed"] + 1), assign=True)
    image = torch.randn((1, 3, 224, 224), generator=synth._gen("clip.image", CLIP_IMAGE_CASE["input_seed"]))
    with torch.no_grad():
        emb = enc(image)
        # IPAdapter._compute_clip_image_embedding + compute_clip_image_embedding (image_prompt.py:457-510), one image
        tokens = torch.cat((proj(torch.zeros_like(emb)), proj(emb)))
    # fine-grained ("plus") adapter: penultimate-layer token grid -> PerceiverResampler (SDXL sizes, xl/image_prompt.py:43-53);
    # the negative tokens come from an all-zero IMAGE (image_prompt.py:516-525)
    grid = IPAdapter.convert_to_grid_features(enc)
    res = PerceiverResampler(latents_dim=1280, num_attention_layers=4, num_attention_heads=20, head_dim=64, num_tokens=16, input_dim=1280, output_dim=2048, device="meta")
    rshapes = synth.model_shapes(res)
    keys = json.loads((GOLD / "clip_image_h_keys.json").read_text())
    keys["perceiver"] = {k: list(v) for k, v in rshapes.items()}
    (GOLD / "clip_image_h_keys.json").write_text(json.dumps(keys))
    res.load_state_dict(synth.synth_state_dict(rshapes, CLIP_IMAGE_CASE["weight_seed"] + 2), assign=True)
    with torch.no_grad():
        feats = grid(image)
        plus = torch.cat((res(grid(torch.zeros_like(image))), res(feats)))
    save_file({"embedding": emb.contiguous(), "clip_image_embedding": tokens.contiguous(), "grid_features_sample": feats[:, ::16, ::16].contiguous(),
               "plus_image_embedding": plus.contiguous()}, str(GOLD / "clip_image_h.safetensors"))
    print(tuple(emb.shape), tuple(tokens.shape), float(emb.std()), float(tokens.std()), tuple(plus.shape), float(plus.std()), float((plus[0] - plus[1]).abs().mean()))


if __name__ == "__main__":
    main()
