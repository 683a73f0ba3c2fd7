"""Golden vectors for the prompt encoder: the REAL reference's DoubleTextEncoder on CPU float32
with the synthetic per-key weights of refiners_amd/synth.py, on two prompts (a long one and the empty negative prompt).
Stores the reference tokenizer's token ids too, so that nothing at test time needs the BPE vocabulary file.
Run in the build container only:  python oracle/make_golden_clip.py"""
from __future__ import annotations

import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "oracle" / "shim"), "/root/reference/src", str(ROOT)]

import torch  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

from refiners.foundationals.clip.tokenizer import CLIPTokenizer  # noqa: E402
from refiners.foundationals.latent_diffusion.stable_diffusion_xl.text_encoder import DoubleTextEncoder  # noqa: E402

from refiners_amd import synth  # noqa: E402
from tests.golden_cases import CLIP_CASE  # noqa: E402

GOLD = ROOT / "tests" / "golden"


def main() -> None:
    enc = DoubleTextEncoder(device="meta")
    shapes = synth.model_shapes(enc)
    (GOLD / "double_text_encoder_keys.json").write_text(json.dumps({k: list(v) for k, v in shapes.items()}))
    enc.load_state_dict(synth.synth_state_dict(shapes, CLIP_CASE["weight_seed"]), assign=True)
    prompts = list(CLIP_CASE["prompts"])
    with torch.no_grad():
        text_embedding, pooled = enc(prompts)
    tok_l = CLIPTokenizer()(prompts)
    tok_g = CLIPTokenizer(pad_token_id=0)(prompts)
    save_file({"tokens_l": tok_l.to(torch.int32), "tokens_g": tok_g.to(torch.int32), "text_embedding": text_embedding.contiguous(), "pooled": pooled.contiguous()},
              str(GOLD / "double_text_encoder.safetensors"))
    print(tuple(text_embedding.shape), tuple(pooled.shape), float(text_embedding.abs().mean()), float(text_embedding.std()), float(pooled.std()))
    print(tok_l[0, :12].tolist(), tok_g[1, :6].tolist())


if __name__ == "__main__":
    main()
