"""Host-side mirror of SDXL's prompt encoder (reference: `foundationals/latent_diffusion/stable_diffusion_xl/
text_encoder.py:14-101`).  Output contract: `(B, 77, 2048)` = CLIP-L's penultimate hidden states next to CLIP-G's, and
`(B, 1280)` = CLIP-G's final-layer state at the first end-of-text token, layer-normed and projected -- the
`clip_text_embedding` / `pooled_text_embedding` pair the UNet consumes.  The Chain layout (and so every state-dict key,
tests/golden/double_text_encoder_keys.json) equals the reference's; refiners_amd/engine/text.py lowers it.
"""
from __future__ import annotations

from typing import Any, Optional

import torch
from torch import Tensor

from ..clip import CLIPTextEncoderG, CLIPTextEncoderL, CLIPTokenizer
from ..fluxion import layers as fl
from ..fluxion.adapters import Adapter

POOLING_CONTEXT, POOLING_KEY = "text_encoder_pooling", "end_of_text_index"


class TextEncoderWithPooling(fl.Chain, Adapter[CLIPTextEncoderG]):
    """Adapter that re-wires CLIP-G into two outputs:

        tokens --SetContext(first <end> position per prompt)--> body = target[1:-2] (embeddings + 31 layers)
          +-- Identity                                                        -> hidden states (B, 77, 1280)
          +-- target[-2:] (last layer, final LayerNorm) -> Linear(no bias) -> row at <end> -> pooled (B, 1280)
    """

    def __init__(self, target: CLIPTextEncoderG, projection: Optional[fl.Linear] = None) -> None:
        with self.setup_adapter(target=target):
            if projection is None:
                projection = fl.Linear(1280, 1280, bias=False, device=target.device, dtype=target.dtype)
            remember_eot = fl.SetContext(context=POOLING_CONTEXT, key=POOLING_KEY, callback=self.set_end_of_text_index)
            pooled_branch = fl.Chain(target[-2:], projection, fl.Lambda(func=self.pool))
            super().__init__(target.ensure_find(CLIPTokenizer), remember_eot, target[1:-2], fl.Parallel(fl.Identity(), pooled_branch))

    def init_context(self) -> dict[str, dict[str, Any]]:
        return {POOLING_CONTEXT: {POOLING_KEY: []}}

    @property
    def tokenizer(self) -> CLIPTokenizer:
        return self.ensure_find(CLIPTokenizer)

    def set_end_of_text_index(self, end_of_text_index: list[int], tokens: Tensor) -> None:
        eot = self.tokenizer.end_of_text_token_id
        for row in tokens:
            hits = (row == eot).nonzero()
            assert hits.numel() == 1, "expected exactly one end-of-text token per prompt"  # (.item() in the reference)
            end_of_text_index.append(int(hits[0, 0]))

    def pool(self, x: Tensor) -> Tensor:
        positions = self.use_context(POOLING_CONTEXT).get(POOLING_KEY, [])
        assert len(positions) == x.shape[0], "End of text index not found."
        return torch.cat([x[i : i + 1, p, :] for i, p in enumerate(positions)], dim=0)


class DoubleTextEncoder(fl.Chain):
    """Parallel(CLIP-L without its last layer and final norm, CLIP-G with pooling) -> (cat along channels, pooled)."""

    def __init__(self, text_encoder_l: Optional[CLIPTextEncoderL] = None, text_encoder_g: Optional[CLIPTextEncoderG] = None,
                 projection: Optional[fl.Linear] = None, device: Any = None, dtype: Any = None) -> None:
        enc_l = text_encoder_l if text_encoder_l is not None else CLIPTextEncoderL(device=device, dtype=dtype)
        enc_g = text_encoder_g if text_encoder_g is not None else CLIPTextEncoderG(device=device, dtype=dtype)
        super().__init__(fl.Parallel(enc_l[:-2], enc_g), fl.Lambda(self.concatenate_embeddings))
        TextEncoderWithPooling(target=enc_g, projection=projection).inject(self.layer("Parallel", fl.Parallel))

    def concatenate_embeddings(self, text_embedding_l: Tensor, text_embedding_with_pooling: tuple[Tensor, Tensor]) -> tuple[Tensor, Tensor]:
        hidden_g, pooled = text_embedding_with_pooling
        return torch.cat((text_embedding_l, hidden_g), dim=-1), pooled
