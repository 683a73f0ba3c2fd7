"""Host-side mirror of SDXL's prompt encoder (`src/refiners/foundationals/latent_diffusion/stable_diffusion_xl/
text_encoder.py:14-101`): CLIP-L's penultimate hidden states next to CLIP-G's, plus CLIP-G's projected end-of-text
embedding -- the (batch, 77, 2048) `clip_text_embedding` and (batch, 1280) `pooled_text_embedding` the UNet consumes.
Same tree and state-dict keys as the reference (tests/golden/double_text_encoder_keys.json)."""
from __future__ import annotations

from typing import Any, Optional, cast

import torch
from torch import Tensor

from ..clip import CLIPTextEncoderG, CLIPTextEncoderL, CLIPTokenizer
from ..fluxion import layers as fl
from ..fluxion.adapters import Adapter


class TextEncoderWithPooling(fl.Chain, Adapter[CLIPTextEncoderG]):
    """CLIP-G re-wired to return (penultimate hidden states, projection of the final-layer state at the first
    end-of-text token)   (`xl/text_encoder.py:14-58`)."""

    def __init__(self, target: CLIPTextEncoderG, projection: Optional[fl.Linear] = None) -> None:
        with self.setup_adapter(target=target):
            tokenizer = target.ensure_find(CLIPTokenizer)
            super().__init__(
                tokenizer,
                fl.SetContext(context="text_encoder_pooling", key="end_of_text_index", callback=self.set_end_of_text_index),
                target[1:-2],
                fl.Parallel(
                    fl.Identity(),
                    fl.Chain(
                        target[-2:],
                        projection or fl.Linear(1280, 1280, bias=False, device=target.device, dtype=target.dtype),
                        fl.Lambda(func=self.pool),
                    ),
                ),
            )

    def init_context(self) -> dict[str, dict[str, Any]]:
        return {"text_encoder_pooling": {"end_of_text_index": []}}

    @property
    def tokenizer(self) -> CLIPTokenizer:
        return self.ensure_find(CLIPTokenizer)

    def set_end_of_text_index(self, end_of_text_index: list[int], tokens: Tensor) -> None:
        for row in torch.split(tokens, 1):
            position = (row == self.tokenizer.end_of_text_token_id).nonzero(as_tuple=True)[1].item()
            end_of_text_index.append(cast(int, position))

    def pool(self, x: Tensor) -> Tensor:
        end_of_text_index = self.use_context("text_encoder_pooling").get("end_of_text_index", [])
        assert len(end_of_text_index) == x.shape[0], "End of text index not found."
        return torch.cat([x[i : i + 1, end_of_text_index[i], :] for i in range(x.shape[0])], dim=0)


class DoubleTextEncoder(fl.Chain):
    """text -> (cat(CLIP-L[-2], CLIP-G[-2]) along channels, pooled CLIP-G)   (`xl/text_encoder.py:61-101`)."""

    def __init__(self, text_encoder_l: Optional[CLIPTextEncoderL] = None, text_encoder_g: Optional[CLIPTextEncoderG] = None,
                 projection: Optional[fl.Linear] = None, device: Any = None, dtype: Any = None) -> None:
        text_encoder_l = text_encoder_l or CLIPTextEncoderL(device=device, dtype=dtype)
        text_encoder_g = text_encoder_g or CLIPTextEncoderG(device=device, dtype=dtype)
        super().__init__(
            fl.Parallel(text_encoder_l[:-2], text_encoder_g),
            fl.Lambda(self.concatenate_embeddings),
        )
        TextEncoderWithPooling(target=text_encoder_g, projection=projection).inject(self.layer("Parallel", fl.Parallel))

    def concatenate_embeddings(self, text_embedding_l: Tensor, text_embedding_with_pooling: tuple[Tensor, Tensor]) -> tuple[Tensor, Tensor]:
        text_embedding_g, pooled_text_embedding = text_embedding_with_pooling
        return torch.cat((text_embedding_l, text_embedding_g), dim=-1), pooled_text_embedding
