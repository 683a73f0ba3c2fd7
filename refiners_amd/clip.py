"""Host-side mirror of refiners' CLIP text encoders (`src/refiners/foundationals/clip/text_encoder.py:8-251`,
`common.py:7-49`, `tokenizer.py:13-129`) -- SURVEY.md section 8(f) next-2, the step before the denoising loop.

Same Chain layout, same constructor arguments and therefore the same state-dict keys as the reference
(`tests/golden/clip_text_keys.json`); the MI355X engine (`refiners_amd/engine/text.py`) lowers these trees (or the
reference's own) onto the hand-written kernels.  The BPE tokenizer is host code: it is re-implemented here from the
published CLIP byte-pair algorithm and needs the OpenAI vocabulary file (`bpe_simple_vocab_16e6.txt.gz`, 1.3 MB, a data
file that ships with refiners and with openai/CLIP; it is NOT vendored in this repository -- pass `vocabulary_path`, set
`REFINERS_AMD_CLIP_VOCAB`, or install refiners).  The engine boundary takes token ids, so nothing on the GPU path
depends on it.
"""
from __future__ import annotations

import gzip
import os
import re
from pathlib import Path
from typing import Any, Optional

import torch
from torch import Tensor, nn

from .fluxion import layers as fl
from .fluxion.tree import WeightedModule


class Embedding(nn.Embedding, WeightedModule):
    """fl.Embedding (`fluxion/layers/embedding.py:7-43`): a torch.nn.Embedding that is a WeightedModule."""

    def __init__(self, num_embeddings: int, embedding_dim: int, device: Any = None, dtype: Any = None) -> None:
        nn.Embedding.__init__(self, num_embeddings=num_embeddings, embedding_dim=embedding_dim, device=device, dtype=dtype)


def _vocabulary_candidates() -> list[Path]:
    out = []
    if os.environ.get("REFINERS_AMD_CLIP_VOCAB"):
        out.append(Path(os.environ["REFINERS_AMD_CLIP_VOCAB"]))
    try:  # an installed refiners ships the file next to its tokenizer
        import importlib.util

        spec = importlib.util.find_spec("refiners")
        if spec is not None and spec.submodule_search_locations:
            for loc in spec.submodule_search_locations:
                out.append(Path(loc) / "foundationals" / "clip" / "bpe_simple_vocab_16e6.txt.gz")
    except Exception:  # pragma: no cover
        pass
    return out


def find_vocabulary() -> Optional[Path]:
    for p in _vocabulary_candidates():
        if p.is_file():
            return p
    return None


class CLIPTokenizer(fl.Module):
    """CLIP's lower-cased byte-level BPE (`clip/tokenizer.py:13-129`): text -> (1, sequence_length) int64 token ids,
    <start> ... <end> then padding."""

    _pattern = re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[a-zA-Z]+|[0-9]|(?:[^\s\w]|_)+", re.IGNORECASE)

    def __init__(self, vocabulary_path: str | Path | None = None, sequence_length: int = 77, start_of_text_token_id: int = 49406,
                 end_of_text_token_id: int = 49407, pad_token_id: int = 49407) -> None:
        super().__init__()
        self.vocabulary_path = Path(vocabulary_path) if vocabulary_path is not None else find_vocabulary()
        self.sequence_length = sequence_length
        self.start_of_text_token_id = start_of_text_token_id
        self.end_of_text_token_id = end_of_text_token_id
        self.pad_token_id = pad_token_id
        self._tables: Optional[tuple[dict[int, str], dict[str, int], dict[tuple[str, str], int]]] = None
        self._cache: dict[str, list[str]] = {}

    # -- vocabulary: 256 byte symbols, the same 256 with an end-of-word mark, 48 894 merges, two specials ------------
    @staticmethod
    def byte_symbols() -> dict[int, str]:
        """byte -> vocabulary symbol, in vocabulary order: the 188 printable Latin-1 bytes first, then the other 68.
        NB (parity target = refiners, `tokenizer.py:69-79`): every byte maps to chr(byte); openai/CLIP maps the 68
        non-printable bytes to chr(256 + n) instead, which only matters for UTF-8 continuation bytes 0x80-0xA0 / 0xAD."""
        printable = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
        order = printable + [b for b in range(256) if b not in printable]
        return {b: chr(b) for b in order}

    def tables(self) -> tuple[dict[int, str], dict[str, int], dict[tuple[str, str], int]]:
        if self._tables is None:
            if self.vocabulary_path is None or not Path(self.vocabulary_path).is_file():
                raise FileNotFoundError("CLIP BPE vocabulary (bpe_simple_vocab_16e6.txt.gz) not found: pass vocabulary_path= or set REFINERS_AMD_CLIP_VOCAB")
            lines = gzip.open(self.vocabulary_path).read().decode("utf-8").split("\n")
            merges = [tuple(line.split()) for line in lines[1 : 49152 - 256 - 2 + 1]]
            sym = self.byte_symbols()
            vocab = list(sym.values()) + [s + "</w>" for s in sym.values()] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
            ids = {tok: i for i, tok in enumerate(vocab)}
            ranks = {m: i for i, m in enumerate(merges)}
            self._tables = (sym, ids, ranks)  # type: ignore[assignment]
        return self._tables  # type: ignore[return-value]

    def bpe(self, word: str) -> list[str]:
        """Greedy lowest-rank-first pair merging of one pre-token (already mapped to byte symbols)."""
        if word in self._cache:
            return self._cache[word]
        _, _, ranks = self.tables()
        parts = list(word[:-1]) + [word[-1] + "</w>"]
        while len(parts) > 1:
            best, best_rank = -1, None
            for i in range(len(parts) - 1):
                r = ranks.get((parts[i], parts[i + 1]))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = i, r
            if best_rank is None:
                break
            # One occurrence per round, leftmost first == openai/CLIP's "merge every occurrence left to right".  The
            # reference picks the occurrence with min() over a *set* of (position, pair) (`tokenizer.py:86-91`), so for
            # words that hold the best pair twice ("aaaa", "bbbbbb") its output depends on PYTHONHASHSEED; every other
            # word tokenizes identically (tests/test_clip_text.py).
            parts = parts[:best] + [parts[best] + parts[best + 1]] + parts[best + 2 :]
        self._cache[word] = parts
        return parts

    def encode(self, text: str, max_length: Optional[int] = None) -> Tensor:
        sym, ids, _ = self.tables()
        text = re.sub(r"\s+", " ", text.lower())
        out: list[int] = []
        limit = None if not max_length else max_length - 2
        for tok in self._pattern.findall(text):
            if tok in ("<|startoftext|>", "<|endoftext|>"):
                # the reference's vocabulary stores the two specials as empty strings, so a literal special in the text
                # goes through BPE as ordinary characters; do the same
                pass
            word = "".join(sym[b] for b in tok.encode("utf-8"))
            for piece in self.bpe(word):
                out.append(ids[piece])
        if limit is not None:
            out = out[:limit]
        return torch.tensor([self.start_of_text_token_id, *out, self.end_of_text_token_id])

    def tokenize_str(self, text: str) -> Tensor:
        t = self.encode(text, max_length=self.sequence_length).unsqueeze(0)
        assert t.shape[1] <= self.sequence_length
        return torch.nn.functional.pad(t, (0, self.sequence_length - t.shape[1]), value=self.pad_token_id)

    def forward(self, text: str | list[str]) -> Tensor:
        if isinstance(text, str):
            return self.tokenize_str(text)
        assert isinstance(text, list), f"Expected type `str` or `list[str]`, got {type(text)}"
        return torch.cat([self.tokenize_str(t) for t in text])


class TokenEncoder(Embedding):
    def __init__(self, vocabulary_size: int, embedding_dim: int, device: Any = None, dtype: Any = None) -> None:
        self.vocabulary_size = vocabulary_size
        self.embedding_dim = embedding_dim
        super().__init__(num_embeddings=vocabulary_size, embedding_dim=embedding_dim, device=device, dtype=dtype)


class PositionalEncoder(fl.Chain):
    """position ids 0..L-1 -> learned embedding (`clip/common.py:7-31`)."""

    def __init__(self, max_sequence_length: int, embedding_dim: int, device: Any = None, dtype: Any = None) -> None:
        self.max_sequence_length = max_sequence_length
        self.embedding_dim = embedding_dim
        super().__init__(fl.Lambda(func=self.get_position_ids), Embedding(max_sequence_length, embedding_dim, device=device, dtype=dtype))

    @property
    def position_ids(self) -> Tensor:
        return torch.arange(self.max_sequence_length, device=self.device).reshape(1, -1)

    def get_position_ids(self, x: Tensor) -> Tensor:
        return self.position_ids[:, : x.shape[1]]


class FeedForward(fl.Chain):
    def __init__(self, embedding_dim: int, feedforward_dim: int, device: Any = None, dtype: Any = None) -> None:
        self.embedding_dim = embedding_dim
        self.feedforward_dim = feedforward_dim
        super().__init__(
            fl.Linear(embedding_dim, feedforward_dim, device=device, dtype=dtype),
            fl.GeLU(),
            fl.Linear(feedforward_dim, embedding_dim, device=device, dtype=dtype),
        )


class TransformerLayer(fl.Chain):
    """x += causal SelfAttention(LN(x)); x += FeedForward(LN(x))   (`clip/text_encoder.py:25-69`)."""

    def __init__(self, embedding_dim: int, feedforward_dim: int, num_attention_heads: int = 1, layer_norm_eps: float = 1e-5,
                 device: Any = None, dtype: Any = None) -> None:
        self.embedding_dim = embedding_dim
        self.num_attention_heads = num_attention_heads
        self.feedforward_dim = feedforward_dim
        self.layer_norm_eps = layer_norm_eps
        super().__init__(
            fl.Residual(
                fl.LayerNorm(embedding_dim, eps=layer_norm_eps, device=device, dtype=dtype),
                fl.SelfAttention(embedding_dim=embedding_dim, num_heads=num_attention_heads, is_causal=True, device=device, dtype=dtype),
            ),
            fl.Residual(
                fl.LayerNorm(embedding_dim, eps=layer_norm_eps, device=device, dtype=dtype),
                FeedForward(embedding_dim=embedding_dim, feedforward_dim=feedforward_dim, device=device, dtype=dtype),
            ),
        )


class CLIPTextEncoder(fl.Chain):
    """tokenizer -> Converter -> token + position embeddings -> N causal transformer layers -> LayerNorm
    (`clip/text_encoder.py:72-148`)."""

    def __init__(self, embedding_dim: int = 768, max_sequence_length: int = 77, vocabulary_size: int = 49408, num_layers: int = 12,
                 num_attention_heads: int = 12, feedforward_dim: int = 3072, layer_norm_eps: float = 1e-5, use_quick_gelu: bool = False,
                 tokenizer: Optional[CLIPTokenizer] = None, device: Any = None, dtype: Any = None) -> None:
        self.embedding_dim = embedding_dim
        self.max_sequence_length = max_sequence_length
        self.vocabulary_size = vocabulary_size
        self.num_layers = num_layers
        self.num_attention_heads = num_attention_heads
        self.feedforward_dim = feedforward_dim
        self.layer_norm_eps = layer_norm_eps
        self.use_quick_gelu = use_quick_gelu
        super().__init__(
            tokenizer or CLIPTokenizer(sequence_length=max_sequence_length),
            fl.Converter(set_dtype=False),
            fl.Sum(
                TokenEncoder(vocabulary_size, embedding_dim, device=device, dtype=dtype),
                PositionalEncoder(max_sequence_length, embedding_dim, device=device, dtype=dtype),
            ),
            *(TransformerLayer(embedding_dim, feedforward_dim, num_attention_heads, layer_norm_eps, device=device, dtype=dtype) for _ in range(num_layers)),
            fl.LayerNorm(embedding_dim, eps=layer_norm_eps, device=device, dtype=dtype),
        )
        if use_quick_gelu:
            for gelu, parent in list(self.walk(predicate=lambda m, _: isinstance(m, fl.GeLU))):
                parent.replace(old_module=gelu, new_module=fl.GeLU(approximation=fl.GeLUApproximation.SIGMOID))


class CLIPTextEncoderL(CLIPTextEncoder):
    """768 wide, 12 layers, 12 heads, QuickGELU (`clip/text_encoder.py:151-186`)."""

    def __init__(self, device: Any = None, dtype: Any = None) -> None:
        super().__init__(embedding_dim=768, num_layers=12, num_attention_heads=12, feedforward_dim=3072, use_quick_gelu=True, device=device, dtype=dtype)


class CLIPTextEncoderH(CLIPTextEncoder):
    """1024 wide, 23 layers, 16 heads (`clip/text_encoder.py:189-217`)."""

    def __init__(self, device: Any = None, dtype: Any = None) -> None:
        super().__init__(embedding_dim=1024, num_layers=23, num_attention_heads=16, feedforward_dim=4096, device=device, dtype=dtype)


class CLIPTextEncoderG(CLIPTextEncoder):
    """1280 wide, 32 layers, 20 heads, pad token 0 (`clip/text_encoder.py:220-251`)."""

    def __init__(self, device: Any = None, dtype: Any = None) -> None:
        super().__init__(embedding_dim=1280, num_layers=32, num_attention_heads=20, feedforward_dim=5120, tokenizer=CLIPTokenizer(pad_token_id=0),
                         device=device, dtype=dtype)
