"""MI355X lowering of the prompt-side text encoders: `CLIPTextEncoder` (L / H / G,
`src/refiners/foundationals/clip/text_encoder.py:72-251`) and SDXL's `DoubleTextEncoder`
(`latent_diffusion/stable_diffusion_xl/text_encoder.py:14-101`), from token ids to the `clip_text_embedding` /
`pooled_text_embedding` tensors the UNet engine consumes -- in HBM, no host round trip.

Same engine as the UNet: the Chain tree (refiners_amd.clip / .latent_diffusion.prompt, or refiners' own classes) is
walked once into a launch program over [B*77, C] token rows.

  Sum(TokenEncoder, PositionalEncoder) -> row gather from the embedding table | + position rows
  TransformerLayer  -> LayerNorm | packed Q|K GEMM (+bias) | V^T GEMM | causal attention (mi355x_attention_general) |
                       out-proj GEMM + residual | LayerNorm | GEMM + (erf- or quick-)GELU epilogue | GEMM + residual
  final LayerNorm; pooled embedding = row gather at the end-of-text positions -> LayerNorm -> bias-free projection GEMM
  (the reference projects all 77 rows and then picks one; picking first is the same numbers with 77x less work)

V's bias never enters the attention kernel: softmax rows sum to one (also under the causal mask), so it passes through
unchanged and is folded into the output projection's bias (b_o' = b_o + W_o b_v, formed once in float32).
"""
from __future__ import annotations

from typing import Any, Optional, Sequence

import torch
from torch import Tensor

from .. import native
from ..fluxion.tree import tree_epoch
from .compiled import Program
from .lowering_blocks import BlockLowering as Lowering
from .packing import PackCache, _expect, cname, isa, kids, launches


class TextLowering(Lowering):
    # ---------------------------------------------------------------------------------------------------------------
    def embed(self, node: Any, tokens: Tensor, B: int, L: int) -> Tensor:
        ch = kids(node)
        _expect(isa(node, "Sum") and len(ch) == 2 and isa(ch[0], "TokenEncoder") and isa(ch[1], "PositionalEncoder"), "unexpected embedding Sum")
        table, pos_emb = ch[0], kids(ch[1])[1]
        _expect(isa(pos_emb, "Embedding") and pos_emb.weight.shape[0] >= L and pos_emb.weight.shape[1] == table.weight.shape[1], "unexpected PositionalEncoder")
        C = table.weight.shape[1]
        x = self.pool.get(B * L, C)
        native.gather_rows(self._w(table.weight), tokens, x)
        pos = self.cache.get(("clip_pos", B, L) + PackCache.ident(pos_emb.weight), lambda: self.cvt(pos_emb.weight)[:L].repeat(B, 1).contiguous())
        native.axpby(x, 1.0, pos, 1.0, x)
        return x

    def biased_self_attention(self, x: Tensor, B: int, L: int, ln: Any, att: Any) -> Tensor:
        """x += W_o SDPA(W_q h + b_q, W_k h + b_k, W_v h + b_v) + b_o, h = LN(x); causal for the text encoders
        (clip/text_encoder.py:41-54), bidirectional for the image encoders (clip/image_encoder.py:80-84);
        fluxion/layers/attentions.py:319-385."""
        (qn, kn, vn), sd, on, ip = self._split_attention(att, allow_causal=True)
        _expect(ip is None, "expected a plain SelfAttention")
        heads = sd.num_heads
        M, C = x.shape
        d = C // heads
        _expect(self.head_kernel(d) is not None, f"head dim {d} has no attention kernel")
        qs, ks, vs, os_ = (self.linear_spec(n) for n in (qn, kn, vn, on))
        _expect(all(s.lora is None for s in (qs, ks, vs, os_)), "un-merged LoRAs on a text encoder attention (use lora_mode='merged')")
        h = self.layernorm(x, ln)
        ident = PackCache.ident(qs.w, ks.w, qs.b, ks.b)
        wqk = self.cache.get(("clip_wqk",) + ident, lambda: torch.cat([qs.w, ks.w], 0).contiguous())
        bqk = None
        if qs.b is not None or ks.b is not None:
            z = lambda s: s.b if s.b is not None else torch.zeros(C, device=self.device, dtype=self.dtype)  # noqa: E731
            bqk = self.cache.get(("clip_bqk",) + ident, lambda: torch.cat([z(qs), z(ks)]).contiguous())
        qk = self.pool.get(M, 2 * C)
        native.gemm([(h, self.kblocked(wqk))], qk, bias=bqk)
        lkp = self._pad_keys(L)
        if lkp == L:
            vt = self.pool.get(C, M)
            native.gemm([(self.kblocked(vs.w), h)], vt, weight_operand="x")
        else:  # every sample's V^T columns start on a 64-key boundary; the padding is zeroed once, here
            vt = torch.zeros(C, B * lkp, device=self.device, dtype=self.dtype)
            self.__dict__.setdefault("_keep", []).append(vt)
            for b in range(B):
                native.gemm([(self.kblocked(vs.w), h[b * L : (b + 1) * L])], vt[:, b * lkp : b * lkp + L], weight_operand="x")
        self.pool.put(h)
        o = self.pool.get(M, C)
        q3 = qk.as_strided((B, L, C), (L * qk.stride(0), qk.stride(0), 1))
        k3 = qk[:, C:].as_strided((B, L, C), (L * qk.stride(0), qk.stride(0), 1))
        native.attention_general(q3, k3, vt.view(C, B, lkp), o.view(B, L, C), heads, L, causal=bool(sd.is_causal))
        self.pool.put(qk)
        if lkp == L:
            self.pool.put(vt)
        bo = os_.b
        if vs.b is not None:  # V's bias rides through softmax (rows sum to 1) into the output projection's bias

            def fold() -> Tensor:
                if self.device.type == "meta":
                    return torch.empty(C, device=self.device, dtype=self.dtype)
                acc = self._mm(os_.w.float(), vs.b.float().unsqueeze(0)).reshape(-1)
                return (acc + os_.b.float() if os_.b is not None else acc).to(self.dtype).contiguous()

            bo = self.cache.get(("clip_bo",) + PackCache.ident(os_.w, os_.b, vs.b), fold)
        native.gemm([(o, self.kblocked(os_.w))], x, bias=bo, res=x)
        self.pool.put(o)
        return x

    def feed_forward_gelu(self, x: Tensor, ln: Any, ff: Any) -> Tensor:
        ch = kids(ff)
        _expect(len(ch) == 3 and isa(ch[1], "GeLU") and ch[1].approximation.value in ("none", "sigmoid"), f"unexpected FeedForward layout in {cname(ff)}")
        h = self.layernorm(x, ln)
        f = self.linear(h, self.linear_spec(ch[0]), gelu="quick" if ch[1].approximation.value == "sigmoid" else True)
        self.pool.put(h)
        self.linear(f, self.linear_spec(ch[2]), res=x, out=x)
        self.pool.put(f)
        return x

    def transformer_layer(self, layer: Any, x: Tensor, B: int, L: int) -> Tensor:
        ch = kids(layer)
        _expect(len(ch) == 2 and all(isa(c, "Residual") for c in ch), "unexpected TransformerLayer layout")
        r1, r2 = kids(ch[0]), kids(ch[1])
        _expect(len(r1) == 2 and isa(r1[0], "LayerNorm") and isa(r1[1], "SelfAttention") and len(r2) == 2 and isa(r2[0], "LayerNorm"), "unexpected TransformerLayer residuals")
        x = self.biased_self_attention(x, B, L, r1[0], r1[1])
        return self.feed_forward_gelu(x, r2[0], r2[1])

    def encoder_body(self, nodes: Sequence[Any], x: Optional[Tensor], tokens: Tensor, B: int, L: int) -> Tensor:
        """Runs the children of a (possibly sliced) CLIPTextEncoder chain; tokenizers / Converter / SetContext are host-side."""
        for n in nodes:
            if isa(n, "CLIPTokenizer") or isa(n, "Converter") or isa(n, "SetContext"):
                continue
            if isa(n, "Sum"):
                x = self.embed(n, tokens, B, L)
            elif isa(n, "TransformerLayer"):
                assert x is not None
                x = self.transformer_layer(n, x, B, L)
            elif isa(n, "LayerNorm"):
                assert x is not None
                y = self.layernorm(x, n)
                self.pool.put(x)
                x = y
            else:
                _expect(False, f"unexpected {cname(n)} in a CLIP text encoder")
        assert x is not None
        return x

    # ---------------------------------------------------------------------------------------------------------------
    def lower_encoder(self, enc: Any, tokens: Tensor, B: int, L: int, out: Tensor) -> None:
        """A whole CLIPTextEncoder (e.g. SD1.5's CLIP-L): out [B*L, C] = final LayerNorm output."""
        with self.in_step():
            x = self.encoder_body(kids(enc), None, tokens, B, L)
            native.axpby(x, 1.0, x, 0.0, out)
            self.pool.put(x)

    def lower_double(self, enc: Any, tokens_l: Tensor, tokens_g: Tensor, eot_rows: Tensor, B: int, L: int, emb: Tensor, pooled: Tensor) -> None:
        ch = kids(enc)
        _expect(len(ch) == 2 and isa(ch[0], "Parallel") and isa(ch[1], "Lambda"), "unexpected DoubleTextEncoder layout")
        enc_l, tewp = kids(ch[0])
        _expect(isa(enc_l, "CLIPTextEncoder") and isa(tewp, "TextEncoderWithPooling"), "unexpected DoubleTextEncoder branches")
        tc = kids(tewp)
        _expect(len(tc) == 4 and isa(tc[2], "CLIPTextEncoder") and isa(tc[3], "Parallel"), "unexpected TextEncoderWithPooling layout")
        pc = kids(tc[3])
        _expect(len(pc) == 2 and isa(pc[0], "Identity") and isa(pc[1], "Chain"), "unexpected pooling branch")
        tail = kids(pc[1])
        _expect(len(tail) == 3 and isa(tail[0], "CLIPTextEncoder") and isa(tail[1], "Linear") and isa(tail[2], "Lambda"), "unexpected pooling chain")
        last = kids(tail[0])
        _expect(len(last) == 2 and isa(last[0], "TransformerLayer") and isa(last[1], "LayerNorm"), "unexpected CLIP-G tail")
        proj = self.linear_spec(tail[1])
        _expect(proj.b is None and proj.lora is None, "text projection with bias / LoRA")
        with self.in_step():
            xl = self.encoder_body(kids(enc_l), None, tokens_l, B, L)
            xg = self.encoder_body(kids(tc[2]), None, tokens_g, B, L)
            native.concat2(xl, xg, emb)
            self.pool.put(xl)
            # pooled branch: the last layer mutates its input in place, xg has been copied into `emb` already
            z = self.transformer_layer(last[0], xg, B, L)
            rows = self.pool.get(B, z.shape[1])
            native.gather_rows(z, eot_rows, rows)
            self.pool.put(z)
            rn = self.layernorm(rows, last[1])
            self.pool.put(rows)
            native.gemm([(rn, proj.w)], pooled)
            self.pool.put(rn)


def _as_tokens(t: Tensor, device: torch.device) -> Tensor:
    return t.to(device=device, dtype=torch.int32).contiguous()


class CompiledDoubleTextEncoder:
    """`fast = CompiledDoubleTextEncoder(double_text_encoder); emb, pooled = fast(prompts)` == `double_text_encoder(prompts)`
    ((B, 77, 2048), (B, 1280)); `fast(tokens=(tokens_l, tokens_g))` skips the host tokenizers."""

    def __init__(self, enc: Any, lora_mode: str = "merged", use_graph: bool = True) -> None:
        native.load()
        self.enc = enc
        self.lora_mode = lora_mode
        self.use_graph = use_graph
        self.cache = PackCache()
        self.key: Any = None
        self.stats: dict[str, Any] = {}

    def tokenizers(self) -> tuple[Any, Any]:
        toks = [m for m in self.enc.modules() if isa(m, "CLIPTokenizer")]
        _expect(len(toks) >= 2, "expected the CLIP-L and CLIP-G tokenizers in the tree")
        enc_l = kids(kids(self.enc)[0])[0]
        tl = next(m for m in enc_l.modules() if isa(m, "CLIPTokenizer"))
        tg = next(m for m in toks if m is not tl)
        return tl, tg

    @torch.no_grad()
    def __call__(self, text: Any = None, *, tokens: Optional[tuple[Tensor, Tensor]] = None) -> tuple[Tensor, Tensor]:
        tl, tg = self.tokenizers()
        if tokens is None:
            tokens = (tl(text), tg(text))
        dev, dtype = self.enc.device, self.enc.dtype
        tok_l, tok_g = _as_tokens(tokens[0], dev), _as_tokens(tokens[1], dev)
        B, L = tok_l.shape
        key = (tree_epoch(), B, L, dtype, dev)
        if key != self.key:
            self.tok_l = torch.empty(B * L, device=dev, dtype=torch.int32)
            self.tok_g = torch.empty(B * L, device=dev, dtype=torch.int32)
            self.eot = torch.empty(B, device=dev, dtype=torch.int32)
            cl = next(p for n, p in self.enc.named_parameters() if "CLIPTextEncoderL" in n and p.dim() == 2).shape[1]
            cg = next(p for n, p in self.enc.named_parameters() if "CLIPTextEncoderG" in n and p.dim() == 2).shape[1]
            self.emb = torch.empty(B * L, cl + cg, device=dev, dtype=dtype)
            self.pooled = torch.empty(B, cg, device=dev, dtype=dtype)
            low = TextLowering(dev, dtype, self.cache, self.lora_mode)
            low.lower_double(self.enc, self.tok_l, self.tok_g, self.eot, B, L, self.emb, self.pooled)
            self.cache.sweep()
            self.low, self.key, self.program = low, key, Program(low.step, self.use_graph, low=low)
            self.stats = dict(low.stats, step_ops=launches(low.step), pool_bytes=low.step_pool.bytes())
        self.tok_l.copy_(tok_l.reshape(-1))
        self.tok_g.copy_(tok_g.reshape(-1))
        # first end-of-text position per prompt (TextEncoderWithPooling.set_end_of_text_index, xl/text_encoder.py:48-51)
        first = (tok_g == tg.end_of_text_token_id).to(torch.int32).argmax(dim=1).to(torch.int32)
        self.eot.copy_(first + torch.arange(B, device=dev, dtype=torch.int32) * L)
        self.program.run()
        return self.emb.view(B, L, -1).clone(), self.pooled.clone()


class CompiledTextEncoder:
    """`fast = CompiledTextEncoder(clip_text_encoder); hidden = fast(prompts)` == `clip_text_encoder(prompts)` (B, 77, C)."""

    def __init__(self, enc: Any, lora_mode: str = "merged", use_graph: bool = True) -> None:
        native.load()
        self.enc = enc
        self.lora_mode = lora_mode
        self.use_graph = use_graph
        self.cache = PackCache()
        self.key: Any = None
        self.stats: dict[str, Any] = {}

    @torch.no_grad()
    def __call__(self, text: Any = None, *, tokens: Optional[Tensor] = None) -> Tensor:
        if tokens is None:
            tokens = next(m for m in self.enc.modules() if isa(m, "CLIPTokenizer"))(text)
        dev, dtype = self.enc.device, self.enc.dtype
        tok = _as_tokens(tokens, dev)
        B, L = tok.shape
        key = (tree_epoch(), B, L, dtype, dev)
        if key != self.key:
            self.tok = torch.empty(B * L, device=dev, dtype=torch.int32)
            self.out = torch.empty(B * L, self.enc.embedding_dim, device=dev, dtype=dtype)
            low = TextLowering(dev, dtype, self.cache, self.lora_mode)
            low.lower_encoder(self.enc, self.tok, B, L, self.out)
            self.cache.sweep()
            self.low, self.key, self.program = low, key, Program(low.step, self.use_graph, low=low)
            self.stats = dict(low.stats, step_ops=launches(low.step), pool_bytes=low.step_pool.bytes())
        self.tok.copy_(tok.reshape(-1))
        self.program.run()
        return self.out.view(B, L, -1).clone()
