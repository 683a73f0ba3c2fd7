"""MI355X lowering of the IP-Adapter image-prompt encoder: `CLIPImageEncoder`
(`src/refiners/foundationals/clip/image_encoder.py:134-239`) and `ImageProjection`
(`latent_diffusion/image_prompt.py:24-45`), from a preprocessed (B, 3, 224, 224) image to the (2B, 4, 2048)
`clip_image_embedding` [negative ; conditional] that `ImageCrossAttention` reads -- in HBM, ready for the UNet engine.

  ViTEmbeddings    -> patchify kernel | GEMM (positions ride in as the residual operand) | class row (token + position 0)
  LayerNorm, 32 x TransformerLayer (bidirectional attention over 257 tokens, 16 heads of 80: mi355x_attention_general)
  class token      -> row gather | LayerNorm | bias-free projection GEMM
  ImageProjection  -> GEMM (+bias) over [zeros ; embedding] | LayerNorm over the (4, 2048) token rows

Fine-grained ("plus") adapters (`image_prompt.py:81-234, 516-525, 553-564`): the encoder stops after its 31st layer (the
penultimate token grid, 257 x 1280) and a PerceiverResampler turns [grid(zero image) ; grid(image)] into 16 tokens each:
  per layer: LayerNorm x2 | per sample one GEMM over [LN1(x) ; LN2(latents)] writing K row-major and V transposed |
             Wq GEMM | flash attention (16 queries, 273 keys, 20 heads of 64) | Wo GEMM (+residual) | LN | GEMM+GELU | GEMM (+residual)
"""
from __future__ import annotations

from typing import Any, Optional

import torch
from torch import Tensor

from .. import native
from ..fluxion.tree import tree_epoch
from .compiled import Program
from .packing import PackCache, _expect, isa, kids, launches
from .text import TextLowering


class ImagePromptLowering(TextLowering):
    def vit_embeddings(self, node: Any, image: Tensor) -> tuple[Tensor, int]:
        ch = kids(node)
        _expect(len(ch) == 2 and isa(ch[0], "Concatenate") and ch[0].dim == 1 and isa(ch[1], "Residual"), "unexpected ViTEmbeddings layout")
        cat = kids(ch[0])
        _expect(len(cat) == 2 and isa(cat[0], "ClassToken") and isa(cat[1], "Chain"), "unexpected ViTEmbeddings Concatenate")
        pe = kids(cat[1])[0]
        conv = kids(pe)[0]
        _expect(isa(pe, "PatchEncoder") and isa(conv, "Conv2d") and conv.bias is None and conv.kernel_size == tuple(conv.stride), "unexpected PatchEncoder")
        pos = kids(kids(ch[1])[0])[1]
        cls = kids(cat[0])[0]
        B, Ci, H, W = image.shape
        P = conv.kernel_size[0]
        n = (H // P) * (W // P)
        C = conv.out_channels
        L = n + 1
        _expect(isa(pos, "Embedding") and tuple(pos.weight.shape) == (L, C) and tuple(cls.weight.shape) == (1, C), "unexpected class token / positions")
        K = Ci * P * P
        Kp = (K + self.kblk - 1) // self.kblk * self.kblk  # 588 -> 640: zero columns, zero weight columns

        def pack_w() -> Tensor:
            w = torch.zeros(C, Kp, device=self.device, dtype=self.dtype)
            if self.device.type != "meta":
                w[:, :K] = conv.weight.detach().reshape(C, K).to(self.device, self.dtype)
            return w

        w = self.cache.get(("vit_patch_w", Kp) + PackCache.ident(conv.weight), pack_w)
        pos_patch = self.cache.get(("vit_pos_patch",) + PackCache.ident(pos.weight), lambda: self.cvt(pos.weight)[1:].contiguous())

        def cls_row() -> Tensor:
            if self.device.type == "meta":
                return torch.empty(1, C, device=self.device, dtype=self.dtype)
            return (cls.weight.detach().float() + pos.weight.detach().float()[:1]).to(self.device, self.dtype).contiguous()

        c0 = self.cache.get(("vit_cls_row",) + PackCache.ident(cls.weight, pos.weight), cls_row)
        cols = torch.zeros(B * n, Kp, device=self.device, dtype=self.dtype)  # pad columns stay zero: patchify writes the first K only
        self.__dict__.setdefault("_keep", []).append(cols)
        native.patchify_nchw(image, P, cols)
        x = self.pool.get(B * L, C)
        for b in range(B):
            native.gemm([(cols[b * n : (b + 1) * n], w)], x[b * L + 1 : (b + 1) * L], res=pos_patch)
            native.axpby(c0, 1.0, c0, 0.0, x[b * L : b * L + 1])
        return x, L

    def lower_image_encoder(self, enc: Any, image: Tensor, cls_rows: Tensor, out: Tensor) -> None:
        """out [B, output_dim] = enc(image); cls_rows: int32 [B] = b * L (filled here)."""
        ch = kids(enc)
        _expect(len(ch) == 6 and isa(ch[0], "ViTEmbeddings") and isa(ch[1], "LayerNorm") and isa(ch[2], "Chain") and isa(ch[3], "Lambda")
                and isa(ch[4], "LayerNorm") and isa(ch[5], "Linear"), "unexpected CLIPImageEncoder layout")
        B = image.shape[0]
        proj = self.linear_spec(ch[5])
        _expect(proj.b is None and proj.lora is None, "image projection with bias / LoRA")
        with self.in_step():
            x, L = self.vit_embeddings(ch[0], image)
            if self.device.type != "meta":
                cls_rows.copy_(torch.arange(B, device=self.device, dtype=torch.int32) * L)
            y = self.layernorm(x, ch[1])
            self.pool.put(x)
            x = y
            for layer in kids(ch[2]):
                x = self.transformer_layer(layer, x, B, L)
            rows = self.pool.get(B, x.shape[1])
            native.gather_rows(x, cls_rows, rows)  # Lambda(cls_token_pooling): x[:, 0, :]
            self.pool.put(x)
            rn = self.layernorm(rows, ch[4])
            self.pool.put(rows)
            native.gemm([(rn, proj.w)], out)
            self.pool.put(rn)

    def lower_grid_encoder(self, enc: Any, image: Tensor) -> tuple[Tensor, int]:
        """x [B * 257, 1280] = the grid-feature encoder (convert_to_grid_features: ViTEmbeddings, LayerNorm, 31 layers)."""
        ch = kids(enc)
        _expect(len(ch) == 3 and isa(ch[0], "ViTEmbeddings") and isa(ch[1], "LayerNorm") and isa(ch[2], "Chain"), "unexpected grid-feature encoder layout")
        B = image.shape[0]
        with self.in_step():
            x, L = self.vit_embeddings(ch[0], image)
            y = self.layernorm(x, ch[1])
            self.pool.put(x)
            x = y
            for layer in kids(ch[2]):
                x = self.transformer_layer(layer, x, B, L)
        return x, L

    def lower_perceiver(self, proj: Any, feats: Tensor, B: int, L: int, tokens: Tensor) -> None:
        """tokens [B * num_tokens, output_dim] = PerceiverResampler(feats [B * L, input_dim])   (image_prompt.py:178-234)."""
        ch = kids(proj)
        _expect(len(ch) == 6 and isa(ch[0], "Linear") and isa(ch[1], "SetContext") and isa(ch[2], "LatentsToken") and isa(ch[3], "Transformer") and isa(ch[4], "Linear")
                and isa(ch[5], "LayerNorm"), "unexpected PerceiverResampler layout")
        T, D = proj.num_tokens, proj.latents_dim
        heads, d = proj.num_attention_heads, proj.head_dim
        inner = heads * d
        _expect(d == 64 and inner % 128 == 0, "the resampler's attention runs on the 64-wide flash kernel")
        Lk = L + T
        Lkp = (Lk + 63) // 64 * 64
        param = kids(ch[2])[0]
        _expect(isa(param, "Parameter") and tuple(param.weight.shape) == (T, D), "unexpected LatentsToken")
        rep = self.cache.get(("latents_rep", B) + PackCache.ident(param.weight), lambda: self.cvt(param.weight).repeat(B, 1).contiguous())
        keep = self.__dict__.setdefault("_keep", [])
        kbuf = torch.zeros(B * Lkp, inner, device=self.device, dtype=self.dtype)  # key rows per sample, padded to 64 keys (padding masked by Lk)
        vt = torch.zeros(inner, B * Lkp, device=self.device, dtype=self.dtype)    # V^T; the padding stays an exact 0
        kvin = torch.empty(B * Lk, D, device=self.device, dtype=self.dtype)
        keep += [kbuf, vt, kvin]
        with self.in_step():
            xp = self.linear(feats, self.linear_spec(ch[0]))
            lat = self.pool.get(B * T, D)
            native.axpby(rep, 1.0, rep, 0.0, lat)  # the learned latent queries, one copy per sample
            for layer in kids(ch[3]):
                r1, r2 = (kids(c) for c in kids(layer))
                _expect(len(r1) == 2 and isa(r1[0], "Parallel") and isa(r1[1], "PerceiverAttention") and len(r2) == 2 and isa(r2[0], "LayerNorm"), "unexpected resampler layer")
                att = kids(r1[1])
                _expect(len(att) == 4 and isa(att[0], "Distribute") and isa(att[1], "Parallel") and isa(att[2], "PerceiverScaledDotProductAttention") and isa(att[3], "Linear"),
                        "unexpected PerceiverAttention layout")
                ln_x, ln_l = kids(att[0])
                wkv, wq = self.linear_spec(kids(kids(att[1])[0])[1]), self.linear_spec(kids(kids(att[1])[1])[1])
                _expect(wkv.b is None and wq.b is None and wkv.N == 2 * inner and wq.N == inner and att[2].num_heads == heads, "unexpected resampler projections")
                xn, ll = self.layernorm(xp, ln_x), self.layernorm(lat, ln_l)
                for b in range(B):  # key / value source of sample b: [LN1(x_b) ; LN2(latents_b)]
                    native.axpby(xn[b * L : (b + 1) * L], 1.0, xn[b * L : (b + 1) * L], 0.0, kvin[b * Lk : b * Lk + L])
                    native.axpby(ll[b * T : (b + 1) * T], 1.0, ll[b * T : (b + 1) * T], 0.0, kvin[b * Lk + L : (b + 1) * Lk])
                    native.gemm([(kvin[b * Lk : (b + 1) * Lk], self.kblocked(wkv.w))], kbuf[b * Lkp : b * Lkp + Lk], out_t=vt[:, b * Lkp :], nt_begin=inner)
                q = self.linear(ll, wq)
                self.pool.put(xn)
                self.pool.put(ll)
                o = self.pool.get(B * T, inner)
                native.attention(q.view(B, T, inner), o.view(B, T, inner), heads, [(kbuf.view(B, Lkp, inner), vt.view(inner, B, Lkp), Lk, 1.0)])
                self.pool.put(q)
                self.linear(o, self.linear_spec(att[3]), res=lat, out=lat)
                self.pool.put(o)
                ff = kids(r2[1])
                _expect(len(ff) == 3 and isa(ff[0], "Linear") and isa(ff[1], "GeLU") and isa(ff[2], "Linear") and ff[1].approximation.value == "none", "unexpected resampler FeedForward")
                h = self.layernorm(lat, r2[0])
                f1 = self.linear(h, self.linear_spec(ff[0]), gelu=True)
                self.pool.put(h)
                self.linear(f1, self.linear_spec(ff[2]), res=lat, out=lat)
                self.pool.put(f1)
            y = self.linear(lat, self.linear_spec(ch[4]))
            self.pool.put(lat)
            self.pool.put(xp)
            native.layernorm(y, self._w(ch[5].weight), self._w(ch[5].bias), ch[5].eps, tokens)
            self.pool.put(y)

    def lower_image_projection(self, proj: Any, both: Tensor, tokens: Tensor) -> None:
        """tokens [2B * num_tokens, C_text] = LayerNorm(reshape(Linear([zeros ; embedding])))."""
        ch = kids(proj)
        _expect(len(ch) == 3 and isa(ch[0], "Linear") and isa(ch[1], "Reshape") and isa(ch[2], "LayerNorm"), "unexpected ImageProjection layout")
        spec = self.linear_spec(ch[0])
        nt = proj.num_tokens
        with self.in_step():
            t = self.linear(both, spec)
            native.layernorm(t.view(both.shape[0] * nt, spec.N // nt), self._w(ch[2].weight), self._w(ch[2].bias), ch[2].eps, tokens)
            self.pool.put(t)


class CompiledImagePrompt:
    """`fast = CompiledImagePrompt(clip_image_encoder, image_proj); tokens = fast(image)` ==
    `cat(image_proj(zeros_like(e)), image_proj(e))` with `e = clip_image_encoder(image)` -- what
    `IPAdapter.compute_clip_image_embedding` hands to `set_clip_image_embedding` for one image per prompt.
    Without `image_proj` it returns `e` itself."""

    def __init__(self, encoder: Any, image_proj: Optional[Any] = None, lora_mode: str = "merged", use_graph: bool = True) -> None:
        native.load()
        self.encoder, self.image_proj = encoder, image_proj
        self.lora_mode, self.use_graph = lora_mode, use_graph
        self.cache = PackCache()
        self.key: Any = None
        self.stats: dict[str, Any] = {}

    @torch.no_grad()
    def __call__(self, image: Tensor) -> Tensor:
        dev, dtype = self.encoder.device, self.encoder.dtype
        B = image.shape[0]
        key = (tree_epoch(), tuple(image.shape), dtype, dev)
        if key != self.key:
            self.x = torch.empty(tuple(image.shape), device=dev, dtype=dtype)
            self.cls_rows = torch.zeros(B, device=dev, dtype=torch.int32)
            od = self.encoder.output_dim
            self.both = torch.zeros(2 * B, od, device=dev, dtype=dtype)  # rows [0, B) stay zero: the negative prompt
            low = ImagePromptLowering(dev, dtype, self.cache, self.lora_mode)
            low.lower_image_encoder(self.encoder, self.x, self.cls_rows, self.both[B:])
            self.tokens = None
            if self.image_proj is not None:
                nt, ct = self.image_proj.num_tokens, self.image_proj.clip_text_embedding_dim
                self.tokens = torch.empty(2 * B * nt, ct, device=dev, dtype=dtype)
                low.lower_image_projection(self.image_proj, self.both, self.tokens)
            self.cache.sweep()
            self.low, self.key, self.program = low, key, Program(low.step, self.use_graph, low=low)
            self.stats = dict(low.stats, step_ops=launches(low.step), pool_bytes=low.step_pool.bytes())
        self.x.copy_(image)
        self.program.run()
        if self.tokens is None:
            return self.both[B:].clone()
        return self.tokens.view(2 * B, self.image_proj.num_tokens, -1).clone()


class CompiledImagePromptPlus:
    """Fine-grained ("plus") IP-Adapter image prompt: `fast = CompiledImagePromptPlus(grid_image_encoder, perceiver_resampler);
    tokens = fast(image)` == `cat(resampler(enc(zeros_like(image))), resampler(enc(image)))` -- what
    `IPAdapter.compute_clip_image_embedding` returns for a fine-grained adapter (image_prompt.py:516-525): (2B, 16, 2048)
    [negative ; conditional] tokens for `set_clip_image_embedding`, in HBM."""

    def __init__(self, grid_encoder: Any, resampler: Any, lora_mode: str = "merged", use_graph: bool = True) -> None:
        native.load()
        self.encoder, self.resampler = grid_encoder, resampler
        self.lora_mode, self.use_graph = lora_mode, use_graph
        self.cache = PackCache()
        self.key: Any = None
        self.stats: dict[str, Any] = {}

    @torch.no_grad()
    def __call__(self, image: Tensor) -> Tensor:
        dev, dtype = self.encoder.device, self.encoder.dtype
        B = image.shape[0]
        key = (tree_epoch(), tuple(image.shape), dtype, dev)
        if key != self.key:
            self.x = torch.zeros((2 * B,) + tuple(image.shape[1:]), device=dev, dtype=dtype)  # rows [0, B) stay zero: the negative prompt's image
            low = ImagePromptLowering(dev, dtype, self.cache, self.lora_mode)
            feats, L = low.lower_grid_encoder(self.encoder, self.x)
            nt, od = self.resampler.num_tokens, self.resampler.output_dim
            self.tokens = torch.empty(2 * B * nt, od, device=dev, dtype=dtype)
            low.lower_perceiver(self.resampler, feats, 2 * B, L, self.tokens)
            self.cache.sweep()
            self.low, self.key, self.program = low, key, Program(low.step, self.use_graph, low=low)
            self.stats = dict(low.stats, step_ops=launches(low.step), pool_bytes=low.step_pool.bytes())
        self.x[B:].copy_(image)
        self.program.run()
        return self.tokens.view(2 * B, self.resampler.num_tokens, -1).clone()
