"""MI355X lowering of the SegmentAnything ViT image encoder (BASELINE.json config 5).

Same engine as the UNet: the Chain tree (refiners_amd.segment_anything, or refiners' own classes) is walked once and
turned into a launch program; tokens stay channels-last [B*64*64, C] end to end.

  PatchEncoder        -> patchify kernel + ONE GEMM (bias and the PositionalEncoder table ride in its epilogue)
  TransformerLayer    -> LayerNorm | row gather (WindowPartition as a static index table, zero rows for the 64 -> 70 padding)
                         | QKV GEMM | attention | row gather (WindowMerge, done BEFORE the output projection so the padded
                         rows are never projected) | out-proj GEMM + residual | LayerNorm | GEMM + erf-GELU epilogue |
                         GEMM + residual
  Neck                -> GEMM (1x1 conv) | LayerNorm (LayerNorm2d == per-token LN in this layout) | implicit-GEMM 3x3 | LayerNorm
  SAMViTAdapter hook  -> one copy of the token buffer after the first global-attention layer

NOT native yet: the attention itself.  ViT-H has 16 heads of 80 and an additive decomposed relative-position bias
(image_encoder.py:82-127); the flash kernel of csrc/attention.hip covers head_dim 64 without bias, so each of the 32
attentions runs the node's own torch forward on the GPU (listed in stats["fallback_nodes"]).  Everything else (5.45 of
the encoder's 5.96 TFLOP) is on the hand-written kernels.
"""
from __future__ import annotations

from typing import Any, Optional

import torch
from torch import Tensor

from .. import native
from ..fluxion.tree import tree_epoch
from .lowering import Act, Lowering, PackCache, Unsupported, _expect, cname, isa, kids


class SAMLowering(Lowering):
    def lower(self, vit: Any, image: Tensor, out: Tensor, early: Optional[Tensor]) -> None:
        ch = kids(vit)
        _expect(len(ch) == 4 and isa(ch[0], "PatchEncoder") and isa(ch[1], "PositionalEncoder") and isa(ch[2], "Transformer") and isa(ch[3], "Neck"), "unexpected SAMViT layout")
        B, C3, H, W = image.shape
        with self.in_step():
            tok, gh, gw = self.patch_encoder(ch[0], ch[1], image)
            for layer in kids(ch[2]):
                tok = self.transformer_layer(layer, tok, B, gh, gw, early)
            self.neck(ch[3], tok, B, gh, gw, out)

    # ---------------------------------------------------------------------------------------------------------------
    def patch_encoder(self, pe: Any, pos: Any, image: Tensor) -> tuple[Tensor, int, int]:
        conv = kids(pe)[0]
        _expect(isa(conv, "Conv2d") and conv.kernel_size == tuple(conv.stride) and conv.kernel_size[0] == conv.kernel_size[1] and isa(kids(pe)[1], "Permute"), "unexpected PatchEncoder")
        P = conv.kernel_size[0]
        B, C, H, W = image.shape
        gh, gw = H // P, W // P
        K = C * P * P
        _expect(K % self.kblk == 0, "patch size not aligned to the GEMM K block")
        table = kids(pos)[0]
        _expect(isa(pos, "Residual") and isa(table, "Parameter") and tuple(table.weight.shape) == (gh, gw, conv.out_channels), "unexpected PositionalEncoder")
        w = self.cache.get(("patch_w",) + PackCache.ident(conv.weight), lambda: self.cvt(conv.weight.detach().reshape(conv.out_channels, K)))
        posr = self.cache.get(("pos", B) + PackCache.ident(table.weight), lambda: self.cvt(table.weight.detach().reshape(gh * gw, -1)).repeat(B, 1).contiguous())
        cols = self.pool.get(B * gh * gw, K)
        native.patchify_nchw(image, P, cols)
        tok = self.pool.get(B * gh * gw, conv.out_channels)
        native.gemm([(cols, w)], tok, bias=self._w(conv.bias), res=posr)
        self.pool.put(cols)
        return tok, gh, gw

    def _window_tables(self, B: int, H: int, W: int, ws: int) -> tuple[Tensor, Tensor, int]:
        def make() -> tuple[Tensor, Tensor, int]:
            nwy, nwx = (H + ws - 1) // ws, (W + ws - 1) // ws
            b, wy, wx, ty, tx = torch.meshgrid(torch.arange(B), torch.arange(nwy), torch.arange(nwx), torch.arange(ws), torch.arange(ws), indexing="ij")
            y, x = wy * ws + ty, wx * ws + tx
            part = torch.where((y < H) & (x < W), b * H * W + y * W + x, torch.full_like(y, -1)).reshape(-1)
            gb, gy, gx = torch.meshgrid(torch.arange(B), torch.arange(H), torch.arange(W), indexing="ij")
            merge = (((gb * nwy + gy // ws) * nwx + gx // ws) * ws + gy % ws) * ws + gx % ws
            return part.to(torch.int32).to(self.device), merge.reshape(-1).to(torch.int32).to(self.device), B * nwy * nwx

        return self.cache.get(("windows", B, H, W, ws), make)

    def transformer_layer(self, layer: Any, tok: Tensor, B: int, gh: int, gw: int, early: Optional[Tensor]) -> Tensor:
        ch = kids(layer)
        _expect(len(ch) >= 2 and isa(ch[0], "Residual") and isa(ch[1], "Residual"), "unexpected TransformerLayer layout")
        r1, r2 = kids(ch[0]), kids(ch[1])
        _expect(len(r1) == 4 and isa(r1[0], "LayerNorm") and isa(r1[2], "FusedSelfAttention"), "unexpected attention residual")
        fsa = kids(r1[2])
        _expect(len(fsa) == 3 and isa(fsa[1], "RelativePositionAttention"), "unexpected FusedSelfAttention layout")
        windowed = isa(r1[1], "WindowPartition")
        _expect(windowed == isa(r1[3], "WindowMerge") and (windowed or isa(r1[1], "Identity")), "inconsistent window partition / merge")
        M, C = tok.shape
        h = self.layernorm(tok, r1[0])
        if windowed:
            ws = layer.window_size
            part, merge, nwin = self._window_tables(B, gh, gw, ws)
            hw = self.pool.get(nwin * ws * ws, C)
            native.gather_rows(h, part, hw)
            self.pool.put(h)
            h, shape = hw, (nwin, ws, ws)
        else:
            shape = (B, gh, gw)
        qkv = self.linear(h, self.linear_spec(fsa[0]))
        self.pool.put(h)
        att = self.pool.get(qkv.shape[0], C)
        node = fsa[1]

        def attend(att: Tensor = att, qkv: Tensor = qkv, shape: tuple = shape, node: Any = node) -> None:
            # the node's own torch forward (head_dim 80 + relative position bias: not covered by the flash kernel);
            # arguments are bound NOW: `att` is rebound to the merged buffer a few lines below
            att.view(*shape, C).copy_(node(qkv.view(*shape, 3 * C)))

        self.python(attend, "torch:RelativePositionAttention")
        self.stats["fallback_nodes"].append(f"RelativePositionAttention(head_dim={node.head_dim})")
        self.pool.put(qkv)
        if windowed:
            am = self.pool.get(M, C)
            native.gather_rows(att, merge, am)
            self.pool.put(att)
            att = am
        self.linear(att, self.linear_spec(fsa[2]), res=tok, out=tok)
        self.pool.put(att)
        _expect(len(r2) == 2 and isa(r2[0], "LayerNorm") and isa(r2[1], "FeedForward"), "unexpected MLP residual")
        ff = kids(r2[1])
        _expect(len(ff) == 3 and isa(ff[1], "GeLU") and ff[1].approximation.value == "none", "unexpected FeedForward layout")
        h2 = self.layernorm(tok, r2[0])
        f = self.linear(h2, self.linear_spec(ff[0]), gelu=True)
        self.pool.put(h2)
        self.linear(f, self.linear_spec(ff[2]), res=tok, out=tok)
        self.pool.put(f)
        for extra in ch[2:]:
            _expect(isa(extra, "SetContext") and early is not None, f"unexpected {cname(extra)} at the end of a TransformerLayer")
            flat = early.view(M, C)
            self.python(lambda flat=flat, tok=tok: flat.copy_(tok), "copy:early_vit_embedding")
        return tok

    def neck(self, neck: Any, tok: Tensor, B: int, gh: int, gw: int, out: Tensor) -> None:
        ch = kids(neck)
        _expect([cname(c) for c in ch] == ["Permute", "Conv2d", "LayerNorm2d", "Conv2d", "LayerNorm2d"], "unexpected Neck layout")
        c1, n1, c2, n2 = ch[1], ch[2], ch[3], ch[4]
        _expect(c1.kernel_size == (1, 1) and c1.bias is None and c2.bias is None, "unexpected Neck convolutions")
        w1 = self.cache.get(("neck1",) + PackCache.ident(c1.weight), lambda: self.cvt(c1.weight.detach().reshape(c1.out_channels, c1.in_channels)))
        y = self.pool.get(tok.shape[0], c1.out_channels)
        native.gemm([(tok, w1)], y)
        self.pool.put(tok)
        y1 = self.pool.get(*y.shape)
        native.layernorm(y, self._w(n1.weight), self._w(n1.bias), n1.eps, y1)
        self.pool.put(y)
        z = self.conv(Act(y1, B, gh, gw), self.conv_spec(c2))
        self.pool.put(y1)
        z1 = self.pool.get(*z.t.shape)
        native.layernorm(z.t, self._w(n2.weight), self._w(n2.bias), n2.eps, z1)
        self.pool.put(z.t)
        native.nhwc_to_nchw(Act(z1, B, gh, gw).tokens(), out, z1.shape[1])
        self.pool.put(z1)


class CompiledSAMViT:
    """`fast = CompiledSAMViT(vit); embedding = fast(image)` == `vit(image)`; with a SAMViTAdapter injected the early ViT
    embedding is written to context "hq_sam".early_vit_embedding exactly like the SetContext node does."""

    def __init__(self, vit: Any, lora_mode: str = "merged") -> None:
        native.load()
        self.vit = vit
        self.lora_mode = lora_mode
        self.cache = PackCache()
        self.key: Any = None
        self.stats: dict[str, Any] = {}

    def _hook(self) -> Optional[Any]:
        for m in self.vit.modules():
            if isa(m, "SetContext") and m.context == "hq_sam":
                return m
        return None

    @torch.no_grad()
    def __call__(self, image: Tensor) -> Tensor:
        dtype = self.vit.dtype
        key = (tree_epoch(), tuple(image.shape), dtype, image.device)
        if key != self.key:
            B = image.shape[0]
            self.x = torch.empty(tuple(image.shape), device=image.device, dtype=dtype)
            neck_conv = [m for m in self.vit.modules() if isa(m, "Conv2d")][-1]
            gh, gw = image.shape[2] // self.vit.patch_size, image.shape[3] // self.vit.patch_size
            self.out = torch.empty(B, neck_conv.out_channels, gh, gw, device=image.device, dtype=dtype)
            self.early = torch.empty(B, gh, gw, self.vit.embedding_dim, device=image.device, dtype=dtype) if self._hook() is not None else None
            low = SAMLowering(image.device, dtype, self.cache, self.lora_mode)
            low.lower(self.vit, self.x, self.out, self.early)
            self.cache.sweep()
            self.low, self.key = low, key
            self.stats = dict(low.stats, step_ops=len(low.step), pool_bytes=low.step_pool.bytes())
        self.x.copy_(image)
        native.replay(self.low.step)
        hook = self._hook()
        if hook is not None:
            hook(self.early.clone())  # SetContext.__call__: stores into the parent's context, as the unfused tree does
        return self.out.clone()
