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

The attention (16 heads of 80 + the additive decomposed relative-position bias of image_encoder.py:82-127) runs on
mi355x_attention_general: the bias is bilinear in the query, bias[q, (a', b')] = q . E1[a - a' + S - 1] + q . E2[b - b' + S - 1],
so it is carried by EXTRA COLUMNS of Q and K instead of an L x L matrix:

  QP GEMM   h -> per head [ q / sqrt(d) | q . E1[r] for all 2S-1 offsets | q . E2[r] | pad ]   (tables folded into the weights)
  relpos_pack  -> Q' = [ q / sqrt(d) | the S offsets this token's row needs | the S its column needs | 0 ]   (contiguous windows)
  K' GEMM   h -> [ k | onehot(a') | onehot(b') | 0 ]   (the one-hot part is a constant residual operand of the GEMM)
  V^T GEMM  (no bias: softmax rows sum to one, so V's bias passes through attention unchanged and is folded into the
             output projection's bias, b_o' = b_o + W_o b_v)
  attention_general(Q', K', V^T, scale = 1)  ==  softmax(q k^T / sqrt(d) + bias) v, the score matrix never touches HBM.
"""
from __future__ import annotations

from typing import Any, Optional

import torch
from torch import Tensor

from .. import native
from ..fluxion.tree import tree_epoch
from .compiled import Program
from .lowering_blocks import BlockLowering as Lowering
from .packing import Act, PackCache, Unsupported, _expect, cname, isa, kids, launches


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
        h = h_full = self.layernorm(tok, r1[0])
        if windowed:
            ws = layer.window_size
            part, merge, nwin = self._window_tables(B, gh, gw, ws)
            h = self.pool.get(nwin * ws * ws, C)
            native.gather_rows(h_full, part, h)
            shape = (nwin, ws, ws)
        else:
            shape = (B, gh, gw)
        node = fsa[1]
        qkv_spec, out_spec = self.linear_spec(fsa[0]), self.linear_spec(fsa[2])
        aligned = windowed or (gh * gw) % 64 == 0  # global attention over a key count that is not a multiple of 64: no padded V^T source
        packs = self.relpos_packs(qkv_spec, out_spec, node, shape) if qkv_spec.lora is None and out_spec.lora is None and aligned else None
        if packs is not None:
            att = self.relpos_attention(h, h_full, packs, node, shape, B, gh, gw, layer.window_size if windowed else None)
            self.pool.put(h)
            if windowed:
                self.pool.put(h_full)
            out_bias = packs["bo"]
        else:
            qkv = self.linear(h, qkv_spec)
            self.pool.put(h)
            if windowed:
                self.pool.put(h_full)
            att = self.pool.get(qkv.shape[0], C)

            def attend(att: Tensor = att, qkv: Tensor = qkv, shape: tuple = shape, node: Any = node) -> None:
                # the node's own torch forward (shapes mi355x_attention_general does not cover, or un-merged LoRAs);
                # arguments are bound NOW: `att` is rebound to the merged buffer a few lines below
                att.view(*shape, C).copy_(node(qkv.view(*shape, 3 * C)))

            self.python(attend, "torch:RelativePositionAttention")
            self.stats["fallback_nodes"].append(f"RelativePositionAttention(head_dim={node.head_dim})")
            self.pool.put(qkv)
            out_bias = out_spec.b
        if windowed:
            am = self.pool.get(M, C)
            native.gather_rows(att, merge, am)
            self.pool.put(att)
            att = am
        native.gemm([(att, self.kblocked(out_spec.w))], tok, bias=out_bias, res=tok)
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

    # ---------------------------------------------------------------------------------------------------------------
    def relpos_packs(self, qkv: Any, out: Any, node: Any, shape: tuple) -> Optional[dict[str, Any]]:
        """Weights of the three projections re-packed for the bias-as-columns attention (module docstring), or None when
        the shape is outside what mi355x_attention_general instantiates."""
        H, d, C = node.num_heads, node.head_dim, node.embedding_dim
        S1, S2 = shape[1], shape[2]
        # (the reference unpacks spatial_size as (width, height) but indexes the grid (first, second); every SAM config is square)
        _expect(S1 == S2 and tuple(node.spatial_size) == (S1, S2), "RelativePositionAttention on a non-square / mismatching grid")
        _expect(tuple(node.vertical_embedding.shape) == (2 * S1 - 1, d) and tuple(node.horizontal_embedding.shape) == (2 * S2 - 1, d), "unexpected relative position tables")
        Dq = (d + S1 + S2 + 7) // 8 * 8
        Lp = (d + 2 * S1 - 1 + 2 * S2 - 1 + 7) // 8 * 8
        if not ((Dq <= 128 or Dq <= 224) and d <= 80 and d % 4 == 0) or qkv.w.shape[0] != 3 * C:
            return None
        o1, o2 = d + 2 * S1 - 1, d + 2 * S1 - 1 + 2 * S2 - 1
        ident = PackCache.ident(*(t for t in (qkv.w, qkv.b, out.w, out.b, node.vertical_embedding, node.horizontal_embedding) if t is not None))

        def make() -> dict[str, Any]:
            f = dict(device=self.device, dtype=torch.float32)
            if self.device.type == "meta":
                z = lambda *sh: torch.empty(*sh, device=self.device, dtype=self.dtype)  # noqa: E731
                return dict(wqp=z(H * Lp, C), bqp=z(H * Lp), wkp=z(H * Dq, C), bkp=z(H * Dq), wv=z(C, C), bo=z(C))
            W = qkv.w.detach().to(**f)
            b = qkv.b.detach().to(**f) if qkv.b is not None else torch.zeros(3 * C, **f)
            Wq, Wk, Wv = W[:C].view(H, d, C), W[C : 2 * C].view(H, d, C), W[2 * C :]
            bq, bk, bv = b[:C].view(H, d), b[C : 2 * C].view(H, d), b[2 * C :]
            E1 = node.vertical_embedding.detach().to(**f).flip(0)    # first grid axis;  P1[r] = q . E1_table[2*S1 - 2 - r]
            E2 = node.horizontal_embedding.detach().to(**f).flip(0)  # second grid axis
            wqp, bqp = torch.zeros(H, Lp, C, **f), torch.zeros(H, Lp, **f)
            wqp[:, :d], bqp[:, :d] = Wq * d ** -0.5, bq * d ** -0.5
            # P_e[h][r][c] = sum_d E_e[r][d] Wq[h][d][c] (and the same with the bias as one more column): per head a small matrix product, on
            # the library's own float32 kernel (Lowering._mm) like every other piece of weight preparation
            Wqb = torch.cat([Wq, bq.unsqueeze(-1)], dim=-1)  # [H, d, C + 1]
            for E, lo, hi in ((E1, d, o1), (E2, o1, o2)):
                for h in range(H):
                    pe = self._mm(E, Wqb[h].t())  # [r, C + 1]
                    wqp[h, lo:hi], bqp[h, lo:hi] = pe[:, :C], pe[:, C]
            wkp, bkp = torch.zeros(H, Dq, C, **f), torch.zeros(H, Dq, **f)
            wkp[:, :d], bkp[:, :d] = Wk, bk
            bo = self._mm(out.w.detach().to(**f), bv.unsqueeze(0)).reshape(-1) + (out.b.detach().to(**f) if out.b is not None else 0)
            c = lambda t: t.to(self.dtype).contiguous()  # noqa: E731
            return dict(wqp=c(wqp.view(H * Lp, C)), bqp=c(bqp.view(-1)), wkp=c(wkp.view(H * Dq, C)), bkp=c(bkp.view(-1)), wv=c(Wv), bo=c(bo))

        packs = dict(self.cache.get(("sam_relpos", S1, S2) + ident, make))
        packs.update(Dq=Dq, Lp=Lp, S1=S1, S2=S2)
        return packs

    def _onehot_rows(self, n: int, H: int, d: int, S1: int, S2: int, Dq: int) -> Tensor:
        """[n * S1 * S2, H * Dq]: ones at (d + a) and (d + S1 + b) of every head for the token at grid position (a, b)."""
        def make() -> Tensor:
            t = torch.zeros(S1 * S2, H, Dq, device=self.device, dtype=self.dtype)
            if self.device.type != "meta":
                idx = torch.arange(S1 * S2, device=self.device)
                t[idx, :, d + idx // S2] = 1
                t[idx, :, d + S1 + idx % S2] = 1
            return t.view(S1 * S2, H * Dq).repeat(n, 1).contiguous()

        return self.cache.get(("sam_onehot", n, H, d, S1, S2, Dq, self.dtype), make)

    def relpos_attention(self, h: Tensor, h_full: Tensor, packs: dict[str, Any], node: Any, shape: tuple, B: int, gh: int, gw: int, ws: Optional[int]) -> Tensor:
        """h: [n * L, C] (window-partitioned rows of h_full when ws is set) -> attention output [n * L, C] (before WindowMerge)."""
        H, d, C = node.num_heads, node.head_dim, node.embedding_dim
        n, S1, S2 = shape
        L, Dq, Lp = S1 * S2, packs["Dq"], packs["Lp"]
        Mw = n * L
        qp = self.pool.get(Mw, H * Lp)
        native.gemm([(h, self.kblocked(packs["wqp"]))], qp, bias=packs["bqp"])
        q = self.pool.get(Mw, H * Dq)
        native.relpos_pack(qp, q, H, d, S1, S2, Lp, Dq)
        self.pool.put(qp)
        k = self.pool.get(Mw, H * Dq)
        native.gemm([(h, self.kblocked(packs["wkp"]))], k, bias=packs["bkp"], res=self._onehot_rows(n, H, d, S1, S2, Dq))
        lkp = (L + 63) // 64 * 64
        if lkp == L:
            vt = self.pool.get(C, Mw)
            native.gemm([(self.kblocked(packs["wv"]), h)], vt, weight_operand="x")
        else:
            # every window's V^T columns start on a 64-key boundary: a second gather of the SAME LayerNorm output with
            # the windows padded to lkp rows (zero rows -> zero V^T columns, which masked keys multiply by exactly 0)
            part_v = self._window_tables_padded(B, gh, gw, ws, lkp)
            hv = self.pool.get(n * lkp, C)
            native.gather_rows(h_full, part_v, hv)
            vt = self.pool.get(C, n * lkp)
            native.gemm([(self.kblocked(packs["wv"]), hv)], vt, weight_operand="x")
            self.pool.put(hv)
        att = self.pool.get(Mw, C)
        native.attention_general(q.view(n, L, H * Dq), k.view(n, L, H * Dq), vt.view(C, n, lkp), att.view(n, L, C), H, L, scale=1.0)
        self.pool.put(q)
        self.pool.put(k)
        self.pool.put(vt)
        return att

    def _window_tables_padded(self, B: int, Hh: int, W: int, ws: int, lkp: int) -> Tensor:
        def make() -> Tensor:
            part, _merge, nwin = self._window_tables(B, Hh, W, ws)
            t = torch.full((nwin, lkp), -1, dtype=torch.int32, device=self.device)
            t[:, : ws * ws] = part.view(nwin, ws * ws)
            return t.reshape(-1).contiguous()

        return self.cache.get(("windows_padded", B, Hh, W, ws, lkp), make)

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

    def __init__(self, vit: Any, lora_mode: str = "merged", use_graph: bool = True) -> None:
        native.load()
        self.vit = vit
        self.lora_mode = lora_mode
        self.use_graph = use_graph
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
            # the torch fallback of an attention (un-merged LoRA, odd shapes) allocates: not capturable
            self.program = Program(low.step, self.use_graph and not low.stats["fallback_nodes"], low=low)
            self.stats = dict(low.stats, step_ops=launches(low.step), pool_bytes=low.step_pool.bytes())
        self.x.copy_(image)
        self.program.run()
        hook = self._hook()
        if hook is not None:
            hook(self.early.clone())  # SetContext.__call__: stores into the parent's context, as the unfused tree does
        return self.out.clone()
