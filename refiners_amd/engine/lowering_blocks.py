"""Block-level emitters of the lowering: attention (self / cross, one or two K/V streams, head shapes), FeedForward, CrossAttentionBlock(2d),
ResidualBlock and the torch fallback for sub-trees nothing matches -- everything between the GEMM / conv / norm emitters of
refiners_amd.engine.lowering.Lowering and the UNet walker of refiners_amd.engine.unet_lowering."""
from __future__ import annotations

import os

import math
from dataclasses import dataclass, field
from typing import Any, Callable, Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .. import native

from .lowering import Lowering
from .packing import Act, CatAct, ConvSpec, LinSpec, LoraPack, PackCache, Pool, Unsupported, _expect, cname, isa, kids, launches  # noqa: F401


class BlockLowering(Lowering):
    def _split_attention(self, att: Any, allow_causal: bool = False) -> tuple[list[Any], Any, Any, Optional[Any]]:
        """Attention | SelfAttention | CrossAttentionAdapter(Attention) -> ([q, k, v nodes], sdpa-like, out node, ip)."""
        if isa(att, "CrossAttentionAdapter"):
            att = kids(att)[0]
        _expect(isa(att, "Attention"), f"expected an Attention chain, got {cname(att)}")
        ch = [c for c in kids(att) if not isa(c, "SelfAttentionMap")]  # the SAG tap (handled by self_attention) stores probabilities, changes nothing
        if isa(att, "SelfAttention"):
            _expect(len(ch) == 4 and isa(ch[0], "Parallel") and all(isa(c, "Identity") for c in kids(ch[0])), "unexpected SelfAttention layout")
            ch = ch[1:]
        _expect(len(ch) == 3 and isa(ch[0], "Distribute") and len(kids(ch[0])) == 3, "unexpected Attention layout")
        sd = ch[1]
        ip = None
        if isa(sd, "Sum"):
            sc = kids(sd)
            _expect(len(sc) == 2 and isa(sc[0], "ScaledDotProductAttention") and isa(sc[1], "ImageCrossAttention"), "unexpected Sum around SDPA")
            ip, sd = sc[1], sc[0]
        _expect(isa(sd, "ScaledDotProductAttention") and (allow_causal or not sd.is_causal), "causal or unknown SDPA node")
        _expect(sd.num_heads == att.num_heads, "head count mismatch")
        return kids(ch[0]), sd, ch[2], ip

    def sdpa(self, q: Tensor, B: int, heads: int, streams: list[tuple[Tensor, Tensor, int, float]], v_plain: Optional[list[Tensor]] = None) -> Tensor:
        """q: [B*Lq, C]; streams: (k [B*Lkp, C], vt [C, B*Lkp], Lk, out_scale) with Lkp = rows per sample."""
        M, C = q.shape
        Lq = M // B
        out = self.pool.get(M, C)
        d = C // heads
        kind = self.head_kernel(d)
        if kind is not None:
            q3 = q.as_strided((B, Lq, C), (Lq * q.stride(0), q.stride(0), 1))
            st = []
            for k, vt, Lk, osc in streams:
                lkp = k.shape[0] // B
                kv = k.as_strided((B, lkp, C), (lkp * k.stride(0), k.stride(0), 1))  # k may be a column slice of a packed [Q|K] buffer
                lv = vt.shape[1] // B
                st.append((kv, vt.as_strided((C, B, lv), (vt.stride(0), lv, 1)), Lk, osc))  # vt rows may be padded (stride > B lv)
            if kind == "flash64":
                native.attention(q3, out.view(B, Lq, C), heads, st)
                return out
            # other head dims (SD1.5: 40 / 80 / 160): one launch per K/V stream, the image-prompt stream is accumulated
            for i, (kv, vt3, Lk, osc) in enumerate(st):
                dst = out if i == 0 else self.pool.get(M, C)
                native.attention_general(q3, kv, vt3, dst.view(B, Lq, C), heads, Lk, out_scale=osc)
                if i > 0:
                    native.axpby(out, 1.0, dst, 1.0, out)
                    self.pool.put(dst)
            return out
        # head dims no flash kernel covers (the VAE's single 512-wide head over H*W tokens): S = Q K^T (float32 scores), row
        # softmax, O = P V as three native launches per (sample, head) -- the score matrix is 1 GB at 1024x1024 px, nothing
        # next to 288 GB of HBM, and both GEMMs run at matrix-core speed (K = 512 and K = H*W)
        assert v_plain is not None
        kblk = 128 // self.es
        if d % kblk == 0 and all(Lk % kblk == 0 and k.shape[0] == B * Lk for (k, _v, Lk, _o) in streams) and len(streams) == 1 and streams[0][3] == 1.0 and self.device.type != "meta":
            (k, _unused, Lk, osc), v = streams[0], v_plain[0]
            vt = self.pool.get(B * C, Lk)  # [B][C][Lk]: V^T per sample
            native.nhwc_to_nchw(v.view(B, Lk, C), vt.view(B, C, Lk, 1), C)
            scores = self.__dict__.setdefault("_wide_scores", {})
            if (Lq, Lk) not in scores:
                scores[(Lq, Lk)] = (torch.empty(Lq, Lk, device=self.device, dtype=torch.float32), torch.empty(Lq, Lk, device=self.device, dtype=self.dtype))
            sc, pr = scores[(Lq, Lk)]
            for b in range(B):
                for hh in range(heads):
                    qb = q[b * Lq : (b + 1) * Lq, hh * d : (hh + 1) * d]
                    kb = k[b * Lk : (b + 1) * Lk, hh * d : (hh + 1) * d]
                    native.gemm([(qb, kb)], sc, out_f32=self.dtype != torch.float32)
                    native.softmax_rows(sc, pr, Lk, d ** -0.5)
                    vtb = vt.view(B, C, Lk)[b, hh * d : (hh + 1) * d]
                    ob = out[b * Lq : (b + 1) * Lq, hh * d : (hh + 1) * d]
                    native.gemm([(pr, vtb)], ob)
            self.pool.put(vt)
            return out

        def run() -> None:
            acc = None
            for (k, _vt, Lk, osc), v in zip(streams, v_plain):
                lkp = k.shape[0] // B
                qh = q.view(B, Lq, heads, d).transpose(1, 2)
                kh = k.view(B, lkp, heads, d)[:, :Lk].transpose(1, 2)
                vh = v.view(B, lkp, heads, d)[:, :Lk].transpose(1, 2)
                y = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(M, C)
                acc = y * osc if acc is None else acc + y * osc
            out.copy_(acc)

        self.python(run, f"torch_sdpa_d{d}")
        self.stats["fallback_nodes"].append(f"SDPA(head_dim={d})")
        return out

    def head_kernel(self, d: int) -> Optional[str]:
        """Which attention kernel serves head dim d: mi355x_attention (64), mi355x_attention_general (<= 160, 16-byte rows), none."""
        if d == 64:
            return "flash64"
        es = 4 if self.dtype == torch.float32 else 2
        if d <= 160 and (d * es) % 16 == 0 and d % 4 == 0:
            return "general"
        return None

    @staticmethod
    def _pad_keys(n: int) -> int:
        return (n + 63) // 64 * 64

    def project_kv(self, src: Tensor, B: int, k_node: Any, v_node: Any, heads: int) -> tuple[Tensor, Tensor, Optional[Tensor]]:
        """K rows [B*Lp, C] and V^T [C, B*Lp] of a key/value source [B*Lp, Ck] (Lp = keys per sample, padded to 64)."""
        ks, vs = self.linear_spec(k_node), self.linear_spec(v_node)
        _expect(ks.b is None and vs.b is None, "key / value projections with bias are not supported")
        C = ks.N
        k = self.pool.get(src.shape[0], C)
        self.pool.pin(k)
        self.linear(src, ks, out=k)
        if self.head_kernel(C // heads) is not None:
            vt = self.pool.get(C, src.shape[0])
            self.pool.pin(vt)
            self.linear_T(src, vs, vt)
            return k, vt, None
        v = self.pool.get(src.shape[0], C)
        self.pool.pin(v)
        self.linear(src, vs, out=v)
        return k, v, v

    def _project_vt(self, h: Tensor, vs: LinSpec, B: int, L: int, C: int) -> Tensor:
        """V^T [C, B * Lp] of h [B * L, Ck] (Lp = L rounded up to 64 keys)."""
        if L % 64 == 0:
            vt = self.pool.get(C, B * L)
            self.linear_T(h, vs, vt)
            return vt
        # token counts that are not a multiple of 64 (e.g. 1216x832 px -> 38x26 = 988 tokens at the deepest level): each
        # sample's V^T columns start on a 64-key boundary (16-byte aligned rows, readable up to the padded length),
        # one projection launch per sample; the padding is zeroed ONCE here (the kernel masks those keys' scores
        # but still multiplies their V by an exact 0, so it must be finite)
        lp = (L + 63) // 64 * 64
        vt = torch.zeros(C, B * lp, device=self.device, dtype=self.dtype)
        self.__dict__.setdefault("_keep", []).append(vt)
        for b in range(B):
            self.linear_T(h[b * L : (b + 1) * L], vs, vt[:, b * lp : b * lp + L])
        return vt

    def self_attention(self, x: Tensor, B: int, ln: Any, att: Any, stats: Optional[Tensor] = None, stats_out: Optional[Tensor] = None) -> Tensor:
        """x += Wo SDPA(Wq h, Wk h, Wv h), h = LN(x)   (cross_attention.py:44-49; attentions.py:319-385).
        `stats`: row statistics of x (LayerNorm then runs inside the projection launches); `stats_out`: buffer for the
        statistics of the updated x."""
        (qn, kn, vn), sd, on, ip = self._split_attention(att)
        _expect(ip is None, "image cross-attention on a self-attention")
        heads = sd.num_heads
        qs, ks, vs = self.linear_spec(qn), self.linear_spec(kn), self.linear_spec(vn)
        _expect(qs.b is None and ks.b is None and vs.b is None, "q/k/v bias not supported")
        M, C = x.shape
        native_path = self.head_kernel(C // heads) is not None
        L = M // B
        fold = self.ln_fusable(stats, qs, ks, vs)
        lnarg = (stats, ln) if fold else None
        h = x if fold else self.layernorm(x, ln)
        no_lora = qs.lora is None and ks.lora is None and vs.lora is None
        all_inlaunch = self.lora_inlaunch and all(sp.lora is not None and sp.lora.a_kb is not None for sp in (qs, ks, vs)) and len({sp.lora.R for sp in (qs, ks, vs)}) == 1
        qk = q = k = vt = None
        if (no_lora or all_inlaunch) and native_path and self.qkv_merge and L % 64 == 0 and C % 128 == 0 and self.device.type != "meta":
            # ONE launch for the three projections: [Wq; Wk; Wv] stacked, Q | K row-major, V stored transposed
            wqkv = LinSpec(self.cache.get(("qkv",) + PackCache.ident(qs.w, ks.w, vs.w), lambda: torch.cat([qs.w, ks.w, vs.w], 0).contiguous()), None)
            qk = self.pool.get(M, 2 * C)
            vt = self.pool.get(C, M)
            if fold:
                wl, ls, lc = self.ln_fold(wqkv, ln)
                lo = sy = None
                if not no_lora:
                    packs = [self.ln_fold_lora(sp.lora, ln) for sp in (qs, ks, vs)]
                    bs = self.cache.get(("qkv_bs",) + PackCache.ident(qs.lora.bs_r, ks.lora.bs_r, vs.lora.bs_r), lambda: torch.cat([qs.lora.bs_r, ks.lora.bs_r, vs.lora.bs_r], 0).contiguous())
                    sy = self.lora_sync(3, M, qs.lora.R)
                    lsc = self.cache.get(("qkv_lsc",) + PackCache.ident(*[t for pk in packs for t in pk[1:]]), lambda: (torch.cat([pk[1] for pk in packs]).contiguous(), torch.cat([pk[2] for pk in packs]).contiguous()))
                    lo = ([(0, packs[0][0]), (C, packs[1][0]), (2 * C, packs[2][0])], bs, lsc[0], lsc[1])
                native.gemm([(h, self.kblocked(wl))], qk, out_t=vt, nt_begin=2 * C, ln=(stats, ls, lc, float(ln.eps)), lora=lo, lora_sync=sy)
                if sy is not None:
                    self.pool.put(sy[0])
            elif no_lora:
                native.gemm([(h, self.kblocked(wqkv.w))], qk, out_t=vt, nt_begin=2 * C)
            else:  # three LoRA sets in one launch: a stacked-down block per column group, the up rows stacked like the weights
                bs = self.cache.get(("qkv_bs",) + PackCache.ident(qs.lora.bs_r, ks.lora.bs_r, vs.lora.bs_r), lambda: torch.cat([qs.lora.bs_r, ks.lora.bs_r, vs.lora.bs_r], 0).contiguous())
                sy = self.lora_sync(3, M, qs.lora.R)
                native.gemm([(h, self.kblocked(wqkv.w))], qk, out_t=vt, nt_begin=2 * C, lora=([(0, qs.lora.a_kb), (C, ks.lora.a_kb), (2 * C, vs.lora.a_kb)], bs), lora_sync=sy)
                self.pool.put(sy[0])
            q, k = qk[:, :C], qk[:, C:]
        else:
            if fold and (not (native_path and L % 64 == 0) or not no_lora):
                h, lnarg, fold = self.layernorm(x, ln), None, False  # the per-sample / torch V paths (and separate LoRA launches) want a materialised h
            if qs.lora is None and ks.lora is None:
                wqk = LinSpec(self.cache.get(("qk",) + PackCache.ident(qs.w, ks.w), lambda: torch.cat([qs.w, ks.w], 0).contiguous()), None)
                qk = self.linear(h, wqk, ln=lnarg)
                q, k = qk[:, :C], qk[:, C:]
            else:
                q = self.linear(h, qs)
                k = self.linear(h, ks)
            if native_path:
                if fold:  # V^T = (Wv LN(x)^T): the transposed column group alone (nt_begin = 0)
                    vt = self.pool.get(C, M)
                    wl, ls, lc = self.ln_fold(vs, ln)
                    native.gemm([(h, self.kblocked(wl))], None, out_t=vt, nt_begin=0, ln=(stats, ls, lc, float(ln.eps)))
                else:
                    vt = self._project_vt(h, vs, B, L, C)
        tap = next((c for c in kids(att) if isa(c, "SelfAttentionMap")), None)
        if tap is not None and getattr(self, "sag_capture", True):
            self.sag_attention_mass(q, k, B, heads, L, C)
        if native_path:
            o = self.sdpa(q, B, heads, [(k, vt, L, 1.0)])
            if L % 64 == 0:
                self.pool.put(vt)
        else:
            v = self.linear(h, vs)
            o = self.sdpa(q, B, heads, [(k, v, M // B, 1.0)], v_plain=[v])
            self.pool.put(v)
        if h is not x:
            self.pool.put(h)
        if qk is not None:
            self.pool.put(qk)
        else:
            self.pool.put(q)
            self.pool.put(k)
        self.linear(o, self.linear_spec(on), res=x, out=x, stats_out=stats_out)
        self.pool.put(o)
        return x

    def sag_attention_mass(self, q: Tensor, k: Tensor, B: int, heads: int, L: int, C: int) -> None:
        """Self-Attention Guidance tap (SelfAttentionMap + SAGAdapter.compute_sag_mask, self_attention_guidance.py:22-84): for the
        UNCONDITIONAL half of the CFG batch, mass[b][j] = mean over heads of the attention key j receives from all queries.  The
        reference materialises softmax(Q K^T / sqrt(d)) for every head and sample; only these column sums are ever used, so per
        (sample, head): scores GEMM (float32) -> row softmax -> column sum, three small launches on an L x L scratch."""
        d = C // heads
        kblk = 128 // self.es
        _expect(d % kblk == 0 and B % 2 == 0, "self-attention guidance tap: head width / batch not supported")
        n = B // 2
        mass = torch.zeros(n, L, device=self.device, dtype=torch.float32)
        sc = torch.empty(L, L, device=self.device, dtype=torch.float32)
        pr = torch.empty(L, L, device=self.device, dtype=self.dtype)
        self.__dict__.setdefault("_keep", []).extend([mass, sc, pr])
        if self.device.type != "meta":
            for b in range(n):
                for h in range(heads):
                    qb, kb = q[b * L : (b + 1) * L, h * d : (h + 1) * d], k[b * L : (b + 1) * L, h * d : (h + 1) * d]
                    native.gemm([(qb, kb)], sc, out_f32=self.dtype != torch.float32)
                    native.softmax_rows(sc, pr, L, d ** -0.5)
                    native.colsum_rows(pr, mass[b], accumulate=h > 0, scale=1.0 / heads)
        self.sag = {"mass": mass, "tokens": L}

    def cross_attention(self, x: Tensor, B: int, ln: Any, par: Any, att: Any, ctx: "UNetContext", stats: Optional[Tensor] = None,
                        stats_out: Optional[Tensor] = None) -> Tensor:
        """x += Wo (SDPA(Wq LN(x), K_text, V_text) [+ s SDPA(q, K_img, V_img)])   (cross_attention.py:50-68,
        image_prompt.py:237-309).  K / V^T of the text and image tokens are produced in the prologue."""
        pc = kids(par)
        _expect(len(pc) == 3 and isa(pc[0], "Identity") and all(isa(c, "UseContext") for c in pc[1:]), "unexpected cross-attention Parallel")
        _expect(pc[1].context == pc[2].context and pc[1].key == pc[2].key, "key and value read different contexts")
        (qn, kn, vn), sd, on, ip = self._split_attention(att)
        heads = sd.num_heads
        src, Lk = ctx.tokens(pc[1].context, pc[1].key)
        with self.in_prologue():
            k, v_or_vt, v_plain = self.project_kv(src, B, kn, vn, heads)
        streams = [(k, v_or_vt, Lk, 1.0)]
        plains = [v_plain]
        if ip is not None:
            ic = kids(ip)
            _expect(len(ic) == 3 and isa(ic[0], "Distribute") and isa(ic[1], "ScaledDotProductAttention") and isa(ic[2], "Multiply"), "unexpected ImageCrossAttention layout")
            dc = kids(ic[0])
            _expect(len(dc) == 3 and isa(dc[0], "Identity"), "unexpected ImageCrossAttention Distribute")
            kc, vc = kids(dc[1]), kids(dc[2])
            _expect(len(kc) == 2 and len(vc) == 2 and isa(kc[0], "UseContext") and isa(vc[0], "UseContext"), "unexpected image K/V branch")
            _expect(ic[2].bias == 0.0 and ic[1].num_heads == heads, "unexpected ImageCrossAttention parameters")
            isrc, ilk = ctx.tokens(kc[0].context, kc[0].key)
            with self.in_prologue():
                k2, v2, vp2 = self.project_kv(isrc, B, kc[1], vc[1], heads)
            streams.append((k2, v2, ilk, float(ic[2].scale)))
            plains.append(vp2)
            self.stats["ip_sites"] += 1
        qspec = self.linear_spec(qn)
        M, C = x.shape[0], qspec.N
        if self.ln_fusable(stats, qspec):
            q = self.linear(x, qspec, ln=(stats, ln))
        else:
            h = self.layernorm(x, ln)
            q = self.linear(h, qspec)
            self.pool.put(h)
        o = self.sdpa(q, B, heads, streams, v_plain=plains if plains[0] is not None else None)
        self.pool.put(q)
        self.linear(o, self.linear_spec(on), res=x, out=x, stats_out=stats_out)
        self.pool.put(o)
        return x

    def feed_forward(self, x: Tensor, ln: Any, w1: Any, glu: Any, w2: Any, stats: Optional[Tensor] = None, stats_out: Optional[Tensor] = None) -> Tensor:
        """x += W2 GEGLU(W1 LN(x))   (cross_attention.py:69-72): GEGLU is the epilogue of the first GEMM."""
        _expect(isa(glu, "GLU") and isa(glu.activation, "GeLU") and glu.activation.approximation.value == "none", "only GLU(GeLU(exact)) is fused")
        s1, s2 = self.linear_spec(w1, geglu=True), self.linear_spec(w2)
        # the intermediate [M, 4C] has 10 KB rows at C = 1280: the second GEMM would stream it at half rate, so the GEGLU epilogue
        # stores it K-blocked (same bytes, [column block][M][128 B]) whenever the kernel's vector store path applies
        inl = lambda sp: sp.lora is None or (sp.lora.a_kb is not None and self.lora_inlaunch)  # noqa: E731  (the two-launch LoRA path reads x row-major)
        blocked = self.kblock_policy > 0 and inl(s1) and inl(s2) and s1.N % 256 == 0 and self.device.type != "meta"
        if self.ln_fusable(stats, s1):
            ff = self.linear(x, s1, out_kblocked=blocked, ln=(stats, ln))
        else:
            h = self.layernorm(x, ln)
            ff = self.linear(h, s1, out_kblocked=blocked)
            self.pool.put(h)
        self.linear(native.KBlocked.adopt(ff.view(-1), ff.shape[0], ff.shape[1]) if blocked else ff, s2, res=x, out=x, stats_out=stats_out)
        self.pool.put(ff)
        return x

    def cross_attention_block(self, blk: Any, x: Tensor, B: int, ctx: "UNetContext", stats: Optional[Tensor] = None, last: bool = True) -> Tensor:
        """`stats`: the statistics buffer of x's size class when x's PRODUCER filled it (else None); every residual update
        inside the block refills it for the next LayerNorm -- except the last one of the last block (`last`)."""
        ch = kids(blk)
        _expect(len(ch) == 3 and all(isa(c, "Residual") for c in ch), "unexpected CrossAttentionBlock layout")
        r1, r2, r3 = (kids(c) for c in ch)
        _expect(len(r1) == 2 and len(r2) == 3 and len(r3) == 4, "unexpected CrossAttentionBlock residual bodies")
        buf = self.row_stats(x.shape[0], x.shape[1])
        x = self.self_attention(x, B, r1[0], r1[1], stats, buf)
        x = self.cross_attention(x, B, r2[0], r2[1], r2[2], ctx, buf, buf)
        return self.feed_forward(x, r3[0], r3[1], r3[2], r3[3], buf, None if last else buf)

    def cross_attention_2d(self, node: Any, a: Act, ctx: "UNetContext") -> Act:
        """CrossAttentionBlock2d (cross_attention.py:92-175).  Token-major layout makes flatten / transpose free."""
        ch = kids(node)
        _expect(len(ch) == 3 and all(isa(c, "Chain") for c in ch), "unexpected CrossAttentionBlock2d layout")
        head, blocks, tail = kids(ch[0]), kids(ch[1]), kids(ch[2])
        _expect(isa(head[0], "GroupNorm"), "CrossAttentionBlock2d must start with GroupNorm")
        proj_in = next((m for m in head[1:] if isa(m, "Linear", "Conv2d", "LoraAdapter")), None)
        proj_out = next((m for m in tail if isa(m, "Linear", "Conv2d", "LoraAdapter")), None)
        _expect(proj_in is not None and proj_out is not None, "projection layers not found")
        others = [m for m in head[1:] + tail if m is not proj_in and m is not proj_out]
        _expect(all(isa(m, "StatefulFlatten", "Transpose", "Parallel", "Unflatten") for m in others), "unexpected layers around the transformer")
        g = self.groupnorm(a, head[0], silu=False)
        pin = self.linear_spec(proj_in)
        stats = self.row_stats(g.t.shape[0], pin.N)
        h = self.linear(g.t, pin, stats_out=stats)
        self.pool.put(g.t)
        for i, blk in enumerate(blocks):
            _expect(isa(blk, "CrossAttentionBlock"), f"unexpected {cname(blk)} among transformer layers")
            h = self.cross_attention_block(blk, h, a.B, ctx, stats, last=i == len(blocks) - 1)
        po = self.linear_spec(proj_out)
        cs = self.colstats_for(h.shape[0], po.N, a.H * a.W)  # the next ResidualBlock normalises this tensor
        out = self.linear(h, po, res=a.t, colstats=cs)
        self.pool.put(h)
        return Act(out, a.B, a.H, a.W, cs)

    # -- ResidualBlock -------------------------------------------------------------------------------------------
    def residual_block(self, node: Any, a: Act, ctx: "UNetContext") -> Act:
        """conv2(SiLU(GN(conv1(SiLU(GN(x))) + time))) + shortcut(x)   (unet.py:6-51 + range_adapter.py:47-86):
        time bias and bias ride in conv1's epilogue, the shortcut (identity or 1x1 conv) in conv2's."""
        ch = kids(node)
        _expect(len(ch) == 2 and isa(ch[0], "Chain"), "unexpected ResidualBlock layout")
        body = kids(ch[0])
        _expect(len(body) == 6 and isa(body[0], "GroupNorm") and isa(body[1], "SiLU") and isa(body[3], "GroupNorm") and isa(body[4], "SiLU"), "unexpected ResidualBlock body")
        c1, c2 = self.conv_spec(body[2]), self.conv_spec(body[5])
        _expect(c1.stride == 1 and c2.stride == 1 and c2.time is None, "unexpected convolutions in ResidualBlock")
        src = a
        if isinstance(a, CatAct):
            # input = a ResidualConcatenator's (x | skip) that was never written: GroupNorm reads both parts, the 1x1 shortcut takes them as two
            # K segments of conv2.  Anything that needs the tensor itself (identity shortcut cannot happen: channel counts differ; a shortcut
            # with LoRAs; a third segment that does not fit) gets it materialised.
            sc0 = None if isa(ch[1], "Identity") else self.conv_spec(ch[1])
            seg_ok = sc0 is not None and sc0.ksize == 1 and sc0.lora is None and (c2.lora is None or (c2.lora.a_kb is not None and self.lora_inlaunch))
            if not seg_ok or os.environ.get("REFINERS_AMD_CAT_FUSE", "1") == "0":
                a = self.materialise(a)
                self.stats["concat_materialised"] = self.stats.get("concat_materialised", 0) + 1
        g1 = self.groupnorm(a, body[0], silu=True)
        rb = ctx.time_bias(c1) if c1.time is not None else None
        h1 = self.conv(g1, c1, rowbias=rb)
        self.pool.put(g1.t)
        g2 = self.groupnorm(h1, body[3], silu=True)
        self.pool.put(h1.t)
        if isa(ch[1], "Identity"):
            out = self.conv(g2, c2, res=a.t)
        else:
            sc = self.conv_spec(ch[1])
            _expect(sc.ksize == 1 and sc.time is None, "unexpected shortcut")
            if sc.lora is None and (c2.lora is None or (c2.lora.a_kb is not None and self.lora_inlaunch)):  # (an in-launch LoRA adapts segment 0: the shortcut rides along)
                both = self.cache.get(("bias_sum",) + PackCache.ident(c2.b, sc.b), lambda: (c2.b.float() + sc.b.float()).to(self.dtype))
                out = self.conv(g2, c2, shortcut=(a, sc), bias=both)
            else:
                s = self.conv(a, sc)
                out = self.conv(g2, c2, res=s.t)
                self.pool.put(s.t)
        self.pool.put(g2.t)
        if isinstance(src, CatAct):  # the block consumed the concatenation: its first part goes back to the pool (the skip stays pinned) ...
            self.pool.put(src.a.t)
            if a is not src:         # ... and so does the materialised copy, where one had to be made
                self.pool.put(a.t)
        return out

    # -- generic fallback -------------------------------------------------------------------------------------------
    def torch_node(self, node: Any, a: Act, out_channels: Optional[int] = None, out_hw: Optional[tuple[int, int]] = None, what: str = "") -> Act:
        """Run an unrecognised sub-tree through its own torch forward on an NCHW copy: the node-level half of the section 8(b) error convention
        ("unsupported => fall back to the stock child loop, never error", fluxion/layers/chain.py:226-243).  Only for sub-trees that do not touch
        the context store: at run time the Chain's contexts do not hold this program's tensors (residual slots, embeddings live in the lowered
        program), so a node that reads or writes them -- FreeU's concatenator, reference-only / StyleAligned attention injections -- makes the WHOLE
        tree fall back to the stock forward instead (Unsupported -> CompiledUNet.__call__).  The output geometry is whatever the node
        produces on a zero image of the input's shape (one trial call at lowering time)."""
        ctx_nodes = sorted({cname(s) for s in node.modules() if isa(s, "UseContext", "SetContext")})
        if ctx_nodes:
            raise Unsupported(f"{cname(node)} is not a lowered pattern and contains {', '.join(ctx_nodes)}: it needs the stock Chain forward's context store")
        C2, (H2, W2) = out_channels or a.C, out_hw or (a.H, a.W)
        nchw = torch.empty(a.B, a.C, a.H, a.W, device=self.device, dtype=self.dtype)
        if out_channels is None and out_hw is None and self.device.type != "meta":
            with torch.no_grad():
                trial = node(torch.zeros_like(nchw))
            if not (isinstance(trial, Tensor) and trial.dim() == 4 and trial.shape[0] == a.B):
                raise Unsupported(f"{cname(node)} is not a lowered pattern and does not map an image to an image")
            C2, H2, W2 = trial.shape[1], trial.shape[2], trial.shape[3]
            del trial
        res = torch.empty(a.B, C2, H2, W2, device=self.device, dtype=self.dtype)
        out = self.pool.get(a.B * H2 * W2, C2)
        native.nhwc_to_nchw(a.tokens(), nchw, a.C)

        def run() -> None:
            y = node(nchw)
            assert tuple(y.shape) == tuple(res.shape), f"fallback node {cname(node)} produced {tuple(y.shape)}, planned {tuple(res.shape)}"
            res.copy_(y)

        self.python(run, f"torch:{cname(node)}")
        oa = Act(out, a.B, H2, W2)
        native.nchw_to_nhwc(res, oa.tokens())
        self.stats["fallback_nodes"].append(what or cname(node))
        return oa
