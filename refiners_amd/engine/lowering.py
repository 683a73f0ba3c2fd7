"""Lowering of a fluxion UNet tree (with whatever adapters are injected in it) to a flat program of MI355X kernels.

The Chain tree stays the user-facing model (weights, inject/eject, scales).  `Lowering` walks it ONCE, pattern-matches
the sub-trees of SURVEY.md section 8(a) and emits native launches (through refiners_amd.native, in recording mode) into
two lists:

  prologue : everything that depends only on the prompt-side inputs (text K / V^T projections incl. their LoRAs,
             IP-Adapter K' / V'^T, TextTimeEmbedding, ControlLora ConditionEncoder) -- the reference recomputes these on
             every step although they are constant over the 50 steps;
  step     : everything that depends on the latents / timestep.

Data layout: every activation is token-major ("NHWC"): a 2-D view [B*H*W, C] whose rows are pixels == tokens, so the
NCHW <-> (B, L, C) transposes around each transformer (cross_attention.py:120-144) and the head split / merge copies
(attentions.py:177-202) do not exist here.  Skip tensors are concatenated by one streaming kernel; everything else is
fused into GEMM prologues / epilogues (see include/mi355x_refiners.h for the kernels).

What is matched (anything else inside a UNet stage falls back to the node's own torch forward on an NCHW view):
  Linear-like  = fl.Linear | LoraAdapter(fl.Linear, LinearLora...)              -> one GEMM (+ one skinny GEMM per
                 distinct input when LoRAs are present; the up-projections ride as an extra K segment)
  Conv-like    = fl.Conv2d | LoraAdapter(fl.Conv2d, Conv2dLora...) | RangeAdapter2d(Conv-like)   -> implicit GEMM
  SDPA-like    = ScaledDotProductAttention | Sum(SDPA, ImageCrossAttention)     -> one flash kernel, 1 or 2 KV streams
  ResidualBlock, CrossAttentionBlock2d (Linear or Conv2d projections), CrossAttentionBlock, Downsample, Upsample,
  ResidualAccumulator / ResidualConcatenator, TimestepEncoder (SDXL and SD1.5), ControlLora / ZeroConvolution.
The matcher works on class NAMES, so it accepts trees built from refiners_amd.fluxion as well as from refiners itself.
"""
from __future__ import annotations

import os

import math
from dataclasses import dataclass, field
from typing import Any, Callable, Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .. import native

from .packing import Act, CatAct, ConvSpec, LinSpec, LoraPack, PackCache, Pool, Unsupported, _expect, cname, isa, kids, launches  # noqa: F401


class Lowering:
    def __init__(self, device: torch.device, dtype: torch.dtype, cache: Optional[PackCache] = None, lora_mode: str = "fused") -> None:
        # lora_mode: "fused"  = LoraAdapter semantics kept at run time: one skinny down-projection launch per adapted
        #                       site, up-projections as an extra K segment of the parent GEMM (exactly the reference's sum);
        #            "merged" = W' = W + sum_i s_i B_i A_i is formed once per (weights, scales) in float32 and rounded to
        #                       the compute dtype: an adapted Linear / Conv2d costs exactly ONE launch and zero extra
        #                       FLOPs.  Scales stay live: changing them re-lowers and re-merges the touched sites only.
        assert lora_mode in ("fused", "merged")
        self.lora_mode = lora_mode
        self.kblock_policy = int(os.environ.get("REFINERS_AMD_KBLOCK", "2"))
        # LayerNorm folded into the GEMM that consumes it (row statistics from the producing GEMM's epilogue) and the three
        # projections of a self-attention as ONE launch (V stored transposed): REFINERS_AMD_LN_FUSE / REFINERS_AMD_QKV_MERGE = 0 switch
        # them off (A/B runs; the unfused kernels stay in the library)
        self.ln_fuse = os.environ.get("REFINERS_AMD_LN_FUSE", "1") != "0"
        self.qkv_merge = os.environ.get("REFINERS_AMD_QKV_MERGE", "1") != "0"
        # lora_mode="fused": the LoRA down / up projections run INSIDE the parent GEMM's / conv's launch (stacked rank <= 128: producer
        # workgroups at the head of the grid, see gemm_kernel.cuh); 0 = the older skinny-GEMM + extra-K-segment pair of launches
        self.lora_inlaunch = os.environ.get("REFINERS_AMD_LORA_INLAUNCH", "1") != "0"
        # GroupNorm statistics from the epilogue of the convolution / GEMM that produces the normalised tensor (mi355x_gemm_args.colstats_out):
        # the statistics pass over that tensor disappears (2 launches per GroupNorm instead of 3); 0 = always the three-kernel GroupNorm
        self.gn_stats = os.environ.get("REFINERS_AMD_GN_STATS", "1") != "0"
        self._cs_buf: list[Tensor] = []
        self._lsync: Any = None          # native.LoraSync: the epoch word every program of this lowering bumps once per replay
        self._sk = native.StreamK(device)  # scratch of this lowering's stream-K launches (tile 8): allocated by the first one, shared by all (one stream)
        self._bumped: set[int] = set()   # id() of the op lists that already start with the bump
        self.device, self.dtype = device, dtype
        self.es = 4 if dtype == torch.float32 else 2
        self.kblk = 128 // self.es  # K granularity of the GEMM kernel (one 128-byte block)
        self.cache = cache or PackCache()
        self.cache.build_device = device
        # two arenas: prologue results must survive across steps, so they never share storage with step temporaries
        self.step_pool = Pool(device, dtype)
        self.prologue_pool = Pool(device, dtype)
        self.pool = self.step_pool
        self.prologue: list = []
        self.step: list = []
        self._target = self.step
        self.stats = {"fallback_nodes": [], "lora_sites": 0, "ip_sites": 0}

    # -- recording targets ---------------------------------------------------------------------------------
    class _Section:
        def __init__(self, low: "Lowering", ops: list) -> None:
            self.low, self.ops = low, ops

        def __enter__(self) -> None:
            self.saved = (self.low._target, self.low.pool)
            self.low._target = self.ops
            self.low.pool = self.low.prologue_pool if self.ops is self.low.prologue else self.low.step_pool
            self.rec = native.recording(self.ops)
            self.rec.__enter__()
            self.sk_saved = native.set_streamk(self.low._sk)

        def __exit__(self, *exc: object) -> None:
            native.set_streamk(self.sk_saved)
            self.rec.__exit__(*exc)
            self.low._target, self.low.pool = self.saved

    def in_prologue(self) -> "Lowering._Section":
        return Lowering._Section(self, self.prologue)

    def in_step(self) -> "Lowering._Section":
        return Lowering._Section(self, self.step)

    def python(self, fn: Callable[[], None], what: str) -> None:
        self._target.append((None, fn, what, ()))

    # -- weight access ---------------------------------------------------------------------------------------
    def cvt(self, t: Tensor) -> Tensor:
        """The leaf's own storage when it already is a contiguous tensor of the compute dtype on the device."""
        if t.device == self.device and t.dtype == self.dtype and t.is_contiguous() and t.data_ptr() % 16 == 0:
            return t.detach()
        return t.detach().to(device=self.device, dtype=self.dtype).contiguous()

    def _w(self, t: Optional[Tensor], tag: str = "cvt") -> Optional[Tensor]:
        if t is None:
            return None
        return self.cache.get((tag,) + PackCache.ident(t), lambda: self.cvt(t))

    def _unwrap_lora(self, node: Any, leaf_cls: str) -> tuple[Any, list[Any]]:
        """node = leaf | LoraAdapter(leaf, lora...) -> (leaf, [lora...])."""
        if isa(node, "LoraAdapter"):
            ch = kids(node)
            _expect(len(ch) >= 1 and isa(ch[0], leaf_cls), f"LoraAdapter target is {cname(ch[0]) if ch else None}, wanted {leaf_cls}")
            loras = ch[1:]
            for lr in loras:
                c = kids(lr)
                _expect(isa(lr, "Lora") and len(c) == 3 and isa(c[2], "Multiply") and c[2].bias == 0.0, "unexpected Lora layout")
            return ch[0], loras
        _expect(isa(node, leaf_cls), f"expected {leaf_cls}-like node, got {cname(node)}")
        return node, []

    def _lora_pack_linear(self, loras: list[Any], n_out: int, k_in: int, row_perm: Optional[Tensor]) -> Optional[LoraPack]:
        if not loras:
            return None
        downs = [kids(lr)[0].weight for lr in loras]
        ups = [kids(lr)[1].weight for lr in loras]
        scales = tuple(float(kids(lr)[2].scale) for lr in loras)
        key = ("lora",) + PackCache.ident(*downs, *ups) + scales + (None if row_perm is None else "geglu",)

        def make() -> LoraPack:
            rt = sum(d.shape[0] for d in downs)
            rpad = (rt + self.kblk - 1) // self.kblk * self.kblk
            a = torch.zeros(rpad, k_in, device=self.device, dtype=self.dtype)
            bs = torch.zeros(n_out, rpad, device=self.device, dtype=self.dtype)
            o = 0
            for d, u, s in zip(downs, ups, scales):
                r = d.shape[0]
                _expect(tuple(d.shape) == (r, k_in) and tuple(u.shape) == (n_out, r), "LoRA shape does not match its target")
                a[o : o + r] = d.detach().to(device=self.device, dtype=self.dtype)
                bs[:, o : o + r] = (u.detach().to(device=self.device, dtype=torch.float32) * s).to(self.dtype)
                o += r
            if row_perm is not None:
                bs = bs[row_perm].contiguous()
            pack = LoraPack(a, bs)
            R = native.lora_rank(rt)
            if R and self.device.type != "meta":  # fits mi355x_gemm's in-launch LoRA: ONE kernel per adapted Linear
                ar = torch.zeros(R, k_in, device=self.device, dtype=self.dtype)
                ar[: min(R, rpad)] = a[: min(R, rpad)]
                br = torch.zeros(n_out, R, device=self.device, dtype=self.dtype)
                br[:, : min(R, rpad)] = bs[:, : min(R, rpad)]
                pack.a_kb = native.KBlocked(ar)
                pack.bs_r = br
            return pack

        self.stats["lora_sites"] += 1
        return self.cache.get(key, make)

    def linear_spec(self, node: Any, geglu: bool = False) -> LinSpec:
        if isa(node, "Conv2d") and not isa(node, "LoraAdapter"):  # 1x1 conv used as a token projection (SD1.5)
            _expect(node.kernel_size == (1, 1) and node.stride == (1, 1), "only 1x1 convs can act as Linear")
            w = self.cache.get(("conv1x1",) + PackCache.ident(node.weight), lambda: self.cvt(node.weight.detach().reshape(node.out_channels, node.in_channels)))
            return LinSpec(w, self._w(node.bias))
        leaf, loras = self._unwrap_lora(node, "Linear")
        w, b = leaf.weight, leaf.bias
        n_out, k_in = w.shape
        _expect(k_in % self.kblk == 0, f"Linear in_features {k_in} is not a multiple of {self.kblk}")
        perm = None
        if geglu:
            _expect(n_out % 64 == 0, "GEGLU width must be a multiple of 64")
            perm = self.cache.get(("geglu_idx", n_out), lambda: native.geglu_pack_index(n_out // 2, device=self.device))
        if loras and self.lora_mode == "merged":
            downs = [kids(lr)[0].weight for lr in loras]
            ups = [kids(lr)[1].weight for lr in loras]
            scales = tuple(float(kids(lr)[2].scale) for lr in loras)

            def merge() -> Tensor:
                acc = w.detach().to(device=self.device, dtype=torch.float32).clone()
                for d, u, sc in zip(downs, ups, scales):
                    _expect(tuple(d.shape)[1] == k_in and tuple(u.shape)[0] == n_out, "LoRA shape does not match its target")
                    acc = self._mm(u.detach().to(self.device, torch.float32) * sc, d.detach().to(self.device, torch.float32).t(), res=acc)  # acc + s B A
                acc = acc.to(self.dtype)
                return acc[perm].contiguous() if perm is not None else acc.contiguous()

            wm = self.cache.get(("merged", geglu) + PackCache.ident(w, *downs, *ups) + scales, merge)
            self.stats["lora_sites"] += 1
            bp = None if b is None else (self.cache.get(("geglu_b",) + PackCache.ident(b), lambda: self.cvt(b)[perm].contiguous()) if geglu else self._w(b))
            return LinSpec(wm, bp, None, geglu=geglu)
        if geglu:
            wp = self.cache.get(("geglu_w",) + PackCache.ident(w), lambda: self.cvt(w)[perm].contiguous())
            bp = None if b is None else self.cache.get(("geglu_b",) + PackCache.ident(b), lambda: self.cvt(b)[perm].contiguous())
            return LinSpec(wp, bp, self._lora_pack_linear(loras, n_out, k_in, perm), geglu=True)
        return LinSpec(self._w(w), self._w(b), self._lora_pack_linear(loras, n_out, k_in, None))

    def conv_spec(self, node: Any, asym: bool = False) -> ConvSpec:
        time = None
        if isa(node, "RangeAdapter2d"):
            ch = kids(node)
            _expect(len(ch) == 2 and isa(ch[1], "Chain"), "unexpected RangeAdapter2d layout")
            tc = kids(ch[1])
            _expect(len(tc) == 4 and isa(tc[0], "UseContext") and isa(tc[1], "SiLU") and isa(tc[3], "Reshape"), "unexpected RangeAdapter2d time branch")
            _expect(tc[0].context == "range_adapter", "RangeAdapter2d reads an unexpected context")
            time = (tc[0].key, self.linear_spec(tc[2]))
            node = ch[0]
        leaf, loras = self._unwrap_lora(node, "Conv2d")
        w = leaf.weight
        o, i, kh, kw = w.shape
        _expect(kh == kw and kh in (1, 3), f"conv kernel {kh}x{kw} not supported")
        _expect(leaf.stride[0] == leaf.stride[1] and leaf.stride[0] in (1, 2), "conv stride not supported")
        pad = leaf.padding if isinstance(leaf.padding, tuple) else (leaf.padding, leaf.padding)
        _expect(tuple(pad) == ((0, 0) if asym else (kh // 2, kh // 2)), "only 'same'-style padding k//2 (or Downsample's explicit bottom/right pad) is supported")
        _expect(leaf.groups == 1 and tuple(leaf.dilation) == (1, 1), "grouped / dilated conv not supported")
        _expect((i * self.es) % 128 == 0, f"conv in_channels {i} not 128-byte aligned")
        wp = self.cache.get(("convw",) + PackCache.ident(w), lambda: native.pack_conv_weight(self.cvt(w)))
        lora = None
        def _pad2(c: Any) -> tuple:
            return tuple(c.padding) if isinstance(c.padding, (tuple, list)) else (c.padding, c.padding)

        def mergeable(lr: Any) -> bool:
            # B.A folds into the target's weights only when the down conv sees the input exactly like the target does and the
            # up conv is a plain per-pixel 1x1 (same checks as the run-time K-segment path below; anything else keeps that path)
            d, u = kids(lr)[0], kids(lr)[1]
            return (u.kernel_size == (1, 1) and tuple(u.stride) == (1, 1) and _pad2(u) == (0, 0) and d.kernel_size == (kh, kh)
                    and tuple(d.stride) == tuple(leaf.stride) and _pad2(d) == _pad2(leaf) and tuple(d.dilation) == (1, 1) and d.groups == 1 and u.groups == 1
                    and d.weight.shape[1] == i and u.weight.shape[0] == o and u.weight.shape[1] == d.weight.shape[0])

        if loras and self.lora_mode == "merged" and all(mergeable(lr) for lr in loras):
            dws = [kids(lr)[0].weight for lr in loras]
            uws = [kids(lr)[1].weight for lr in loras]
            scs = tuple(float(kids(lr)[2].scale) for lr in loras)

            def merge_conv() -> Tensor:
                acc = w.detach().to(device=self.device, dtype=torch.float32).clone()
                for d, u, sc in zip(dws, uws, scs):  # (B A)[o, i, ky, kx] = sum_r B[o, r] A[r, i, ky, kx]
                    d32 = d.detach().to(self.device, torch.float32)
                    acc = self._mm(u.detach().to(self.device, torch.float32)[:, :, 0, 0] * sc, d32.reshape(d32.shape[0], -1).t(), res=acc.reshape(o, -1)).reshape(acc.shape)
                return native.pack_conv_weight(acc.to(self.dtype))

            wp = self.cache.get(("merged_conv",) + PackCache.ident(w, *dws, *uws) + scs, merge_conv)
            self.stats["lora_sites"] += 1
            loras = []
        if loras:
            downs = [kids(lr)[0] for lr in loras]
            ups = [kids(lr)[1] for lr in loras]
            kd, ku = downs[0].kernel_size[0], ups[0].kernel_size[0]
            _expect(all(d.kernel_size == (kd, kd) and tuple(d.stride) == tuple(leaf.stride) for d in downs), "Conv2dLora down convs differ")
            _expect(all(u.kernel_size == (ku, ku) and tuple(u.stride) == (1, 1) for u in ups), "Conv2dLora up convs differ")
            _expect(kd in (1, 3) and ku in (1, 3), "Conv2dLora kernel not supported")
            scales = tuple(float(kids(lr)[2].scale) for lr in loras)
            key = ("convlora",) + PackCache.ident(*[d.weight for d in downs], *[u.weight for u in ups]) + scales

            def make() -> LoraPack:
                rt = sum(d.weight.shape[0] for d in downs)
                rpad = (rt + self.kblk - 1) // self.kblk * self.kblk
                a = torch.zeros(rpad, kd * kd * i, device=self.device, dtype=self.dtype)
                bs4 = torch.zeros(o, rpad, ku, ku, device=self.device, dtype=torch.float32)
                off = 0
                for d, u, s in zip(downs, ups, scales):
                    r = d.weight.shape[0]
                    a[off : off + r] = native.pack_conv_weight(d.weight.detach().to(device=self.device, dtype=self.dtype))
                    bs4[:, off : off + r] = u.weight.detach().to(device=self.device, dtype=torch.float32) * s
                    off += r
                pack = LoraPack(a, native.pack_conv_weight(bs4.to(self.dtype)), conv=(kd, ku, leaf.stride[0]))
                R = native.lora_rank(rt)
                if ku == 1 and kd == kh and R and self.device.type != "meta" and all(_pad2(d) == _pad2(leaf) for d in downs):
                    # Conv2dLora inside the parent conv's launch: down conv = the parent's kernel size / stride / padding, 1x1 up conv
                    ar = torch.zeros(R, kd * kd * i, device=self.device, dtype=self.dtype)
                    ar[: min(R, rpad)] = a[: min(R, rpad)]
                    br = torch.zeros(o, R, device=self.device, dtype=self.dtype)
                    br[:, : min(R, rpad)] = bs4[:, : min(R, rpad), 0, 0].to(self.dtype)
                    pack.a_kb = native.KBlocked(ar)
                    pack.bs_r = br
                return pack

            lora = self.cache.get(key, make)
            self.stats["lora_sites"] += 1
        return ConvSpec(wp, self._w(leaf.bias), i, o, kh, leaf.stride[0], lora, time, asym)

    # -- emitters: GEMM family -------------------------------------------------------------------------------
    def _mm(self, x: Tensor, w: Tensor, res: Optional[Tensor] = None) -> Tensor:
        """x [M, K] @ w [N, K]^T (+ res), float32: weight preparation at lowering time, on the library's own f32 MFMA kernel where there is
        a GPU (native.matmul_f32), plain torch on the meta / CPU devices of the dry-lowering tests."""
        if self.device.type == "cuda":
            return native.matmul_f32(x.contiguous(), w.contiguous(), None if res is None else res.contiguous())
        y = x @ w.t()
        return y if res is None else y + res

    def lora_sync(self, groups: int, M: int, R: int) -> tuple:
        """(t scratch, flags, LoraSync) of one in-launch LoRA site (native._lora_fill); the first site of a program puts the epoch bump at
        the program's head.  The scratch comes from the pool (give it back with pool.put once the launch is recorded); flags are the site's own."""
        if self._lsync is None:
            self._lsync = native.LoraSync(self.device)
        if id(self._target) not in self._bumped:
            self._target.insert(0, self._lsync.bump_op())
            self._bumped.add(id(self._target))
        return self.pool.get(native.lora_scratch_rows(groups, M, R, self.dtype), R), self._lsync.flags(groups, M), self._lsync

    def handover_pending(self) -> list:
        """Device-side "an in-launch hand-over was lost" indicators of this lowering's programs (0-d bool tensors; no host synchronisation): the in-launch
        LoRA's error words (a tile gave up waiting for t = x A^T) and the stream-K scratch's (a partial tile never arrived).  See CompiledUNet.check_handovers."""
        out = []
        for st in (self._lsync, self._sk):
            bad = st.pending() if st is not None else None
            if bad is not None:
                out.append(bad)
        return out

    def handover_raise(self) -> None:
        for st in (self._lsync, self._sk):
            if st is not None:
                st.check()

    def colstats_for(self, M: int, N: int, HW: int) -> Any:
        """The buffer for the column statistics of an [M, N] image tensor about to be produced, or None when its consumer could not use them
        (32-pixel blocks must not straddle samples).  One buffer per PRODUCER (they total ~60 MB for the SDXL step, nothing next to 288 GB):
        the statistics of a tensor stay valid for as long as the tensor does -- a skip tensor is normalised (as one half of a
        ResidualConcatenator's output) long after the next tensor of its shape has been written."""
        if not self.gn_stats or HW % 32 or N % 16 or self.device.type == "meta" or self._target is self.prologue:
            return None  # (prologue launches -- the ConditionEncoder's convolutions at up to 1024 x 1024 pixels -- feed no GroupNorm: round-4 advisor)
        cs = torch.empty(native.colstats_shape(M, N), device=self.device, dtype=torch.float32)
        self._cs_buf.append(cs)
        return cs

    def lora_down(self, x: Tensor, lora: LoraPack) -> Tensor:
        t = self.pool.get(x.shape[0], lora.a_cat.shape[0])
        native.gemm([(x, lora.a_cat)], t)
        return t

    def kblocked(self, w: Tensor) -> Any:
        """The K-blocked copy of a weight matrix whose rows are long enough for the re-layout to pay (native.KBlocked): row
        strides of 5 KB and more halve the kernel's global -> LDS streaming rate; a 3x3 convolution's packed weights have
        23-46 KB rows.  Policy REFINERS_AMD_KBLOCK: 0 = never, 1 = rows of >= 5120 bytes, 2 (default) = every weight (SDXL step 32.1 -> 31.1 -> 30.3 ms on one box)."""
        row_bytes = w.shape[1] * w.element_size()
        if self.kblock_policy == 0 or (self.kblock_policy == 1 and row_bytes < 5120) or w.device.type == "meta" or not w.is_contiguous():
            return w
        return self.cache.get(("kblocked",) + PackCache.ident(w), lambda: native.KBlocked(w))

    def linear(self, x: Any, spec: LinSpec, *, res: Optional[Tensor] = None, out: Optional[Tensor] = None,
               rows: Optional[int] = None, lora_t: Optional[Tensor] = None, gelu: bool = False, out_kblocked: bool = False,
               ln: Optional[tuple] = None, stats_out: Optional[Tensor] = None, colstats: Any = None) -> Tensor:
        """out[M, N(/2 if geglu)] = epi(x W^T + b (+ LoRA) (+ res)).  `colstats` = colstats_for(...): also write the output's GroupNorm statistics.  `lora_t` lets callers share one down-projection
        launch between Linears that read the same x.  `ln` = (stats, LayerNorm node): x is the UN-normalised tensor and the
        LayerNorm is applied inside this launch (ln_fold); `stats_out`: also write the output rows' statistics."""
        M = x.shape[0]
        n_cols = spec.N // 2 if spec.geglu else spec.N
        if out is None:
            out = self.pool.get(M, n_cols)
        cso = colstats
        if ln is not None:
            stats, node = ln
            wl, ls, lc = self.ln_fold(spec, node)
            lo = sy = None
            if spec.lora is not None:  # LayerNorm AND the LoRAs inside the parent launch
                al, als, alc = self.ln_fold_lora(spec.lora, node)
                lo = ([(0, al)], spec.lora.bs_r, als, alc)
                sy = self.lora_sync(1, M, spec.lora.R)
            native.gemm([(x, self.kblocked(wl))], out, res=res, geglu=spec.geglu, gelu=gelu, out_kblocked=out_kblocked, ln=(stats, ls, lc, float(node.eps)),
                        stats_out=stats_out, lora=lo, lora_sync=sy, colstats_out=cso)
            if sy is not None:
                self.pool.put(sy[0])
            return out
        if spec.lora is not None and spec.lora.a_kb is not None and self.lora_inlaunch:
            # LoraAdapter = Sum(target, loras) as ONE launch: producer workgroups compute x A_cat^T once per row block, the up-projections are the tiles' last K steps
            sy = self.lora_sync(1, M, spec.lora.R)
            native.gemm([(x, self.kblocked(spec.w))], out, bias=spec.b, res=res, geglu=spec.geglu, gelu=gelu, out_kblocked=out_kblocked, stats_out=stats_out,
                        lora=([(0, spec.lora.a_kb)], spec.lora.bs_r), lora_sync=sy, colstats_out=cso)
            self.pool.put(sy[0])
            return out
        segs = [(x, self.kblocked(spec.w))]
        t = None
        if spec.lora is not None:
            t = lora_t if lora_t is not None else self.lora_down(x, spec.lora)
            segs.append((t, spec.lora.bs_cat))
        native.gemm(segs, out, bias=spec.b, res=res, geglu=spec.geglu, gelu=gelu, out_kblocked=out_kblocked, stats_out=stats_out, colstats_out=cso)
        if t is not None and lora_t is None:
            self.pool.put(t)
        return out

    # -- LayerNorm folded into its consumer ------------------------------------------------------------------------------------
    def ln_fold(self, spec: LinSpec, node: Any) -> tuple[Tensor, Tensor, Tensor]:
        """(W', s, c) of `Linear(LayerNorm(x))` for mi355x_gemm's ln_* arguments: W' = W diag(gamma) in the compute dtype,
        s[n] = sum_k W'[n][k] (of the ROUNDED W', which is what the matrix cores multiply), c = W beta + b, both float32.
        spec.w may already be row-permuted (GEGLU) or a merged LoRA weight: everything here is per row."""
        _expect(isa(node, "LayerNorm") and tuple(node.normalized_shape) == (spec.K,) and node.weight is not None, "LayerNorm shape mismatch")
        key = ("ln_fold",) + PackCache.ident(spec.w, spec.b, node.weight, node.bias)

        def make() -> tuple[Tensor, Tensor, Tensor]:
            w32 = spec.w.detach().to(self.device, torch.float32)
            g32 = node.weight.detach().to(self.device, torch.float32)
            wl = (w32 * g32.unsqueeze(0)).to(self.dtype).contiguous()
            ls = wl.to(torch.float32).sum(dim=1).contiguous()
            lc = torch.zeros(w32.shape[0], device=self.device, dtype=torch.float32)
            if node.bias is not None:
                lc = self._mm(w32, node.bias.detach().to(self.device, torch.float32).unsqueeze(0)).reshape(-1)
            if spec.b is not None:
                lc = lc + spec.b.detach().to(self.device, torch.float32)
            return wl, ls, lc.contiguous()

        return self.cache.get(key, make)

    def ln_fusable(self, stats: Optional[Tensor], *specs: LinSpec) -> bool:
        """LayerNorm can ride in the consumer launch when every consumer is a plain Linear or carries its LoRAs in-launch."""
        ok_lora = lambda sp: sp.lora is None or (sp.lora.a_kb is not None and self.lora_inlaunch)  # noqa: E731
        return (stats is not None and self.ln_fuse and self.device.type != "meta" and all(ok_lora(sp) and sp.N % 64 == 0 and sp.K % 64 == 0 for sp in specs))

    def ln_fold_lora(self, lora: LoraPack, node: Any) -> tuple[Any, Tensor, Tensor]:
        """(A' K-blocked, sA [R], cA [R]) of the stacked down-projections behind a folded LayerNorm (see ln_fold)."""
        a = lora.a_kb.dense()
        key = ("ln_fold_lora",) + PackCache.ident(lora.a_cat, node.weight, node.bias)

        def make() -> tuple[Any, Tensor, Tensor]:
            a32 = a.detach().to(self.device, torch.float32)
            g32 = node.weight.detach().to(self.device, torch.float32)
            al = (a32 * g32.unsqueeze(0)).to(self.dtype).contiguous()
            ls = al.to(torch.float32).sum(dim=1).contiguous()
            lc = self._mm(a32, node.bias.detach().to(self.device, torch.float32).unsqueeze(0)).reshape(-1).contiguous() if node.bias is not None else torch.zeros_like(ls)
            return native.KBlocked(al), ls, lc

        return self.cache.get(key, make)

    def row_stats(self, M: int, C: int) -> Optional[Tensor]:
        """The [C / 32][M][2] float32 buffer a producing GEMM's epilogue fills with per-row (mean, M2) partials.  One per (M, C):
        the program is sequential, every consumer of a set of statistics runs before the next producer of that size."""
        if not self.ln_fuse or C % 64 or self.device.type == "meta":
            return None
        store = self.__dict__.setdefault("_row_stats", {})
        if (M, C) not in store:
            store[(M, C)] = torch.empty(C // 32, M, 2, device=self.device, dtype=torch.float32)
        return store[(M, C)]

    def linear_T(self, x: Tensor, spec: LinSpec, out_t: Tensor) -> Tensor:
        """out_t[N, M] = W x^T (+ LoRA): the V^T layout mi355x_attention consumes (operands swapped, no bias)."""
        _expect(spec.b is None, "transposed projection with bias is not supported")
        if spec.lora is not None and spec.lora.a_kb is not None and self.lora_inlaunch and out_t.stride(1) == 1:
            sy = self.lora_sync(1, x.shape[0], spec.lora.R)
            native.gemm([(x, self.kblocked(spec.w))], None, out_t=out_t, nt_begin=0, lora=([(0, spec.lora.a_kb)], spec.lora.bs_r), lora_sync=sy)
            self.pool.put(sy[0])
            return out_t
        segs = [(self.kblocked(spec.w), x)]
        t = None
        if spec.lora is not None:
            t = self.lora_down(x, spec.lora)
            segs.append((spec.lora.bs_cat, t))
        native.gemm(segs, out_t, weight_operand="x")
        self.pool.put(t)
        return out_t

    def conv(self, a: Act, spec: ConvSpec, *, rowbias: Optional[Tensor] = None, res: Optional[Tensor] = None, ups: int = 1,
             shortcut: Optional[tuple[Act, ConvSpec]] = None, bias: Optional[Tensor] = "spec") -> Act:  # type: ignore[assignment]
        """Implicit-GEMM convolution of a token-major image; optional fused extras:
        rowbias = per-sample channel bias (time embedding), res = residual rows, ups = nearest upsampling of the input,
        shortcut = a second (1x1) convolution of another image accumulated into the same output tile."""
        H, W = a.H * ups, a.W * ups
        OH, OW = (H + spec.stride - 1) // spec.stride, (W + spec.stride - 1) // spec.stride
        out = self.pool.get(a.B * OH * OW, spec.cout)
        segs = [(a.image(), self.kblocked(spec.w), spec.ksize, spec.stride, ups, int(spec.asym))]
        t = None
        lo = sy = None
        if spec.lora is not None and spec.lora.a_kb is not None and self.lora_inlaunch:
            lo = ([(0, spec.lora.a_kb)], spec.lora.bs_r)  # Conv2dLora inside this launch
        elif spec.lora is not None:
            kd, ku, st = spec.lora.conv  # type: ignore[misc]
            t = self.pool.get(a.B * OH * OW, spec.lora.a_cat.shape[0])
            native.conv_gemm([(a.image(), spec.lora.a_cat, kd, st, ups, int(spec.asym))], t, a.B, OH, OW)
            segs.append((Act(t, a.B, OH, OW).image(), spec.lora.bs_cat, ku, 1, 1))
        b = spec.b if isinstance(bias, str) else bias
        if shortcut is not None:
            sa, sspec = shortcut
            _expect(sspec.lora is None and sspec.ksize == 1 and sspec.stride == 1, "unsupported shortcut convolution")
            if isinstance(sa, CatAct):  # 1x1 convolution of a concatenation = two K segments, one per part, against the matching weight columns
                c1 = sa.a.C
                w1, w2 = self.cache.get(("sc_split", c1) + PackCache.ident(sspec.w), lambda: (sspec.w[:, :c1].contiguous(), sspec.w[:, c1:].contiguous()))
                segs.append((sa.a.image(), w1, 1, 1, 1))
                segs.append((sa.b.image(), w2, 1, 1, 1))
            else:
                segs.append((sa.image(), sspec.w, 1, 1, 1))
        _expect(len(segs) <= native.MAX_SEG, "too many K segments for one conv launch")
        # few output tiles but a very long K (the 32 x 32-resolution convolutions of a CFG pair: 160 tiles, K = 11 520):
        # split K three ways over 128 x 128 tiles (measured 169 -> 108 us, profiles/r01_h_probe_splitk.log); the float32
        # partials are summed in a fixed order by a second launch, so the result stays bit-reproducible.
        M_out = a.B * OH * OW
        tiles128 = ((M_out + 127) // 128) * ((spec.cout + 127) // 128)
        total_kb = sum(sg[1].shape[1] for sg in segs) * self.es // 128
        tile, ksplit, ws = 0, 1, None
        if tiles128 <= 192 and total_kb >= 96:
            tile, ksplit = 1, 3
            ws = self.splitk_workspace(ksplit * M_out * spec.cout)
        if lo is not None:
            sy = self.lora_sync(1, M_out, spec.lora.R)
        cs = self.colstats_for(M_out, spec.cout, OH * OW)  # most convolution outputs of a UNet are normalised next (ResidualBlock, unet.py:6-51)
        native.conv_gemm(segs, out, a.B, OH, OW, bias=b, rowbias=rowbias, rows_per_group=OH * OW, res=res, tile=tile, ksplit=ksplit, ws=ws, lora=lo, lora_sync=sy,
                         colstats_out=cs, table_may_replace_split=True)
        if sy is not None:
            self.pool.put(sy[0])
        self.pool.put(t)
        return Act(out, a.B, OH, OW, cs)

    def splitk_workspace(self, floats: int) -> Tensor:
        """One float32 scratch per size class, shared by every split-K launch of the (sequential) program."""
        store = self.__dict__.setdefault("_splitk_ws", {})
        if floats not in store:
            store[floats] = torch.empty(floats, device=self.device, dtype=torch.float32)
        return store[floats]

    # -- emitters: norms / glue ------------------------------------------------------------------------------
    def groupnorm(self, a: Any, gn: Any, silu: bool) -> Act:
        """GroupNorm (+ SiLU) of an activation, or of a never-materialised concatenation (CatAct: two sources).  Statistics from the producing
        launch(es) where every part carries them."""
        _expect(isa(gn, "GroupNorm") and gn.num_channels == a.C, "GroupNorm channel mismatch")
        out = self.pool.get(a.M, a.C)
        oa = Act(out, a.B, a.H, a.W)
        if isinstance(a, CatAct):
            both = a.a.cs is not None and a.b.cs is not None and a.HW % 32 == 0
            if both:
                self.stats["gn_from_producer"] = self.stats.get("gn_from_producer", 0) + 1
            native.groupnorm_nhwc(a.a.tokens(), self._w(gn.weight), self._w(gn.bias), gn.num_groups, gn.eps, silu, oa.tokens(), x2=a.b.tokens(),
                                  colstats=a.a.cs if both else None, colstats2=a.b.cs if both else None)
            return oa
        cs = a.cs if a.HW % 32 == 0 else None
        if cs is not None:
            self.stats["gn_from_producer"] = self.stats.get("gn_from_producer", 0) + 1
        native.groupnorm_nhwc(a.tokens(), self._w(gn.weight), self._w(gn.bias), gn.num_groups, gn.eps, silu, oa.tokens(), colstats=cs)
        return oa

    def materialise(self, a: Any) -> Act:
        """The real tensor of a CatAct (one streaming concat launch), for consumers that cannot read two sources."""
        if not isinstance(a, CatAct):
            return a
        cat = self.pool.get(a.M, a.C)
        native.concat2(a.a.t, a.b.t, cat)
        return Act(cat, a.B, a.H, a.W)

    def layernorm(self, x: Tensor, ln: Any) -> Tensor:
        _expect(isa(ln, "LayerNorm") and tuple(ln.normalized_shape) == (x.shape[1],), "LayerNorm shape mismatch")
        out = self.pool.get(x.shape[0], x.shape[1])
        native.layernorm(x, self._w(ln.weight), self._w(ln.bias), ln.eps, out)
        return out

    # -- attention ---------------------------------------------------------------------------------------------
