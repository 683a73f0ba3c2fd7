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


class Unsupported(Exception):
    """A sub-tree does not have the shape this lowering knows; the caller falls back to the unfused path."""


# ------------------------------------------------------------------------------------------------ tree matching helpers
def isa(m: Any, *names: str) -> bool:
    return any(c.__name__ in names for c in type(m).__mro__)


def kids(m: Any) -> list[Any]:
    return list(m._modules.values())


def launches(ops: list) -> int:
    """Number of kernel-launching entries of a recorded program (Python glue such as stream fork / join excluded)."""
    return sum(1 for e in ops if e[0] is not None)


def cname(m: Any) -> str:
    return type(m).__name__


def _expect(cond: bool, what: str) -> None:
    if not cond:
        raise Unsupported(what)


@dataclass
class Act:
    """A token-major activation: `t` is a [B*H*W, C] view with unit channel stride."""

    t: Tensor
    B: int
    H: int
    W: int

    @property
    def C(self) -> int:
        return self.t.shape[1]

    @property
    def M(self) -> int:
        return self.t.shape[0]

    @property
    def HW(self) -> int:
        return self.H * self.W

    def image(self) -> Tensor:
        ld = self.t.stride(0)
        return self.t.as_strided((self.B, self.H, self.W, self.C), (self.HW * ld, self.W * ld, ld, 1))

    def tokens(self) -> Tensor:
        ld = self.t.stride(0)
        return self.t.as_strided((self.B, self.HW, self.C), (self.HW * ld, ld, 1))


@dataclass
class LoraPack:
    a_cat: Tensor  # [rpad, K(...)] stacked down weights, zero padded rows
    bs_cat: Tensor  # [N, rpad] stacked (scale * up) columns
    conv: Optional[tuple[int, int, int]] = None  # (down ksize, up ksize, stride) for Conv2dLora
    a_kb: Any = None  # native.KBlocked of the first R rows of a_cat (R = stacked rank rounded up to 32), when R fits the in-launch path (<= 128)
    bs_r: Optional[Tensor] = None  # [N, R]: for a Conv2dLora with a 1x1 up convolution, the up weights as a matrix

    @property
    def R(self) -> int:
        return 0 if self.bs_r is None else int(self.bs_r.shape[1])


@dataclass
class LinSpec:
    w: Tensor  # [N, K]
    b: Optional[Tensor]
    lora: Optional[LoraPack] = None
    geglu: bool = False

    @property
    def N(self) -> int:
        return self.w.shape[0]

    @property
    def K(self) -> int:
        return self.w.shape[1]


@dataclass
class ConvSpec:
    w: Tensor  # packed [O, k*k*I]
    b: Optional[Tensor]
    cin: int
    cout: int
    ksize: int
    stride: int
    lora: Optional[LoraPack] = None
    time: Optional[tuple[str, LinSpec]] = None  # (context key, Linear(1280 -> cout)) of a RangeAdapter2d
    asym: bool = False  # padding only after the last row / column (fl.Downsample(padding=0))


class Pool:
    """Static device buffers for the program, reused as soon as the emitting code gives them back."""

    def __init__(self, device: torch.device, dtype: torch.dtype) -> None:
        self.device, self.dtype = device, dtype
        self.free_list: dict[int, list[Tensor]] = {}
        self.all: list[Tensor] = []
        self.pinned: set[int] = set()

    def get(self, rows: int, cols: int, dtype: Optional[torch.dtype] = None) -> Tensor:
        dtype = dtype or self.dtype
        n = rows * cols
        if dtype == self.dtype:
            bucket = self.free_list.get(n)
            if bucket:
                return bucket.pop().view(rows, cols)
        t = torch.empty(n, device=self.device, dtype=dtype)
        self.all.append(t)
        return t.view(rows, cols)

    def put(self, t: Optional[Tensor]) -> None:
        if t is None or t.dtype != self.dtype or not t.is_contiguous():
            return
        base = t.reshape(-1)
        if base.data_ptr() in self.pinned:
            return
        self.free_list.setdefault(base.numel(), []).append(base)

    def pin(self, t: Tensor) -> None:
        self.pinned.add(t.data_ptr())

    def bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.all)


class _Src:
    """Key element for one source tensor: compares by (id, _version, data_ptr) and HOLDS the tensor, so that for as long as
    a cache entry exists its sources stay alive and neither their id() nor their storage address can be handed to another
    tensor (a same-shaped LoRA loaded after an eject would otherwise hit the stale merged / packed copy)."""

    __slots__ = ("t", "sig")

    def __init__(self, t: Tensor) -> None:
        self.t = t
        self.sig = (id(t), t._version, t.data_ptr())

    def __hash__(self) -> int:
        return hash(self.sig)

    def __eq__(self, other: object) -> bool:
        return isinstance(other, _Src) and self.sig == other.sig

    def __repr__(self) -> str:
        return f"_Src{self.sig}"


class PackCache:
    """Packed / converted copies of leaf weights, keyed on the identity and version of the source tensors (which the key
    keeps alive, see _Src), so that a re-lowering after inject / eject / scale change only re-packs what actually changed."""

    def __init__(self) -> None:
        self.store: dict[tuple, Any] = {}
        self.hits = 0
        self.used: set[tuple] = set()

    @staticmethod
    def ident(*tensors: Optional[Tensor]) -> tuple:
        return tuple(_Src(t) if t is not None else None for t in tensors)

    def get(self, key: tuple, make: Callable[[], Any]) -> Any:
        self.used.add(key)
        if key in self.store:
            self.hits += 1
            return self.store[key]
        v = make()
        self.store[key] = v
        return v

    def sweep(self) -> None:
        for k in list(self.store):
            if k not in self.used:
                del self.store[k]
        self.used = set()


# ------------------------------------------------------------------------------------------------ the lowering
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
        # independent projections of one attention on a second stream (native.side_branch).  OFF by default: measured on
        # MI355X the forked Q|K / V^T pair makes the SDXL step 2 % SLOWER (29.7 vs 29.0 ms; the join edges cost more than the
        # overlap gains, both GEMMs pull from the same L2s).  Kept as a switch for larger batches / other trees.
        self.side_branches = device.type == "cuda" and os.environ.get("REFINERS_AMD_SIDE_BRANCHES", "0") == "1"
        # LayerNorm folded into the GEMM that consumes it (row statistics from the producing GEMM's epilogue) and the three
        # projections of a self-attention as ONE launch (V stored transposed): REFINERS_AMD_LN_FUSE / REFINERS_AMD_QKV_MERGE = 0 switch
        # them off (A/B runs; the unfused kernels stay in the library)
        self.ln_fuse = os.environ.get("REFINERS_AMD_LN_FUSE", "1") != "0"
        self.qkv_merge = os.environ.get("REFINERS_AMD_QKV_MERGE", "1") != "0"
        # lora_mode="fused": the LoRA down / up projections run INSIDE the parent GEMM's / conv's launch (stacked rank <= 128: producer
        # workgroups at the head of the grid, see gemm_kernel.cuh); 0 = the older skinny-GEMM + extra-K-segment pair of launches
        self.lora_inlaunch = os.environ.get("REFINERS_AMD_LORA_INLAUNCH", "1") != "0"
        self._lsync: Any = None          # native.LoraSync: the epoch word every program of this lowering bumps once per replay
        self._bumped: set[int] = set()   # id() of the op lists that already start with the bump
        # extra elements per row of a self-attention's V^T buffer [C][B L (+ pad)]: with a row stride of exactly B L elements (4 / 16 KB at 1024 / 4096
        # tokens) the 64 rows of a V^T tile sit a power of two apart in memory
        self.vt_pad = int(os.environ.get("REFINERS_AMD_VT_PAD", "0"))
        # EXPERIMENTAL, off: a cross-attention's q-projection and its SDPA over the (short, prologue-resident) text / image keys as ONE launch
        # (mi355x_gemm's xattn epilogue).  Kernel-level parity is tested (tests/kernel_cases.py: xattn_*); measured level with the two launches
        # at a CFG pair and 8 % ahead at 4 images (profiles/r02_w_probe_xattn.log); this switch has not been through the engine tests yet
        self.xattn_fuse = os.environ.get("REFINERS_AMD_XATTN_FUSE", "0") == "1"
        self.device, self.dtype = device, dtype
        self.es = 4 if dtype == torch.float32 else 2
        self.kblk = 128 // self.es  # K granularity of the GEMM kernel (one 128-byte block)
        self.cache = cache or PackCache()
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

        def __exit__(self, *exc: object) -> None:
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
        return self.pool.get(groups * M, R), self._lsync.flags(groups, M), self._lsync

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
               ln: Optional[tuple] = None, stats_out: Optional[Tensor] = None) -> Tensor:
        """out[M, N(/2 if geglu)] = epi(x W^T + b (+ LoRA) (+ res)).  `lora_t` lets callers share one down-projection
        launch between Linears that read the same x.  `ln` = (stats, LayerNorm node): x is the UN-normalised tensor and the
        LayerNorm is applied inside this launch (ln_fold); `stats_out`: also write the output rows' statistics."""
        M = x.shape[0]
        n_cols = spec.N // 2 if spec.geglu else spec.N
        if out is None:
            out = self.pool.get(M, n_cols)
        if ln is not None:
            stats, node = ln
            wl, ls, lc = self.ln_fold(spec, node)
            lo = sy = None
            if spec.lora is not None:  # LayerNorm AND the LoRAs inside the parent launch
                al, als, alc = self.ln_fold_lora(spec.lora, node)
                lo = ([(0, al)], spec.lora.bs_r, als, alc)
                sy = self.lora_sync(1, M, spec.lora.R)
            native.gemm([(x, self.kblocked(wl))], out, res=res, geglu=spec.geglu, gelu=gelu, out_kblocked=out_kblocked, ln=(stats, ls, lc, float(node.eps)),
                        stats_out=stats_out, lora=lo, lora_sync=sy)
            if sy is not None:
                self.pool.put(sy[0])
            return out
        if spec.lora is not None and spec.lora.a_kb is not None and self.lora_inlaunch:
            # LoraAdapter = Sum(target, loras) as ONE launch: producer workgroups compute x A_cat^T once per row block, the up-projections are the tiles' last K steps
            sy = self.lora_sync(1, M, spec.lora.R)
            native.gemm([(x, self.kblocked(spec.w))], out, bias=spec.b, res=res, geglu=spec.geglu, gelu=gelu, out_kblocked=out_kblocked, stats_out=stats_out,
                        lora=([(0, spec.lora.a_kb)], spec.lora.bs_r), lora_sync=sy)
            self.pool.put(sy[0])
            return out
        segs = [(x, self.kblocked(spec.w))]
        t = None
        if spec.lora is not None:
            t = lora_t if lora_t is not None else self.lora_down(x, spec.lora)
            segs.append((t, spec.lora.bs_cat))
        native.gemm(segs, out, bias=spec.b, res=res, geglu=spec.geglu, gelu=gelu, out_kblocked=out_kblocked, stats_out=stats_out)
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
        native.conv_gemm(segs, out, a.B, OH, OW, bias=b, rowbias=rowbias, rows_per_group=OH * OW, res=res, tile=tile, ksplit=ksplit, ws=ws, lora=lo, lora_sync=sy)
        if sy is not None:
            self.pool.put(sy[0])
        self.pool.put(t)
        return Act(out, a.B, OH, OW)

    def splitk_workspace(self, floats: int) -> Tensor:
        """One float32 scratch per size class, shared by every split-K launch of the (sequential) program."""
        store = self.__dict__.setdefault("_splitk_ws", {})
        if floats not in store:
            store[floats] = torch.empty(floats, device=self.device, dtype=torch.float32)
        return store[floats]

    # -- emitters: norms / glue ------------------------------------------------------------------------------
    def groupnorm(self, a: Act, gn: Any, silu: bool) -> Act:
        _expect(isa(gn, "GroupNorm") and gn.num_channels == a.C, "GroupNorm channel mismatch")
        out = self.pool.get(a.M, a.C)
        native.groupnorm_nhwc(a.tokens(), self._w(gn.weight), self._w(gn.bias), gn.num_groups, gn.eps, silu, Act(out, a.B, a.H, a.W).tokens())
        return Act(out, a.B, a.H, a.W)

    def layernorm(self, x: Tensor, ln: Any) -> Tensor:
        _expect(isa(ln, "LayerNorm") and tuple(ln.normalized_shape) == (x.shape[1],), "LayerNorm shape mismatch")
        out = self.pool.get(x.shape[0], x.shape[1])
        native.layernorm(x, self._w(ln.weight), self._w(ln.bias), ln.eps, out)
        return out

    # -- attention ---------------------------------------------------------------------------------------------
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
        qk = q = k = vt = vt_full = None
        if (no_lora or all_inlaunch) and native_path and self.qkv_merge and L % 64 == 0 and C % 128 == 0 and self.device.type != "meta":
            # ONE launch for the three projections: [Wq; Wk; Wv] stacked, Q | K row-major, V stored transposed
            wqkv = LinSpec(self.cache.get(("qkv",) + PackCache.ident(qs.w, ks.w, vs.w), lambda: torch.cat([qs.w, ks.w, vs.w], 0).contiguous()), None)
            qk = self.pool.get(M, 2 * C)
            vt_full = self.pool.get(C, M + self.vt_pad)
            vt = vt_full[:, :M] if self.vt_pad else vt_full
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
            # The V^T projection and the packed Q|K projection read the same h and do not depend on each other; neither fills the
            # chip at a CFG pair's 2048 rows, so V^T may be issued on the side stream (native.side_branch) and joined before attention.
            if fold and (not (native_path and L % 64 == 0) or not no_lora):
                h, lnarg, fold = self.layernorm(x, ln), None, False  # the per-sample / torch V paths (and separate LoRA launches) want a materialised h
            concurrent = native_path and self.side_branches and vs.lora is None and not fold
            if concurrent:
                native.fork()
                with native.side_branch():
                    vt = self._project_vt(h, vs, B, L, C)
            if qs.lora is None and ks.lora is None:
                wqk = LinSpec(self.cache.get(("qk",) + PackCache.ident(qs.w, ks.w), lambda: torch.cat([qs.w, ks.w], 0).contiguous()), None)
                qk = self.linear(h, wqk, ln=lnarg)
                q, k = qk[:, :C], qk[:, C:]
            else:
                q = self.linear(h, qs)
                k = self.linear(h, ks)
            if native_path:
                if concurrent:
                    native.join()
                elif fold:  # V^T = (Wv LN(x)^T): the transposed column group alone (nt_begin = 0)
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
                self.pool.put(vt_full if vt_full is not None else vt)
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
        if (self.xattn_fuse and self.device.type != "meta" and qspec.lora is None and plains[0] is None and self.head_kernel(C // heads) == "flash64"
                and (M // B) % 128 == 0 and C % 128 == 0 and all(lk <= 80 for (_k, _v, lk, _s) in streams) and sum((lk + 15) // 16 for (_k, _v, lk, _s) in streams) <= 6):
            st = []
            for kk, vt, lk, osc in streams:  # the views mi355x_attention takes (see sdpa)
                lkp, lv = kk.shape[0] // B, vt.shape[1] // B
                st.append((kk.as_strided((B, lkp, C), (lkp * kk.stride(0), kk.stride(0), 1)), vt.as_strided((C, B, lv), (vt.stride(0), lv, 1)), lk, osc))
            o = self.pool.get(M, C)
            if self.ln_fusable(stats, qspec):
                wl, ls, lc = self.ln_fold(qspec, ln)
                native.gemm([(x, self.kblocked(wl))], o, ln=(stats, ls, lc, float(ln.eps)), xattn=(st, M // B, None))
            else:
                h = self.layernorm(x, ln)
                native.gemm([(h, self.kblocked(qspec.w))], o, bias=qspec.b, xattn=(st, M // B, None))
                self.pool.put(h)
            self.stats["xattn_fused"] = self.stats.get("xattn_fused", 0) + 1
            self.linear(o, self.linear_spec(on), res=x, out=x, stats_out=stats_out)
            self.pool.put(o)
            return x
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
        out = self.linear(h, self.linear_spec(proj_out), res=a.t)
        self.pool.put(h)
        return Act(out, a.B, a.H, a.W)

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
            if sc.lora is None and c2.lora is None:
                both = self.cache.get(("bias_sum",) + PackCache.ident(c2.b, sc.b), lambda: (c2.b.float() + sc.b.float()).to(self.dtype))
                out = self.conv(g2, c2, shortcut=(a, sc), bias=both)
            else:
                s = self.conv(a, sc)
                out = self.conv(g2, c2, res=s.t)
                self.pool.put(s.t)
        self.pool.put(g2.t)
        return out

    # -- generic fallback -------------------------------------------------------------------------------------------
    def torch_node(self, node: Any, a: Act, out_channels: Optional[int] = None, out_hw: Optional[tuple[int, int]] = None, what: str = "") -> Act:
        """Run an unrecognised sub-tree through its own torch forward on an NCHW copy (shape-preserving unless told)."""
        C2 = out_channels or a.C
        H2, W2 = out_hw or (a.H, a.W)
        nchw = torch.empty(a.B, a.C, a.H, a.W, device=self.device, dtype=self.dtype)
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


# ------------------------------------------------------------------------------------------------ UNet-level context
@dataclass
class UNetContext:
    """Compile-time stand-in for the reference's context store during one UNet forward."""

    low: Lowering
    B: int
    text: dict[tuple[str, str], tuple[Tensor, int]] = field(default_factory=dict)  # padded token buffers + true length
    temb_silu: dict[str, Tensor] = field(default_factory=dict)  # context key -> SiLU(timestep embedding) [B, 1280]
    residuals: list[Any] = field(default_factory=list)
    shapes: list[tuple[int, int]] = field(default_factory=list)
    time_table: dict[tuple[str, int], Tensor] = field(default_factory=dict)  # (context key, id(packed weight)) -> [B, cout] view of the batched launch

    def tokens(self, context: str, key: str) -> tuple[Tensor, int]:
        got = self.text.get((context, key))
        if got is None:
            raise Unsupported(f"context {context}.{key} is not a registered token input")
        return got

    def time_bias(self, spec: ConvSpec) -> Tensor:
        key, lin = spec.time  # type: ignore[misc]
        src = self.temb_silu.get(key)
        if src is None:
            raise Unsupported(f"timestep embedding '{key}' has not been produced yet")
        got = self.time_table.get((key, id(lin.w)))
        if got is not None:  # a column slice of the one launch UNetLowering.batch_time_biases issued for every RangeAdapter2d of this key
            return got
        out = self.low.pool.get(self.B, lin.N)
        self.low.pool.pin(out)
        self.low.linear(src, lin, out=out)
        return out


# ------------------------------------------------------------------------------------------------ whole-UNet lowering
def sinusoid_rows(x: Tensor, dim: int) -> Tensor:
    """range_adapter.py:11-22 on a 1-D float tensor: [cos | sin] of x * 10000^(-i/half), float32."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32, device=x.device) / half
    angle = x.float().unsqueeze(1) * torch.exp(exponent).unsqueeze(0)
    return torch.cat([torch.cos(angle), torch.sin(angle)], dim=-1)


@dataclass
class UNetIO:
    """Static input / output buffers of a lowered UNet (filled by CompiledUNet before a replay)."""

    x: Tensor  # [B, Cin, H, W] compute dtype, NCHW (the reference's input layout)
    timestep: Tensor  # [B] float32
    out: Tensor  # [B, Cout, H, W]
    pooled: Optional[Tensor] = None  # [B, 1280]
    time_ids: Optional[Tensor] = None  # [B, 6] float32
    tokens: dict[tuple[str, str], tuple[Tensor, int]] = field(default_factory=dict)  # (context, key) -> ([B*Lp, width], L)
    conditions: dict[str, Tensor] = field(default_factory=dict)  # control context name -> [B, 3, 8H, 8W]
    t2i: dict[str, list[Tensor]] = field(default_factory=dict)  # T2I-Adapter name -> its feature maps, NCHW, batch 1 or B


class UNetLowering(Lowering):
    """Lowers SDXLUNet / SD1UNet trees (reference xl/unet.py:258-351, sd1/unet.py:165-249), with ControlLoras at
    index 0 (xl/control_lora.py:144-248), into `prologue` + `step`."""

    def lower(self, unet: Any, io: UNetIO) -> None:
        B, _, H, W = io.x.shape
        self.io = io
        ctx = UNetContext(self, B)
        ctx.text = dict(io.tokens)
        n_slots = len(unet.init_context()["unet"]["residuals"])
        ctx.residuals = [None] * n_slots
        cur: Any = None
        with self.in_step():
            for child in kids(unet):
                if isa(child, "ControlLora"):
                    self.control_lora(child, ctx, H, W)
                elif isa(child, "Controlnet"):
                    self.controlnet(child, ctx, H, W)
                elif isa(child, "TimestepEncoder"):
                    self.timestep_encoder(child, ctx, scope=unet)
                elif cname(child) in ("DownBlocks", "UpBlocks"):
                    for stage in kids(child):
                        _expect(isa(stage, "Chain"), "UNet stages must be Chains")
                        for piece in kids(stage):
                            cur = self.piece(piece, cur, ctx, H, W)
                elif cname(child) == "MiddleBlock":
                    for piece in kids(child):
                        cur = self.piece(piece, cur, ctx, H, W)
                elif isa(child, "Residual") and len(kids(child)) == 1 and self._reads_residuals(kids(child)[0]):
                    cur = self.add_last_residual(cur, ctx)  # xl/unet.py:282
                elif isa(child, "Sum") and len(kids(child)) == 2 and self._reads_residuals(kids(child)[0]) and cname(kids(child)[1]) == "MiddleBlock":
                    for piece in kids(kids(child)[1]):  # sd1/unet.py:193-196: residuals[-1] + MiddleBlock(x)
                        cur = self.piece(piece, cur, ctx, H, W)
                    cur = self.add_last_residual(cur, ctx)
                elif isa(child, "Chain") and [cname(k) for k in kids(child)] == ["GroupNorm", "SiLU", "Conv2d"]:
                    cur = self.output_block(child, cur)
                else:
                    raise Unsupported(f"unexpected top-level UNet child {cname(child)}")
            _expect(isinstance(cur, Tensor), "UNet did not end with an output block")

    @staticmethod
    def _reads_residuals(m: Any) -> bool:
        return isa(m, "UseContext") and m.context == "unet" and m.key == "residuals"

    # -- timestep ----------------------------------------------------------------------------------------------
    def _range_encoder(self, enc: Any, res: Optional[Tensor]) -> Tensor:
        ch = kids(enc)
        _expect(len(ch) == 5 and isa(ch[0], "Lambda") and isa(ch[1], "Converter") and isa(ch[3], "SiLU"), "unexpected RangeEncoder layout")
        l1, l2 = self.linear_spec(ch[2]), self.linear_spec(ch[4])
        B = self.io.timestep.shape[0]
        sin = self.pool.get(B, enc.sinusoidal_embedding_dim)
        native.sinusoidal(self.io.timestep, enc.sinusoidal_embedding_dim, sin)
        e1 = self.linear(sin, l1)
        self.pool.put(sin)
        e1s = self.pool.get(B, l1.N)
        native.silu(e1, e1s)
        te = self.linear(e1s, l2, res=res)
        self.pool.put(e1)
        self.pool.put(e1s)
        return te

    def timestep_encoder(self, node: Any, ctx: UNetContext, scope: Any = None) -> None:
        """`scope`: the sub-tree whose RangeAdapter2d's read this encoder's context key (their projections are then batched)."""
        ch = kids(node)
        B = ctx.B
        if len(ch) == 2 and isa(ch[0], "Sum"):  # SDXL: Sum(Chain(UseContext timestep, RangeEncoder), TextTimeEmbedding)
            sc = kids(ch[0])
            _expect(len(sc) == 2 and isa(sc[1], "TextTimeEmbedding") and isa(kids(sc[0])[1], "RangeEncoder"), "unexpected SDXL TimestepEncoder layout")
            tt = kids(sc[1])
            _expect(len(tt) == 5 and isa(tt[0], "Concatenate") and isa(tt[1], "Converter") and isa(tt[3], "SiLU"), "unexpected TextTimeEmbedding layout")
            l1, l2 = self.linear_spec(tt[2]), self.linear_spec(tt[4])
            _expect(self.io.pooled is not None and self.io.time_ids is not None, "SDXL needs pooled_text_embedding and time_ids")
            with self.in_prologue():  # constant over the sampling loop
                pooled, ids, dim = self.io.pooled, self.io.time_ids, sc[1].time_ids_embedding_dim
                _expect(pooled.shape[1] + ids.shape[1] * dim == l1.K, "TextTimeEmbedding width mismatch")
                emb = self.pool.get(B, ids.shape[1] * dim)
                native.sinusoidal(ids, dim, emb, group=ids.shape[1])
                cat = self.pool.get(B, l1.K)
                native.concat2(pooled, emb, cat)
                t1 = self.linear(cat, l1)
                self.pool.put(emb)
                self.pool.put(cat)
                t1s = self.pool.get(B, l1.N)
                native.silu(t1, t1s)
                tte = self.pool.get(B, l2.N)
                self.pool.pin(tte)
                self.linear(t1s, l2, out=tte)
                self.pool.put(t1)
                self.pool.put(t1s)
            temb = self._range_encoder(kids(sc[0])[1], res=tte)
            writer = ch[1]
        else:  # SD1.5: Passthrough(UseContext timestep, RangeEncoder, SetContext)
            _expect(len(ch) == 3 and isa(ch[1], "RangeEncoder"), "unexpected TimestepEncoder layout")
            temb = self._range_encoder(ch[1], res=None)
            writer = ch[2]
        _expect(isa(writer, "SetContext") and writer.context == "range_adapter", "TimestepEncoder must write context range_adapter")
        ts = self.pool.get(B, temb.shape[1])
        self.pool.pin(ts)
        native.silu(temb, ts)
        self.pool.put(temb)
        ctx.temb_silu[writer.key] = ts
        if scope is not None:
            self.batch_time_biases(scope, writer.key, ts, ctx)

    def batch_time_biases(self, scope: Any, key: str, src: Tensor, ctx: UNetContext) -> None:
        """Every RangeAdapter2d below `scope` computes Linear_i(SiLU(timestep embedding)) from the same [B, 1280] row pair
        (range_adapter.py:47-86): one GEMM against the row-concatenated weights instead of one 12-14 us, 2-row launch per ResidualBlock
        (19 per SDXL step); each block's conv then reads its [B, cout] column slice as `rowbias` (ld_rowbias = total width).
        Sites whose Linear carries run-time LoRAs keep their own launch."""
        if os.environ.get("REFINERS_AMD_TIME_BATCH", "1") == "0":
            return
        specs: list[LinSpec] = []

        def visit(m: Any) -> None:
            if isa(m, "RangeAdapter2d"):
                ch = kids(m)
                tc = kids(ch[1]) if len(ch) == 2 and isa(ch[1], "Chain") else []
                if len(tc) == 4 and isa(tc[0], "UseContext") and tc[0].context == "range_adapter" and tc[0].key == key and isa(tc[1], "SiLU"):
                    sp = self.linear_spec(tc[2])
                    if sp.lora is None and not sp.geglu and sp.K == src.shape[1] and (sp.N * self.es) % 16 == 0 and all(sp.w is not o.w for o in specs):
                        specs.append(sp)
                return
            for c in kids(m):
                visit(c)

        visit(scope)
        if len(specs) < 2:
            return
        ck = ("time_cat",) + PackCache.ident(*[sp.w for sp in specs], *[sp.b for sp in specs])

        def make() -> tuple[Tensor, Tensor]:
            w = torch.cat([sp.w for sp in specs], dim=0).contiguous()
            b = torch.cat([sp.b if sp.b is not None else torch.zeros(sp.N, device=sp.w.device, dtype=sp.w.dtype) for sp in specs]).contiguous()
            return w, b

        w, b = self.cache.get(ck, make)
        total = w.shape[0]
        out = self.pool.get(ctx.B, total)
        self.pool.pin(out)
        self.linear(src, LinSpec(w, b), out=out)
        off = 0
        for sp in specs:
            ctx.time_table[(key, id(sp.w))] = out[:, off:off + sp.N]
            off += sp.N
        self.stats["time_bias_batched"] = self.stats.get("time_bias_batched", 0) + len(specs)

    # -- stage pieces ------------------------------------------------------------------------------------------
    def stem(self, conv: Any) -> Act:
        """First convolution, straight from the NCHW latents (im2col of the tiny-channel image, then one GEMM)."""
        _expect(isa(conv, "Conv2d") and conv.kernel_size == (3, 3) and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (1, 1), "unsupported stem conv")
        B, cin, H, W = self.io.x.shape
        _expect(conv.in_channels == cin, "stem conv channel mismatch")
        kp = (9 * cin + self.kblk - 1) // self.kblk * self.kblk

        def pack() -> Tensor:
            wp = torch.zeros(conv.out_channels, kp, device=self.device, dtype=self.dtype)
            wp[:, : 9 * cin] = native.pack_conv_weight(self.cvt(conv.weight))
            return wp

        wp = self.cache.get(("stem", kp) + PackCache.ident(conv.weight), pack)
        cols = self.pool.get(B * H * W, kp)
        native.im2col3x3_nchw(self.io.x, cols)
        out = self.pool.get(B * H * W, conv.out_channels)
        native.gemm([(cols, wp)], out, bias=self._w(conv.bias))
        self.pool.put(cols)
        return Act(out, B, H, W)

    def _release(self, a: Optional[Act]) -> None:
        if a is not None:
            self.pool.put(a.t)

    def piece(self, m: Any, cur: Optional[Act], ctx: UNetContext, H: int, W: int) -> Optional[Act]:
        if cur is None:
            return self.stem(m)
        if isa(m, "ResidualBlock"):
            out = self.residual_block(m, cur, ctx)
        elif isa(m, "CrossAttentionBlock2d"):
            out = self.cross_attention_2d(m, cur, ctx)
        elif isa(m, "Downsample"):
            ch = kids(m)
            conv = ch[-1]
            _expect(all(isa(c, "SetContext", "Lambda") for c in ch[:-1]), "unexpected Downsample layout")
            if any(isa(c, "SetContext") for c in ch[:-1]):
                ctx.shapes.append((cur.H, cur.W))
            explicit_pad = any(isa(c, "Lambda") for c in ch[:-1])  # padding=0 variant: F.pad(x, (0, 1, 0, 1)) + unpadded conv
            _expect(explicit_pad == (m.padding == 0), "Downsample padding attribute and layout disagree")
            spec = self.conv_spec(conv, asym=explicit_pad)
            _expect(spec.stride == 2 and spec.ksize == 3, "unexpected Downsample convolution")
            out = self.conv(cur, spec)
        elif isa(m, "Upsample"):
            ch = kids(m)
            _expect(len(ch) == 3 and isa(ch[0], "Parallel") and isa(ch[1], "Interpolate") and ch[1].mode == "nearest", "unexpected Upsample layout")
            src = kids(ch[0])[1]
            th, tw = ctx.shapes.pop() if isa(src, "UseContext") else (cur.H * m.upsample_factor, cur.W * m.upsample_factor)
            _expect((th, tw) == (2 * cur.H, 2 * cur.W), "only exact 2x nearest upsampling is lowered")
            out = self.conv(cur, self.conv_spec(ch[2]), ups=2)
        elif isa(m, "ResidualAccumulator"):
            self.accumulate(ctx, m.n, cur)
            return cur
        elif isa(m, "ResidualConcatenator"):
            skip = self.slot(ctx, m.n)
            _expect(skip is not None and (skip.B, skip.H, skip.W) == (cur.B, cur.H, cur.W), "skip tensor missing or of another size")
            cat = self.pool.get(cur.M, cur.C + skip.C)
            native.concat2(cur.t, skip.t, cat)
            out = Act(cat, cur.B, cur.H, cur.W)
        elif isa(m, "ZeroConvolution"):
            self.zero_convolution(m, cur, ctx)
            return cur
        elif isa(m, "Residual") and len(kids(m)) == 2 and isa(kids(m)[0], "UseContext") and isa(kids(m)[1], "ConditionEncoder"):
            out = self.add_condition(m, cur)
        elif isa(m, "T2IFeatures"):
            out = self.add_t2i_features(m, cur)
        elif isa(m, "SelfAttentionShape"):  # SAG: remembers the (H, W) of the feature map the tapped attention runs on
            self.sag_shape = (cur.H, cur.W)
            return cur
        else:
            out = self.torch_node(m, cur)
        self._release(cur)
        return out

    def slot(self, ctx: UNetContext, n: int) -> Optional[Act]:
        return ctx.residuals[n]

    def accumulate(self, ctx: UNetContext, n: int, a: Act) -> None:
        """residuals[n] <- a + residuals[n]   (unet.py:54-66); aliasing `a` when the slot still holds its initial 0.0."""
        prev = ctx.residuals[n]
        if prev is None:
            self.pool.pin(a.t)
            ctx.residuals[n] = a
            return
        _expect((prev.B, prev.H, prev.W, prev.C) == (a.B, a.H, a.W, a.C), "residual slot shape mismatch")
        s = self.pool.get(a.M, a.C)
        self.pool.pin(s)
        native.axpby(a.t, 1.0, prev.t, 1.0, s)
        ctx.residuals[n] = Act(s, a.B, a.H, a.W)

    def add_last_residual(self, cur: Act, ctx: UNetContext) -> Act:
        last = ctx.residuals[-1]
        if last is None:  # 0.0 in the reference: x + 0.0
            return cur
        out = self.pool.get(cur.M, cur.C)
        native.axpby(cur.t, 1.0, last.t, 1.0, out)
        self._release(cur)
        return Act(out, cur.B, cur.H, cur.W)

    def output_block(self, node: Any, cur: Act) -> Tensor:
        gn, _, conv = kids(node)
        g = self.groupnorm(cur, gn, silu=True)
        self._release(cur)
        y = self.conv(g, self.conv_spec(conv))
        self.pool.put(g.t)
        native.nhwc_to_nchw(y.tokens(), self.io.out, y.C)
        self.pool.put(y.t)
        return self.io.out

    # -- ControlLora -------------------------------------------------------------------------------------------
    def zero_convolution(self, m: Any, cur: Act, ctx: UNetContext) -> None:
        """residuals[n] += scale * conv1x1(x)   (control_lora.py:90-141): the scale is folded into the packed weights."""
        ch = kids(m)
        _expect(len(ch) == 3 and isa(ch[0], "Conv2d") and isa(ch[1], "Multiply") and isa(ch[2], "ResidualAccumulator") and ch[1].bias == 0.0, "unexpected ZeroConvolution layout")
        conv, scale = ch[0], float(ch[1].scale)
        _expect(conv.kernel_size == (1, 1), "ZeroConvolution must be 1x1")
        w = self.cache.get(("zc_w", scale) + PackCache.ident(conv.weight), lambda: (conv.weight.detach().to(self.device, torch.float32).reshape(conv.out_channels, conv.in_channels) * scale).to(self.dtype).contiguous())
        b = self.cache.get(("zc_b", scale) + PackCache.ident(conv.bias), lambda: (conv.bias.detach().to(self.device, torch.float32) * scale).to(self.dtype).contiguous())
        prev = ctx.residuals[ch[2].n]
        z = self.pool.get(cur.M, conv.out_channels)
        self.pool.pin(z)
        native.gemm([(cur.t, w)], z, bias=b, res=None if prev is None else prev.t)
        ctx.residuals[ch[2].n] = Act(z, cur.B, cur.H, cur.W)

    def _padded_conv_spec(self, conv: Any, cin_pad: int, cout_pad: int) -> ConvSpec:
        """Conv2d whose channel counts are below the kernel's 128-byte granularity: zero-pad input channels (weight
        columns) and output channels (weight rows, bias) so that padded activations stay exactly zero."""
        _expect(isa(conv, "Conv2d") and conv.kernel_size == (3, 3) and tuple(conv.padding) == (1, 1) and conv.stride[0] == conv.stride[1], "unexpected ConditionEncoder conv")
        o, i = conv.out_channels, conv.in_channels

        def pack() -> tuple[Tensor, Tensor]:
            w = torch.zeros(cout_pad, cin_pad, 3, 3, device=self.device, dtype=self.dtype)
            w[:o, :i] = conv.weight.detach().to(device=self.device, dtype=self.dtype)
            b = torch.zeros(cout_pad, device=self.device, dtype=self.dtype)
            b[:o] = conv.bias.detach().to(device=self.device, dtype=self.dtype)
            return native.pack_conv_weight(w), b

        wp, bp = self.cache.get(("padconv", cin_pad, cout_pad) + PackCache.ident(conv.weight, conv.bias), pack)
        return ConvSpec(wp, bp, cin_pad, cout_pad, 3, conv.stride[0])

    def condition_encoder(self, enc: Any, cond: Tensor) -> Act:
        """ConditionEncoder (control_lora.py:14-87): (B, 3, 8H, 8W) -> (B, 320, H, W), eight 3x3 convs with SiLU.  Runs in
        the prologue (the control image is constant over the sampling loop) on channel-padded NHWC activations."""
        convs = [m for m in enc.modules() if isa(m, "Conv2d")]
        order = [m for m in enc.modules() if isa(m, "Conv2d", "SiLU")]
        _expect(len(convs) == 8 and isa(order[-1], "Conv2d"), "unexpected ConditionEncoder layout")
        pad = lambda c: (c + self.kblk - 1) // self.kblk * self.kblk
        B, C, H, W = cond.shape
        x = torch.zeros(B * H * W, pad(C), device=self.device, dtype=self.dtype)  # padding channels stay zero for ever
        self.prologue_keep = getattr(self, "prologue_keep", []) + [x]
        a = Act(x, B, H, W)
        native.nchw_to_nhwc(cond, a.tokens())
        for k, m in enumerate(order):
            if isa(m, "SiLU"):
                native.silu(a.t, a.t)
                continue
            last = m is order[-1]
            spec = self._padded_conv_spec(m, a.C, m.out_channels if last else pad(m.out_channels))
            nxt = self.conv(a, spec)
            if a.t is not x:
                self.pool.put(a.t)
            a = nxt
        return a

    def add_condition(self, m: Any, cur: Act) -> Act:
        """x + ConditionEncoder(condition)   (control_lora.py:190-202), encoder output produced in the prologue."""
        reader, enc = kids(m)
        # ControlLora: one context per adapter, key "condition"; SD1.5 Controlnet: shared context "controlnet", key "condition_<name>"
        cname_ = reader.context if reader.key == "condition" else f"{reader.context}.{reader.key}"
        cond = self.io.conditions.get(cname_)
        _expect(cond is not None, f"no condition image registered for {cname_}")
        with self.in_prologue():
            e = self.condition_encoder(enc, cond)
            self.pool.pin(e.t)
        _expect((e.B, e.H, e.W, e.C) == (cur.B, cur.H, cur.W, cur.C), "ConditionEncoder output does not match the UNet stem")
        out = self.pool.get(cur.M, cur.C)
        native.axpby(cur.t, 1.0, e.t, 1.0, out)
        return Act(out, cur.B, cur.H, cur.W)

    def add_t2i_features(self, m: Any, cur: Act) -> Act:
        """x + scale * features[index]   (latent_diffusion/t2i_adapter.py:166-177): the feature map comes from the
        T2I-Adapter's condition encoder, once per image, as NCHW; it is turned token-major in the prologue (broadcast over
        the CFG batch when it has batch 1) and added with the node's live scale in one launch per step."""
        feats = self.io.t2i.get(m.name)
        _expect(feats is not None and 0 <= m.index < len(feats), f"no T2I-Adapter features registered for '{m.name}'")
        f = feats[m.index]
        fb, fc, fh, fw = f.shape
        _expect((fc, fh, fw) == (cur.C, cur.H, cur.W) and fb in (1, cur.B), f"T2I feature {m.index} of '{m.name}' is {tuple(f.shape)}, the UNet has {(cur.B, cur.C, cur.H, cur.W)} here")
        with self.in_prologue():
            tok = self.pool.get(cur.M, cur.C)
            self.pool.pin(tok)
            hw = fh * fw
            if fb == cur.B:
                native.nchw_to_nhwc(f, tok.view(cur.B, hw, cur.C))
            else:
                for b in range(cur.B):
                    native.nchw_to_nhwc(f, tok[b * hw : (b + 1) * hw].view(1, hw, cur.C))
        out = self.pool.get(cur.M, cur.C)
        native.axpby(cur.t, 1.0, tok, float(m.scale), out)
        self.stats["t2i_sites"] = self.stats.get("t2i_sites", 0) + 1
        return Act(out, cur.B, cur.H, cur.W)

    # -- SD1.5 ControlNet -----------------------------------------------------------------------------------------
    def controlnet(self, node: Any, ctx: UNetContext, H: int, W: int) -> None:
        """Controlnet = Passthrough(TimestepEncoder', Slicing(:4), DownBlocks', MiddleBlock') (stable_diffusion_1/controlnet.py:72-166):
        a second, separately weighted encoder half in front of the UNet; after every one of its 12 down blocks and after its
        middle block, residuals[n] += scale * scale_decay^(12 - n) * conv1x1_n(x) (:152-166).  Same shape of work as ControlLora:
        each tap is one GEMM (scale folded into the packed 1x1 weights, previous slot value as the residual operand)."""
        ch = kids(node)
        _expect(len(ch) == 4 and isa(ch[0], "TimestepEncoder") and isa(ch[1], "Slicing") and cname(ch[2]) == "DownBlocks" and cname(ch[3]) == "MiddleBlock",
                "unexpected Controlnet layout")
        _expect(ch[1].dim == 1 and ch[1].start == 0 and ch[1].end == 4 and self.io.x.shape[1] == 4, "Controlnet on a UNet input with more than 4 channels is not lowered")
        sub = UNetContext(self, ctx.B, text=ctx.text, temb_silu=ctx.temb_silu, residuals=ctx.residuals, shapes=[])
        self.timestep_encoder(ch[0], sub, scope=node)
        cur: Optional[Act] = None
        stages = [(n, kids(stage)) for n, stage in enumerate(kids(ch[2]))] + [(12, kids(ch[3]))]
        _expect(len(stages) == 13, "Controlnet must have 12 down blocks and a middle block")
        for n, pieces in stages:
            for piece in pieces:
                if isa(piece, "Passthrough") and len(kids(piece)) == 2 and isa(kids(piece)[0], "Conv2d") and isa(kids(piece)[1], "Lambda"):
                    self.controlnet_tap(node, kids(piece)[0], n, cur, sub)
                else:
                    cur = self.piece(piece, cur, sub, H, W)
        self._release(cur)
        self.stats["controlnets"] = self.stats.get("controlnets", 0) + 1

    def controlnet_tap(self, node: Any, conv: Any, n: int, cur: Optional[Act], ctx: UNetContext) -> None:
        _expect(cur is not None and conv.kernel_size == (1, 1) and conv.in_channels == cur.C, "unexpected Controlnet residual tap")
        scale = float(node.scale) * float(node.scale_decays[n])
        w = self.cache.get(("cn_w", scale) + PackCache.ident(conv.weight), lambda: (conv.weight.detach().to(self.device, torch.float32).reshape(conv.out_channels, conv.in_channels) * scale).to(self.dtype).contiguous())
        b = self.cache.get(("cn_b", scale) + PackCache.ident(conv.bias), lambda: (conv.bias.detach().to(self.device, torch.float32) * scale).to(self.dtype).contiguous())
        prev = ctx.residuals[n]
        z = self.pool.get(cur.M, conv.out_channels)
        self.pool.pin(z)
        native.gemm([(cur.t, w)], z, bias=b, res=None if prev is None else prev.t)
        ctx.residuals[n] = Act(z, cur.B, cur.H, cur.W)

    def control_lora(self, node: Any, ctx: UNetContext, H: int, W: int) -> None:
        """Passthrough(TimestepEncoder', DownBlocks', MiddleBlock'): fills ctx.residuals, returns nothing."""
        ch = kids(node)
        _expect(len(ch) == 3 and isa(ch[0], "TimestepEncoder") and cname(ch[1]) == "DownBlocks" and cname(ch[2]) == "MiddleBlock", "unexpected ControlLora layout")
        sub = UNetContext(self, ctx.B, text=ctx.text, temb_silu=ctx.temb_silu, residuals=ctx.residuals, shapes=[])
        self.timestep_encoder(ch[0], sub, scope=node)
        cur: Optional[Act] = None
        for stage in kids(ch[1]):
            for piece in kids(stage):
                cur = self.piece(piece, cur, sub, H, W)
        for piece in kids(ch[2]):
            cur = self.piece(piece, cur, sub, H, W)
        self._release(cur)
