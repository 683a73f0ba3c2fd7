"""Value types of the lowering (refiners_amd.engine.lowering): tree matchers that work on class NAMES (so that trees built from refiners'
own classes are accepted), activation views, packed-weight specs, the static buffer pool and the PackCache of converted / packed weights."""
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
