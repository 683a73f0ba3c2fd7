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
    cs: Any = None  # column statistics [M / 32, C, 2] float32 written by the launch that produced `t` (Lowering.colstats_for), or None

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


class CatAct:
    """Concatenate(a, b) along the channels that is never materialised (ResidualConcatenator, unet.py:69-85): its only consumers inside a
    ResidualBlock -- the first GroupNorm and the 1x1 shortcut convolution -- read the two parts where they are (two-source GroupNorm, two K
    segments).  Anything else asks Lowering.materialise() for the real tensor."""

    def __init__(self, a: Act, b: Act) -> None:
        assert (a.B, a.H, a.W) == (b.B, b.H, b.W)
        self.a, self.b = a, b
        self.B, self.H, self.W = a.B, a.H, a.W

    @property
    def C(self) -> int:
        return self.a.C + self.b.C

    @property
    def M(self) -> int:
        return self.a.M

    @property
    def HW(self) -> int:
        return self.H * self.W


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
    keeps alive, see _Src), so that a re-lowering after inject / eject / scale change only re-packs what actually changed.

    Multi-GPU (refiners_amd.parallel.broadcast_packs): only the source rank COMPUTES the packs.  It lowers normally and publishes
    `manifest()` -- for every entry, in creation order, either "alias" (the value is the leaf's own storage: every rank makes it
    itself, nothing to move) or the shapes / dtypes of the tensors inside the value; the other ranks lower with `adopt(manifest)`,
    which answers each miss with uninitialised storage of that description instead of calling `make()` (no K-blocking, merging,
    LayerNorm folding, concatenation on the receivers), and one bucketed broadcast then fills `leaves()` in place (in place: the
    recorded programs already hold the tensors' addresses)."""

    def __init__(self) -> None:
        self.store: dict[tuple, Any] = {}
        self.hits = 0
        self.used: set[tuple] = set()
        self.order: list[tuple] = []          # keys in creation order (what the manifest walks)
        self._adopt: Optional[list] = None    # receiver side: the source's manifest
        self._cursor = 0
        self._mark = 0                        # len(order) when the current hand-over window opened (see mark())
        self.made = 0                         # make() calls that really ran here (receivers: aliases only)
        self.build_device: Optional[torch.device] = None  # where a receiver allocates (the Lowering that owns the cache sets it)

    @staticmethod
    def ident(*tensors: Optional[Tensor]) -> tuple:
        return tuple(_Src(t) if t is not None else None for t in tensors)

    def get(self, key: tuple, make: Callable[[], Any]) -> Any:
        self.used.add(key)
        if key in self.store:
            self.hits += 1
            return self.store[key]
        if self._adopt is not None:
            assert self._cursor < len(self._adopt), "the source rank lowered fewer packed weights than this rank asks for (different trees?)"
            spec = self._adopt[self._cursor]
            self._cursor += 1
            assert spec[-1] == _key_desc(key), (
                f"packed-weight hand-over out of step at entry {self._cursor - 1}: the source made {spec[-1]}, this rank asks for {_key_desc(key)} "
                "(the two ranks did not lower the same tree from the same cache state)")
            if spec[0] == "alias":
                v = make()
                self.made += 1
            else:
                v = _build(spec[1], self.build_device or _key_device(key))
        else:
            v = make()
            self.made += 1
        self.store[key] = v
        self.order.append(key)
        return v

    def sweep(self) -> None:
        for k in list(self.store):
            if k not in self.used:
                del self.store[k]
        self._mark = sum(1 for k in self.order[: self._mark] if k in self.store)  # the window keeps starting at the same surviving entry
        self.order = [k for k in self.order if k in self.store]
        self.used = set()

    # -- multi-GPU hand-over ----------------------------------------------------------------------------------------------------
    # A hand-over covers a WINDOW of the cache: the entries created by one lowering.  Both sides call mark() right before that lowering
    # (the receivers through adopt()), so entries left by earlier lowerings -- a first broadcast, a LoRA scale change, sweep() survivors -- are
    # neither published nor consumed, and each manifest entry carries a rank-independent description of its key (tag, scalars, source shapes)
    # that the receiver compares before it builds anything: a sequence that is out of step asserts instead of mis-assigning storage.
    def mark(self) -> int:
        self._mark = len(self.order)
        return self._mark

    def manifest(self) -> list:
        """Source side: one picklable description per entry created since mark(), in creation order."""
        out = []
        for key in self.order[self._mark :]:
            v = self.store[key]
            src = {s.t.data_ptr() for s in key if isinstance(s, _Src)}
            leaves = _leaves(v)
            if not leaves or not src or all(t.data_ptr() in src for t in leaves):
                # nothing to move: no tensors inside, a pure function of the key (index tables: no source tensor in the key), or the source
                # tensors' own storage -- every rank makes these itself.  (A value that mixes views and fresh tensors travels whole: its fresh
                # part may have been computed from other packs, which a receiver does not hold yet while it lowers.)
                out.append(("alias", _key_desc(key)))
            else:
                out.append(("recv", _describe(v), _key_desc(key)))
        return out

    def adopt(self, manifest: Optional[list]) -> None:
        """Receiver side, BEFORE lowering: answer misses from the source's manifest (None: leave adopt mode).  Opens the window."""
        self._adopt, self._cursor = manifest, 0
        if manifest is not None:
            self.mark()

    def leaves(self, manifest: list) -> list[Tensor]:
        """The tensors of every "recv" entry of the window, in manifest order: what the broadcast writes (source) / fills (receivers)."""
        window = self.order[self._mark :]
        assert len(manifest) == len(window), f"the source published {len(manifest)} packed-weight entries, this rank made {len(window)} in the same lowering"
        out: list[Tensor] = []
        for key, spec in zip(window, manifest):
            assert spec[-1] == _key_desc(key), (spec[-1], _key_desc(key))
            if spec[0] == "recv":
                out.extend(_leaves(self.store[key]))
        self._adopt = None
        return out


def _key_desc(key: tuple) -> tuple:
    """What two ranks can compare about a cache key: its scalars as they are, its source tensors as (shape, dtype)."""
    out = []
    for s in key:
        if isinstance(s, _Src):
            out.append(("src", tuple(s.t.shape), str(s.t.dtype).replace("torch.", "")))
        elif s is None or isinstance(s, (int, float, str, bool)):
            out.append(s)
        else:
            out.append(repr(type(s).__name__))
    return tuple(out)


def _key_device(key: tuple) -> torch.device:
    return next((s.t.device for s in key if isinstance(s, _Src)), torch.device("cpu"))


def _leaves(v: Any) -> list[Tensor]:
    if isinstance(v, Tensor):
        return [v]
    if isinstance(v, native.KBlocked):
        return [v.t]
    if isinstance(v, LoraPack):
        return [t for f in ("a_cat", "bs_cat", "a_kb", "bs_r") for t in _leaves(getattr(v, f))]
    if isinstance(v, (tuple, list)):
        return [t for x in v for t in _leaves(x)]
    return []


def _describe(v: Any) -> Any:
    if isinstance(v, Tensor):
        return ("T", tuple(v.shape), str(v.dtype).replace("torch.", ""), v.device.type)
    if isinstance(v, native.KBlocked):
        return ("KB", v.N, v.K, _describe(v.t))
    if isinstance(v, LoraPack):
        return ("LP", _describe(v.a_cat), _describe(v.bs_cat), v.conv, _describe(v.a_kb), _describe(v.bs_r))
    if isinstance(v, tuple):
        return ("tuple", [_describe(x) for x in v])
    if isinstance(v, list):
        return ("list", [_describe(x) for x in v])
    assert v is None or isinstance(v, (int, float, str, bool)), f"PackCache value of type {type(v).__name__} cannot be handed to another rank"
    return ("V", v)


def _build(spec: Any, device: torch.device) -> Any:
    kind = spec[0]
    if kind == "T":
        return torch.empty(spec[1], dtype=getattr(torch, spec[2]), device=device if spec[3] == device.type else torch.device(spec[3]))
    if kind == "KB":
        return native.KBlocked(_build(spec[3], device), _adopt=(spec[1], spec[2]))
    if kind == "LP":
        return LoraPack(_build(spec[1], device), _build(spec[2], device), None if spec[3] is None else tuple(spec[3]), _build(spec[4], device), _build(spec[5], device))
    if kind == "tuple":
        return tuple(_build(x, device) for x in spec[1])
    if kind == "list":
        return [_build(x, device) for x in spec[1]]
    return spec[1]
