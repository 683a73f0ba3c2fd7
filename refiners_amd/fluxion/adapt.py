"""Adapters: the plugin mechanism of the Chain tree, and the LoRA family built on it.

An adapter is a Chain that wraps a target node and takes its place in the target's parent (`inject`), or gives the
place back (`eject`).  This is the drop-in boundary of the whole project: the MI355X fused nodes of
refiners_amd/engine are adapters too.  Mirrors:

* `Adapter`            reference src/refiners/fluxion/adapters/adapter.py:14-127
* `Lora` / `LinearLora` / `Conv2dLora` / `LoraAdapter` / `auto_attach_loras`
                       reference src/refiners/fluxion/adapters/lora.py:14-523
"""
from __future__ import annotations

import contextlib
from typing import Any, Generic, Iterator, TypeVar

import torch
from torch import Tensor, nn

from . import leaves as L
from .tree import Chain, ContextModule, Sum, WeightedModule, bump_epoch

T = TypeVar("T", bound=nn.Module)


class Adapter(Generic[T]):
    """Mixin for Chain subclasses that wrap a `target` and can splice themselves in and out of its parent."""

    _target: list[T]

    def __init_subclass__(cls, **kwargs: Any) -> None:
        super().__init_subclass__(**kwargs)
        assert issubclass(cls, Chain), f"Adapter {cls.__name__} must be a Chain"

    @property
    def target(self) -> T:
        return self._target[0]

    @contextlib.contextmanager
    def setup_adapter(self, target: T) -> Iterator[None]:
        """To be wrapped around the Chain constructor call of the adapter: records the target and keeps the target's
        parent link untouched while the adapter adopts it as a child."""
        assert isinstance(self, Chain)
        assert not hasattr(self, "_modules") or len(self) == 0, "Call the Chain constructor in the setup_adapter context."
        self._target = [target]
        if isinstance(target, ContextModule):
            with target.no_parent_refresh():
                yield
        else:
            yield

    def inject(self, parent: Chain | None = None) -> Any:
        """Replace the target by this adapter inside the target's parent (found from `parent` when the target is a
        plain leaf that does not know its parent)."""
        assert isinstance(self, Chain)
        if parent is None and isinstance(self.target, ContextModule):
            parent = self.target.parent
            if parent is not None:
                assert isinstance(parent, Chain), f"{self.target} has invalid parent {parent}"
        inner_parent = self.find_parent(self.target)
        if parent is None:
            if isinstance(self.target, ContextModule):
                self.target._set_parent(inner_parent)
            return self
        parent.ensure_find_parent(self.target).replace(self.target, self, old_module_parent=inner_parent)
        return self

    def eject(self) -> None:
        """Inverse of inject: the (possibly re-adapted) target takes the adapter's place again."""
        assert isinstance(self, Chain)
        restored = lookup_top_adapter(self, self.target)
        parent = self.parent
        if parent is None:
            if isinstance(restored, ContextModule):
                restored._set_parent(None)
        else:
            parent.replace(self, restored)

    def _pre_structural_copy(self) -> None:
        if isinstance(self.target, Chain):
            raise RuntimeError(f"Chain adapters ({self}) typically cannot be copied, eject them first.")

    def _post_structural_copy(self, source: "Adapter[T]") -> None:
        self._target = [source.target]


def lookup_top_adapter(top: Chain, target: nn.Module) -> nn.Module:
    """The outermost adapter between `top` and `target` (or the target itself when there is none)."""
    p = top.find_parent(target)
    if p is None or p is top:
        return target
    best: nn.Module = target
    while p is not top:
        if isinstance(p, Adapter):
            best = p
        assert p.parent, f"parent tree of {top} is broken"
        p = p.parent
    return best


# ------------------------------------------------------------------------------------------------ LoRA
class Lora(Chain):
    """Chain(down, up, Multiply(scale)): the low-rank update scale * up(down(x)).

    Abstract: use LinearLora or Conv2dLora.  `down ~ N(0, 1/rank)`, `up = 0` at construction (lora.py:57-60).
    """

    def __init__(self, name: str, /, rank: int = 16, scale: float = 1.0, device: Any = None, dtype: Any = None) -> None:
        self.name = name
        self._rank = rank
        self._scale = scale
        super().__init__(*self.lora_layers(device=device, dtype=dtype), L.Multiply(scale))
        self.reset_parameters()

    def lora_layers(self, device: Any = None, dtype: Any = None) -> tuple[WeightedModule, WeightedModule]:
        raise NotImplementedError

    def is_compatible(self, layer: WeightedModule, /) -> bool:
        raise NotImplementedError

    def reset_parameters(self) -> None:
        nn.init.normal_(self.down.weight, std=1 / self.rank)
        nn.init.zeros_(self.up.weight)

    @property
    def down(self) -> Any:
        assert isinstance(self[0], WeightedModule)
        return self[0]

    @property
    def up(self) -> Any:
        assert isinstance(self[1], WeightedModule)
        return self[1]

    @property
    def rank(self) -> int:
        return self._rank

    @property
    def scale(self) -> float:
        return self._scale

    @scale.setter
    def scale(self, value: float) -> None:
        self._scale = value
        self.ensure_find(L.Multiply).scale = value

    @classmethod
    def from_weights(cls, name: str, /, down: Tensor, up: Tensor) -> "Lora":
        if up.ndim == 2 and down.ndim == 2:
            return LinearLora.from_weights(name, down=down, up=up)
        if up.ndim == 4 and down.ndim == 4:
            return Conv2dLora.from_weights(name, down=down, up=up)
        raise ValueError(f"Unsupported weight shapes: up={up.shape}, down={down.shape}")

    @classmethod
    def from_dict(cls, name: str, /, state_dict: dict[str, Tensor]) -> dict[str, "Lora"]:
        """`state_dict` lists (down, up) weight pairs in order; the key of a pair is its down key minus the last two
        dotted components."""
        items = [(k, v) for k, v in state_dict.items() if ".weight" in k]
        out: dict[str, Lora] = {}
        for (down_key, down), (_, up) in zip(items[::2], items[1::2]):
            out[".".join(down_key.split(".")[:-2])] = cls.from_weights(name, down=down, up=up)
        return out

    def auto_attach(
        self, target: Chain, include: list[str] | None = None, exclude: list[str] | None = None
    ) -> "tuple[LoraAdapter, Chain | None] | None":
        """First shape-compatible layer of `target` (walk order) that does not carry a LoRA of this name yet."""
        for layer, parent in target.walk(self.up.__class__):
            if isinstance(parent, Lora):
                continue
            if include is not None or exclude is not None:
                lineage = [p.__class__.__name__ for p in parent.get_parents() + [parent]]
                if include is not None and not any(n in include for n in lineage):
                    continue
                if exclude is not None and any(n in exclude for n in lineage):
                    continue
            if not self.is_compatible(layer):
                continue
            if isinstance(parent, LoraAdapter):
                if self.name in parent.names:
                    continue
                parent.add_lora(self)
                return parent, None
            return LoraAdapter(layer, self), parent
        return None

    def load_weights(self, down_weight: Tensor, up_weight: Tensor) -> None:
        assert down_weight.shape == self.down.weight.shape
        assert up_weight.shape == self.up.weight.shape
        self.down.weight = nn.Parameter(down_weight.to(device=self.device, dtype=self.dtype))
        self.up.weight = nn.Parameter(up_weight.to(device=self.device, dtype=self.dtype))
        bump_epoch()


class LinearLora(Lora):
    def __init__(
        self, name: str, /, in_features: int, out_features: int, rank: int = 16, scale: float = 1.0, device: Any = None, dtype: Any = None
    ) -> None:
        self.in_features = in_features
        self.out_features = out_features
        super().__init__(name, rank=rank, scale=scale, device=device, dtype=dtype)

    @classmethod
    def from_weights(cls, name: str, /, down: Tensor, up: Tensor) -> "LinearLora":
        assert up.ndim == 2 and down.ndim == 2
        assert down.shape[0] == up.shape[1], f"Rank mismatch: down rank={down.shape[0]} and up rank={up.shape[1]}"
        lora = cls(name, in_features=down.shape[1], out_features=up.shape[0], rank=down.shape[0], device=up.device, dtype=up.dtype)
        lora.load_weights(down_weight=down, up_weight=up)
        return lora

    def lora_layers(self, device: Any = None, dtype: Any = None) -> tuple[L.Linear, L.Linear]:
        return (
            L.Linear(self.in_features, self.rank, bias=False, device=device, dtype=dtype),
            L.Linear(self.rank, self.out_features, bias=False, device=device, dtype=dtype),
        )

    def is_compatible(self, layer: WeightedModule, /) -> bool:
        return isinstance(layer, L.Linear) and layer.in_features == self.in_features and layer.out_features == self.out_features


class Conv2dLora(Lora):
    def __init__(
        self,
        name: str,
        /,
        in_channels: int,
        out_channels: int,
        rank: int = 16,
        scale: float = 1.0,
        kernel_size: tuple[int, int] = (1, 3),
        stride: tuple[int, int] = (1, 1),
        padding: tuple[int, int] = (0, 1),
        device: Any = None,
        dtype: Any = None,
    ) -> None:
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.stride = stride
        self.padding = padding
        super().__init__(name, rank=rank, scale=scale, device=device, dtype=dtype)

    @classmethod
    def from_weights(cls, name: str, /, down: Tensor, up: Tensor) -> "Conv2dLora":
        assert up.ndim == 4 and down.ndim == 4
        assert down.shape[0] == up.shape[1], f"Rank mismatch: down rank={down.shape[0]} and up rank={up.shape[1]}"
        kd, ku = down.shape[2], up.shape[2]
        lora = cls(
            name, in_channels=down.shape[1], out_channels=up.shape[0], rank=down.shape[0], kernel_size=(kd, ku),
            padding=(1 if kd == 3 else 0, 1 if ku == 3 else 0), device=up.device, dtype=up.dtype,
        )
        lora.load_weights(down_weight=down, up_weight=up)
        return lora

    def lora_layers(self, device: Any = None, dtype: Any = None) -> tuple[L.Conv2d, L.Conv2d]:
        return (
            L.Conv2d(self.in_channels, self.rank, kernel_size=self.kernel_size[0], stride=self.stride[0],
                     padding=self.padding[0], use_bias=False, device=device, dtype=dtype),
            L.Conv2d(self.rank, self.out_channels, kernel_size=self.kernel_size[1], stride=self.stride[1],
                     padding=self.padding[1], use_bias=False, device=device, dtype=dtype),
        )

    def is_compatible(self, layer: WeightedModule, /) -> bool:
        if isinstance(layer, L.Conv2d) and layer.in_channels == self.in_channels and layer.out_channels == self.out_channels:
            self.down.stride = layer.stride  # the down conv follows the target's stride (lora.py:377)
            return True
        return False


class LoraAdapter(Sum, Adapter[WeightedModule]):
    """target(x) + sum_i lora_i(x)  (reference: fluxion/adapters/lora.py:383-448)."""

    def __init__(self, target: WeightedModule, /, *loras: Lora) -> None:
        with self.setup_adapter(target):
            super().__init__(target, *loras)

    @property
    def lora_layers(self) -> Iterator[Lora]:
        return self.layers(Lora)

    @property
    def names(self) -> list[str]:
        return [lora.name for lora in self.lora_layers]

    @property
    def loras(self) -> dict[str, Lora]:
        return {lora.name: lora for lora in self.lora_layers}

    @property
    def scales(self) -> dict[str, float]:
        return {lora.name: lora.scale for lora in self.lora_layers}

    @scales.setter
    def scale(self, values: dict[str, float]) -> None:
        for name, value in values.items():
            self.loras[name].scale = value

    def add_lora(self, lora: Lora, /) -> None:
        assert lora.name not in self.names, f"LoRA layer with name {lora.name} already exists"
        self.append(lora)

    def remove_lora(self, name: str, /) -> Lora | None:
        lora = self.loras.get(name)
        if lora is not None:
            self.remove(lora)
        return lora


def _attach_all(
    loras: dict[str, Lora], target: Chain, include: list[str] | None, exclude: list[str] | None, debug_map: list[tuple[str, str]] | None
) -> list[str]:
    failed: list[str] = []
    for key, lora in loras.items():
        got = lora.auto_attach(target, include=include, exclude=exclude)
        if got is None:
            failed.append(key)
            continue
        adapter, parent = got
        if parent is None:
            if debug_map is not None:
                debug_map.append((key, adapter.get_path()))
            continue
        if debug_map is not None:
            debug_map.append((key, adapter.target.get_path(parent)))
        adapter.inject(parent)
    return failed


def auto_attach_loras(
    loras: dict[str, Lora],
    target: Chain,
    /,
    include: list[str] | None = None,
    exclude: list[str] | None = None,
    sanity_check: bool = True,
    debug_map: list[tuple[str, str]] | None = None,
) -> list[str]:
    """Attach each LoRA to the first compatible layer; returns the keys that found no home (lora.py:479-523).

    With `sanity_check` every LoRA must find a home, and a second attempt with clones of the same LoRAs must place
    none of them (each compatible layer already carries that name).
    """
    if not sanity_check:
        return _attach_all(loras, target, include, exclude, debug_map)
    clones = {k: Lora.from_weights(v.name, v.down.weight, v.up.weight) for k, v in loras.items()}
    first: list[tuple[str, str]] = []
    failed = _attach_all(loras, target, include, exclude, first)
    if debug_map is not None:
        debug_map += first
    if failed or len(first) != len(loras):
        raise ValueError(f"sanity check failed: {len(first)} / {len(loras)} LoRA layers attached, {len(failed)} failed")
    second: list[tuple[str, str]] = []
    skipped = _attach_all(clones, target, include, exclude, second)
    if second or len(skipped) != len(loras):
        raise ValueError(f"sanity check failed: {len(second)} / {len(loras)} LoRA layers attached twice, {len(skipped)} skipped")
    return failed
