"""The Chain tree: host-side mirror of refiners' fluxion interpreter, written from scratch for this project.

What is mirrored (so that refiners user code keeps working when `refiners.fluxion.layers` is swapped for
`refiners_amd.fluxion.layers`), with the reference location of each behaviour:

* `Chain` calls its children in order, splatting tuple results into the next child's arguments, then re-applies its
  `init_context()`                                            (reference src/refiners/fluxion/layers/chain.py:245-257)
* children are registered under `ClassName` / `ClassName_<i>` keys, which is what makes state-dict keys and
  `chain.layer(("DownBlocks", 4, ...))` paths identical to the reference's              (chain.py:19-38)
* the context side channel: a dict of dicts per Chain, pushed down to every sub-Chain by aliasing when the child has
  no entry of that name and by `dict.update` when it has one       (src/refiners/fluxion/context.py:9-46, chain.py:131-156)
* tree surgery: insert / append / pop / remove / replace / structural_copy (leaves shared)   (chain.py:485-639)
* combinators Parallel / Distribute / Passthrough / Sum / Residual / Concatenate / Matmul    (chain.py:756-1006)
* `UseContext(...).compose(f)`, `SetContext(..., callback=)`, `Lambda`                      (chain.py:645-753)
* the tree `repr` (tags, folding of identical siblings, `#n` numbering, depth 7)   (layers/module.py:267-378)

One addition the reference does not have: every structural mutation bumps `tree_epoch()`, a process-wide counter the
MI355X engine (refiners_amd/engine) uses to know that a compiled launch plan is stale.
"""
from __future__ import annotations

import contextlib
import inspect
import re
import sys
import traceback
from collections import Counter
from pathlib import Path
from types import ModuleType
from typing import Any, Callable, Iterable, Iterator, Sequence, TypeVar

import torch
from torch import Tensor, nn

T = TypeVar("T", bound="Module")

_EPOCH = [0]


def tree_epoch() -> int:
    """Monotonic counter of structural mutations / scale changes of any tree in this process."""
    return _EPOCH[0]


def bump_epoch() -> None:
    _EPOCH[0] += 1


# ------------------------------------------------------------------------------------------------ context store
Context = dict[str, Any]
Contexts = dict[str, Context]


class ContextProvider:
    """Named dictionaries shared along a Chain tree (reference: fluxion/context.py:9-46)."""

    def __init__(self) -> None:
        self.contexts: Contexts = {}

    @staticmethod
    def create(contexts: Contexts) -> "ContextProvider":
        p = ContextProvider()
        p.update_contexts(contexts)
        return p

    def set_context(self, key: str, value: Context) -> None:
        self.contexts[key] = value

    def get_context(self, key: str) -> Any:
        return self.contexts.get(key)

    def update_contexts(self, new_contexts: Contexts) -> None:
        # a missing entry ALIASES the incoming dict (that is how a whole subtree comes to share one store); an
        # existing entry is merged in place
        for name, ctx in new_contexts.items():
            mine = self.contexts.get(name)
            if mine is None and name not in self.contexts:
                self.contexts[name] = ctx
            else:
                self.contexts[name].update(ctx)

    def __repr__(self) -> str:
        def show(v: Any) -> str:
            if isinstance(v, Tensor):
                return f"Tensor(shape={v.shape}, dtype={v.dtype}, device={v.device})"
            return repr(v)

        body = {name: {k: show(v) for k, v in ctx.items()} for name, ctx in self.contexts.items()}
        return f"ContextProvider(contexts={body})"


# ------------------------------------------------------------------------------------------------ module bases
_BASIC = (str, float, int, bool)


class Module(nn.Module):
    """torch.nn.Module with the printing helpers fluxion models rely on (reference: layers/module.py:23-150)."""

    _tag: str = ""

    def __init__(self, *args: Any, **kwargs: Any) -> None:
        super().__init__(*args, **kwargs)

    def load_from_safetensors(self: T, tensors_path: str | Path, strict: bool = True) -> T:
        from safetensors.torch import load_file

        self.load_state_dict(load_file(str(tensors_path)), strict=strict)
        return self

    def load_state_dict(self, *args: Any, **kwargs: Any) -> Any:  # type: ignore[override]
        # weights may be replaced (assign=True swaps the Parameter objects) or rewritten in place: either way every program
        # lowered from a tree that contains this module holds stale converted / merged / K-blocked copies
        out = super().load_state_dict(*args, **kwargs)
        bump_epoch()
        return out

    def to(self: T, device: Any = None, dtype: Any = None) -> T:  # type: ignore[override]
        bump_epoch()  # parameters move / change dtype
        return super().to(device=device, dtype=dtype)  # type: ignore[return-value]

    # -- printing ------------------------------------------------------------------------------------------
    def basic_attributes(self, init_attrs_only: bool = False) -> dict[str, Any]:
        params = inspect.signature(self.__init__).parameters
        defaults = {k: p.default for k, p in params.items() if p.default is not inspect.Parameter.empty}

        def basic(v: Any) -> bool:
            return isinstance(v, _BASIC) or (isinstance(v, Sequence) and all(isinstance(e, _BASIC) for e in v))

        out: dict[str, Any] = {}
        for k, v in self.__dict__.items():
            if k.startswith("_") or not basic(v):
                continue
            if init_attrs_only and (k not in params or k == "self" or v == defaults.get(k)):
                continue
            out[k] = v
        return out

    def __str__(self) -> str:
        attrs = ", ".join(f"{k}={v}" for k, v in self.basic_attributes(init_attrs_only=True).items())
        return f"{self.__class__.__name__}({attrs})"

    def __repr__(self) -> str:
        return render_tree(self, depth=7)

    def pretty_print(self, depth: int = -1) -> None:
        print(render_tree(self, depth=depth))

    def _show_only_tag(self) -> bool:
        return False

    def get_path(self, parent: "Chain | None" = None, top: "Module | None" = None) -> str:
        if parent is None or self is top:
            return self.__class__.__name__
        for key, child in parent._modules.items():
            if child is self:
                return parent.get_path(parent=parent.parent, top=top) + "." + key
        raise ValueError(f"{self} not found in {parent}")


class ContextModule(Module):
    """A module that knows its parent Chain and reads contexts through it (reference: layers/module.py:153-235)."""

    _can_refresh_parent: bool = True

    def __init__(self, *args: Any, **kwargs: Any) -> None:
        super().__init__(*args, **kwargs)
        self._parent: list[Chain] = []  # boxed so torch does not register the parent as a submodule

    @property
    def parent(self) -> "Chain | None":
        return self._parent[0] if self._parent else None

    @property
    def ensure_parent(self) -> "Chain":
        assert self._parent, "module does not have a parent"
        return self._parent[0]

    def get_parents(self) -> "list[Chain]":
        out: list[Chain] = []
        p = self.parent
        while p is not None:
            out.append(p)
            p = p.parent
        return out

    def _set_parent(self, parent: "Chain | None") -> None:
        if not self._can_refresh_parent:
            return
        if parent is None:
            self._parent = []
        else:
            assert any(m is self for m in parent), f"{self} not in {parent}"
            self._parent = [parent]

    @contextlib.contextmanager
    def no_parent_refresh(self) -> Iterator[None]:
        saved = self._can_refresh_parent
        self._can_refresh_parent = False
        try:
            yield
        finally:
            self._can_refresh_parent = saved

    @property
    def provider(self) -> ContextProvider:
        return self.ensure_parent.provider

    def use_context(self, context_name: str) -> Context:
        ctx = self.provider.get_context(context_name)
        assert ctx is not None, f"Context {context_name} not found."
        return ctx

    def structural_copy(self: T) -> T:
        clone = object.__new__(self.__class__)
        for k, v in self.__dict__.items():
            if k.startswith("_"):
                continue
            mod = sys.modules.get(type(v).__module__)
            if isinstance(mod, ModuleType) and "torch" not in mod.__name__:
                object.__setattr__(clone, k, v)
        ContextModule.__init__(clone)
        return clone

    def get_path(self, parent: "Chain | None" = None, top: "Module | None" = None) -> str:
        return super().get_path(parent=parent or self.parent, top=top)


class WeightedModule(Module):
    """A leaf with a `weight` tensor (reference: layers/module.py:238-264)."""

    weight: Tensor

    @property
    def device(self) -> torch.device:
        return self.weight.device

    @property
    def dtype(self) -> torch.dtype:
        return self.weight.dtype

    def __str__(self) -> str:
        head = super().__str__().removesuffix(")")
        return f"{head}, device={self.device}, dtype={str(self.dtype).removeprefix('torch.')})"


# ------------------------------------------------------------------------------------------------ tree printing
def _node(module: nn.Module) -> dict[str, Any]:
    if isinstance(module, Module):
        tag = module._tag
        if not tag:
            value = str(module)
        elif module._show_only_tag():
            value = f"({tag})"
        else:
            value = f"({tag}) {module}"
        kids = [_node(c) for c in module.children()]
    else:
        value, kids = str(module), []
    return {"value": value, "class_name": module.__class__.__name__, "children": kids}


def _fold(node: dict[str, Any]) -> None:
    kids = node["children"]
    i = 0
    while i < len(kids):
        j = i + 1
        while j < len(kids) and kids[j] == kids[i]:
            j += 1
        if j - i > 1:
            kids[i]["value"] += f" (x{j - i})"
            del kids[i + 1 : j]
        _fold(kids[i])
        i += 1


def _render(node: dict[str, Any], prefix: str, last: bool, root: bool, depth: int) -> str:
    if depth == 0 and node["children"]:
        return f"{prefix}{'└── ' if last else '├── '}{node['value']} ..."
    if depth > 0:
        depth -= 1
    lines = [f"{prefix}{'' if root else ('└── ' if last else '├── ')}{node['value']}"]
    totals = Counter(c["class_name"] for c in node["children"])
    seen: Counter[str] = Counter()
    pad = prefix + ("    " if last else "│   ")
    n = len(node["children"])
    for i, c in enumerate(node["children"]):
        seen[c["class_name"]] += 1
        value = f"{c['value']} #{seen[c['class_name']]}" if totals[c["class_name"]] > 1 else c["value"]
        lines.append(_render({**c, "value": value}, pad, i == n - 1, False, depth))
    return "\n".join(lines)


def render_tree(module: nn.Module, depth: int = 7) -> str:
    root = _node(module)
    _fold(root)
    return _render(root, "", True, True, depth)


# ------------------------------------------------------------------------------------------------ Chain
def unique_child_names(modules: Sequence[nn.Module]) -> dict[str, nn.Module]:
    """`ClassName` if that class occurs once among the siblings, else `ClassName_<1-based index>` (chain.py:19-38)."""
    totals = Counter(m.__class__.__name__ for m in modules)
    seen: Counter[str] = Counter()
    named: dict[str, nn.Module] = {}
    for m in modules:
        cn = m.__class__.__name__
        seen[cn] += 1
        named[f"{cn}_{seen[cn]}" if totals[cn] > 1 else cn] = m
    return named


class ChainError(RuntimeError):
    """Raised by a Chain when one of its children fails; carries the position in the tree and the argument summary."""


def _summarize(x: Any) -> str:
    if not isinstance(x, Tensor):
        return repr(x)
    info = [f"shape=({', '.join(map(str, x.shape))})", f"dtype={str(x.dtype).removeprefix('torch.')}", f"device={x.device}"]
    if x.numel() and x.device.type != "meta" and not x.is_complex():
        f = x.float()
        info += [f"min={f.min():.2f}", f"max={f.max():.2f}", f"mean={f.mean():.2f}"]
    return "Tensor(" + ", ".join(info) + ")"


def _flatten_args(args: Any) -> list[Any]:
    if isinstance(args, tuple):
        return [leaf for a in args for leaf in _flatten_args(a)]
    return [args]


class Chain(ContextModule):
    """Sequential composition with a context store; the building block of every model in this package."""

    _tag = "CHAIN"

    def __init__(self, *args: nn.Module | Iterable[nn.Module]) -> None:
        super().__init__()
        self._provider = ContextProvider()
        if len(args) == 1 and isinstance(args[0], Iterable) and not isinstance(args[0], Chain):
            modules = tuple(args[0])
        else:
            modules = tuple(args)  # type: ignore[assignment]
        for m in modules:
            ok = (
                not isinstance(m, ContextModule)
                or not m._can_refresh_parent
                or m.parent is None
                or m.parent is self
            )
            assert ok, f"{m.__class__.__name__} already has parent {m.parent.__class__.__name__}"  # type: ignore[union-attr]
        self._rename_children(modules)
        self._reset_context()
        for m in self:
            if isinstance(m, ContextModule) and m.parent is not self:
                m._set_parent(self)

    def __setattr__(self, name: str, value: Any) -> None:
        if isinstance(value, nn.Module):
            raise ValueError(
                "Chain does not support setting modules by attribute. Instead, use a mutation method like `append` or"
                " wrap it within a single element list to prevent pytorch from registering it as a submodule."
            )
        super().__setattr__(name, value)

    # -- context -------------------------------------------------------------------------------------------
    @property
    def provider(self) -> ContextProvider:
        return self._provider

    def init_context(self) -> Contexts:
        return {}

    def _register_provider(self, context: Contexts | None = None) -> None:
        if context:
            self._provider.update_contexts(context)
        mine = self._provider.contexts
        for m in self._modules.values():
            if isinstance(m, Chain):
                m._register_provider(mine)

    def _reset_context(self) -> None:
        self._register_provider(self.init_context())

    def set_context(self, context: str, value: Any) -> None:
        self._provider.set_context(context, value)
        self._register_provider()

    # -- execution -----------------------------------------------------------------------------------------
    def _call_layer(self, layer: nn.Module, name: str, /, *args: Any) -> Any:
        try:
            return layer(*args)
        except Exception as exc:  # noqa: BLE001 -- every failure is re-raised as a located ChainError
            raise ChainError(self._describe_failure(exc, name, args)) from None

    def _describe_failure(self, exc: Exception, name: str, args: tuple[Any, ...]) -> str:
        frames = traceback.extract_tb(exc.__traceback__)
        noise = (r"torch/nn/modules/", r"torch/nn/functional\.py", r"fluxion/tree\.py")
        kept = [f for f in frames if not any(re.search(p, f.filename) for p in noise) and not f.name.startswith("_")]
        where = "".join(traceback.format_list(kept))
        text = re.sub(r"\n\s*\n", "\n", str(exc))
        excerpt = self._show_error_in_tree(name)
        shown = "\n".join(f"{i}: {_summarize(a)}" for i, a in enumerate(_flatten_args(args)))
        msg = f"{where}\n{text}\n---------------\n{excerpt}\n{shown}"
        if "Error" not in text:
            msg = f"{type(exc).__name__}:\n {msg}"
        return msg

    def _show_error_in_tree(self, name: str, /, max_lines: int = 20) -> str:
        """Depth-3 rendering of this Chain with the child registered under `name` flagged `>>> ... | <state-dict path>`,
        cut to `max_lines` lines around the flag (reference chain.py:158-188)."""
        root = _node(self)
        _fold(root)
        top = (self.get_parents() or [self])[-1]
        prefix = next((k for k, m in top.named_modules() if m is self), "")
        cls, _, ordinal = name.rpartition("_") if "_" in name else (name, "", "1")
        seen = 0
        for child in root["children"]:
            if child["class_name"] == cls:
                seen += 1
                if ordinal.isdigit() and seen == int(ordinal):
                    child["value"] = f">>> {child['value']} | {'.'.join(p for p in (prefix, name) if p)}"
                    break
        lines = _render(root, "", True, True, 3).split("\n")
        at = next((i for i, ln in enumerate(lines) if ">>> " in ln), 0)
        return "\n".join(lines[max(0, at - max_lines // 2) : min(len(lines), at + max_lines // 2 + 1)])

    def forward(self, *args: Any) -> Any:
        result: Any = None
        flowing: tuple[Any, ...] = args
        for name, layer in self._modules.items():
            result = self._call_layer(layer, name, *flowing)
            flowing = result if isinstance(result, tuple) else (result,)
        self._reset_context()
        return result

    # -- container protocol --------------------------------------------------------------------------------
    def _rename_children(self, modules: Iterable[nn.Module]) -> None:
        self._modules = unique_child_names(tuple(modules))  # type: ignore[assignment]
        bump_epoch()

    def __getitem__(self, key: int | str | slice) -> Any:
        if isinstance(key, slice):
            clone = self.structural_copy()
            clone._rename_children(list(clone)[key])
            return clone
        if isinstance(key, str):
            return self._modules[key]
        return list(self._modules.values())[key]

    def __iter__(self) -> Iterator[nn.Module]:
        return iter(self._modules.values())

    def __len__(self) -> int:
        return len(self._modules)

    def __contains__(self, module: object) -> bool:
        return any(m is module for m in self._modules.values())

    @property
    def device(self) -> torch.device | None:
        wm = self.find(WeightedModule)
        return None if wm is None else wm.device

    @property
    def dtype(self) -> torch.dtype | None:
        wm = self.find(WeightedModule)
        return None if wm is None else wm.dtype

    # -- search --------------------------------------------------------------------------------------------
    def walk(
        self, predicate: type | Callable[[nn.Module, "Chain"], bool] | None = None, recurse: bool = False
    ) -> Iterator[tuple[Any, "Chain"]]:
        """Depth-first (module, parent) pairs matching `predicate`; matched Chains are not entered unless `recurse`."""
        if predicate is not None and not isinstance(predicate, type) and hasattr(predicate, "__origin__"):
            raise ValueError("subscripted generics cannot be used as predicates")
        if isinstance(predicate, type):
            cls = predicate
            test: Callable[[nn.Module, Chain], bool] = lambda m, _p: isinstance(m, cls)
        elif predicate is None:
            test = lambda _m, _p: True
        else:
            test = predicate
        return self._walk(test, recurse)

    def _walk(self, test: Callable[[nn.Module, "Chain"], bool], recurse: bool) -> Iterator[tuple[nn.Module, "Chain"]]:
        for m in list(self._modules.values()):
            try:
                hit = test(m, self)
            except StopIteration:
                continue
            if hit:
                yield m, self
                if not recurse:
                    continue
            if isinstance(m, Chain):
                yield from m._walk(test, recurse)

    def layers(self, layer_type: type[T], recurse: bool = False) -> Iterator[T]:
        for m, _ in self.walk(layer_type, recurse):
            yield m

    def find(self, layer_type: type[T]) -> T | None:
        return next(self.layers(layer_type), None)

    def ensure_find(self, layer_type: type[T]) -> T:
        found = self.find(layer_type)
        assert found is not None, f"could not find {layer_type} in {self}"
        return found

    def layer(self, key: str | int | Sequence[str | int], layer_type: type[T] = Module) -> T:  # type: ignore[assignment]
        if isinstance(key, (str, int)):
            got = self[key]
            assert isinstance(got, layer_type), f"layer {key} is {type(got)}, not {layer_type}"
            return got
        if len(key) == 0:
            assert isinstance(self, layer_type), f"layer is {type(self)}, not {layer_type}"
            return self  # type: ignore[return-value]
        if len(key) == 1:
            return self.layer(key[0], layer_type)
        return self.layer(key[0], Chain).layer(key[1:], layer_type)

    def find_parent(self, module: nn.Module) -> "Chain | None":
        if module in self:
            return self
        for _, p in self.walk(lambda m, _p: m is module):
            return p
        return None

    def ensure_find_parent(self, module: nn.Module) -> "Chain":
        p = self.find_parent(module)
        assert p is not None, f"could not find {module} in {self}"
        return p

    # -- surgery -------------------------------------------------------------------------------------------
    def insert(self, index: int, module: nn.Module) -> None:
        kids = list(self)
        if index < 0:
            index = max(0, len(kids) + index + 1)
        kids.insert(index, module)
        self._rename_children(kids)
        if isinstance(module, ContextModule):
            module._set_parent(self)
        self._register_provider()

    def insert_before_type(self, module_type: type, new_module: nn.Module) -> None:
        for i, m in enumerate(self):
            if isinstance(m, module_type):
                return self.insert(i, new_module)
        raise ValueError(f"No module of type {module_type.__name__} found in the chain.")

    def insert_after_type(self, module_type: type, new_module: nn.Module) -> None:
        for i, m in enumerate(self):
            if isinstance(m, module_type):
                return self.insert(i + 1, new_module)
        raise ValueError(f"No module of type {module_type.__name__} found in the chain.")

    def append(self, module: nn.Module) -> None:
        self.insert(-1, module)

    def pop(self, index: int = -1) -> nn.Module:
        kids = list(self)
        if index < 0:
            index += len(kids)
        if not 0 <= index < len(kids):
            raise IndexError("Index out of range.")
        gone = kids.pop(index)
        if isinstance(gone, ContextModule):
            gone._set_parent(None)
        self._rename_children(kids)
        return gone

    def remove(self, module: nn.Module) -> None:
        kids = list(self)
        at = next((i for i, m in enumerate(kids) if m is module), None)
        if at is None:
            raise ValueError(f"{module} is not in {self}")
        del kids[at]
        self._rename_children(kids)
        if isinstance(module, ContextModule):
            module._set_parent(None)

    def replace(self, old_module: nn.Module, new_module: nn.Module, old_module_parent: "Chain | None" = None) -> None:
        kids = list(self)
        at = next((i for i, m in enumerate(kids) if m is old_module), None)
        if at is None:
            raise ValueError(f"{old_module} is not in {self}")
        kids[at] = new_module
        self._rename_children(kids)
        if isinstance(new_module, ContextModule):
            new_module._set_parent(self)
        if isinstance(old_module, ContextModule):
            old_module._set_parent(old_module_parent)
        self._register_provider()

    def structural_copy(self: T) -> T:
        """Duplicate the inner nodes of the tree, share the leaves (and therefore the weights)."""
        hook = getattr(self, "_pre_structural_copy", None)
        if callable(hook):
            hook()
        kids = [m.structural_copy() if isinstance(m, ContextModule) else m for m in self]  # type: ignore[attr-defined]
        clone = super().structural_copy()  # type: ignore[misc]
        clone._provider = ContextProvider.create(clone.init_context())
        for m in kids:
            clone.append(m)
        hook = getattr(clone, "_post_structural_copy", None)
        if callable(hook):
            hook(self)
        return clone

    def _show_only_tag(self) -> bool:
        return self.__class__ is Chain


# ------------------------------------------------------------------------------------------------ context leaves
class UseContext(ContextModule):
    """Returns `func(contexts[context][key])`, ignoring its inputs (chain.py:645-675)."""

    def __init__(self, context: str, key: str) -> None:
        super().__init__()
        self.context = context
        self.key = key
        self.func: Callable[[Any], Any] = lambda x: x

    def __call__(self, *args: Any) -> Any:
        ctx = self.use_context(self.context)
        assert ctx, f"context {self.context} is unset"
        value = ctx.get(self.key)
        assert value is not None, f"context entry {self.context}.{self.key} is unset"
        return self.func(value)

    def compose(self, func: Callable[[Any], Any]) -> "UseContext":
        self.func = func
        return self

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(context={self.context!r}, key={self.key!r})"


class SetContext(ContextModule):
    """Stores its input under contexts[context][key] (or hands (current, input) to `callback`); returns the input."""

    def __init__(self, context: str, key: str, callback: Callable[[Any, Any], Any] | None = None) -> None:
        super().__init__()
        self.context = context
        self.key = key
        self.callback = callback

    def __call__(self, x: Tensor) -> Tensor:
        ctx = self.use_context(self.context)
        if ctx:
            if self.callback is None:
                ctx[self.key] = x
            else:
                self.callback(ctx[self.key], x)
        return x

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(context={self.context!r}, key={self.key!r})"


class Lambda(Module):
    def __init__(self, func: Callable[..., Any]) -> None:
        super().__init__()
        self.func = func

    def forward(self, *args: Any) -> Any:
        return self.func(*args)

    def __str__(self) -> str:
        name = getattr(self.func, "__name__", "partial_function")
        return f"Lambda({name}{inspect.signature(self.func)})"


# ------------------------------------------------------------------------------------------------ combinators
class Parallel(Chain):
    """Every child gets the same inputs; returns the tuple of results."""

    _tag = "PAR"

    def forward(self, *args: Any) -> tuple[Any, ...]:
        return tuple(self._call_layer(m, k, *args) for k, m in self._modules.items())

    def _show_only_tag(self) -> bool:
        return self.__class__ is Parallel


class Distribute(Chain):
    """Child i gets input i; returns the tuple of results."""

    _tag = "DISTR"

    def forward(self, *args: Any) -> tuple[Any, ...]:
        n, m = len(args), len(self._modules)
        assert n == m, f"Number of positional arguments ({n}) must match number of sub-modules ({m})."
        return tuple(self._call_layer(mod, k, a) for a, (k, mod) in zip(args, self._modules.items()))

    def _show_only_tag(self) -> bool:
        return self.__class__ is Distribute


class Passthrough(Chain):
    """Runs its children for their side effects and returns its inputs unchanged."""

    _tag = "PASS"

    def forward(self, *inputs: Any) -> Any:
        super().forward(*inputs)
        return inputs

    def _show_only_tag(self) -> bool:
        return self.__class__ is Passthrough


class Sum(Chain):
    """Adds the results of its children (each fed the same inputs), left to right."""

    _tag = "SUM"

    def forward(self, *inputs: Any) -> Any:
        total: Any = None
        for m in self:
            y = m(*inputs)
            if isinstance(y, tuple):
                y = sum(y)
            total = y if total is None else total + y
        return total

    def _show_only_tag(self) -> bool:
        return self.__class__ is Sum


class Residual(Chain):
    """chain(x) + x."""

    _tag = "RES"

    def forward(self, *inputs: Any) -> Any:
        assert len(inputs) == 1, "Residual connection can only be used with a single input."
        return super().forward(*inputs) + inputs[0]


class Concatenate(Chain):
    _tag = "CAT"

    def __init__(self, *modules: nn.Module, dim: int = 0) -> None:
        super().__init__(*modules)
        self.dim = dim

    def forward(self, *args: Any) -> Tensor:
        parts = [m(*args) for m in self]
        return torch.cat([p for p in parts if p is not None], dim=self.dim)

    def _show_only_tag(self) -> bool:
        return self.__class__ is Concatenate


class Matmul(Chain):
    _tag = "MATMUL"

    def __init__(self, input: nn.Module, other: nn.Module) -> None:
        super().__init__(input, other)

    def forward(self, *args: Tensor) -> Tensor:
        return torch.matmul(self[0](*args), self[1](*args))


class ReturnException(Exception):
    def __init__(self, value: Tensor):
        self.value = value


class Return(Module):
    def forward(self, x: Tensor) -> None:
        raise ReturnException(x)
