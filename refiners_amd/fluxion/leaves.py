"""Leaf layers and small composite layers of the fluxion mirror.

Each class keeps the constructor signature, attribute names and parameter names of its refiners counterpart (cited per
class) because state-dict keys, `repr()` output and the adapters' pattern matching all depend on them.  The `forward`
of every leaf here is the *unfused* torch path: it is what runs on CPU, on unsupported shapes, and inside sub-trees the
MI355X engine does not recognise.  The engine (refiners_amd/engine) never calls these forwards for the sub-trees it
lowers; it reads their weights and hyper-parameters.
"""
from __future__ import annotations

import math
from enum import Enum
from typing import Any, Callable

import torch
import torch.nn.functional as F
from torch import Size, Tensor, nn

from .tree import Chain, ContextModule, Contexts, Distribute, Lambda, Module, Parallel, SetContext, UseContext, WeightedModule, bump_epoch


# ------------------------------------------------------------------------------------------------ weighted leaves
class Linear(nn.Linear, WeightedModule):
    """y = x W^T + b (reference: fluxion/layers/linear.py:9-56)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device: Any = None, dtype: Any = None) -> None:
        self.in_features = in_features
        self.out_features = out_features
        super().__init__(in_features=in_features, out_features=out_features, bias=bias, device=device, dtype=dtype)


class MultiLinear(Chain):
    def __init__(self, input_dim: int, output_dim: int, inner_dim: int, num_layers: int, device: Any = None, dtype: Any = None) -> None:
        mods: list[nn.Module] = []
        for i in range(num_layers - 1):
            mods += [Linear(input_dim if i == 0 else inner_dim, inner_dim, device=device, dtype=dtype), ReLU()]
        mods.append(Linear(inner_dim, output_dim, device=device, dtype=dtype))
        super().__init__(mods)


class Conv2d(nn.Conv2d, WeightedModule):
    """NCHW cross-correlation (reference: fluxion/layers/conv.py:6-61). `use_bias` mirrors the reference's keyword."""

    def __init__(
        self,
        in_channels: int,
        out_channels: int,
        kernel_size: int | tuple[int, int],
        stride: int | tuple[int, int] = (1, 1),
        padding: int | tuple[int, int] | str = (0, 0),
        groups: int = 1,
        use_bias: bool = True,
        dilation: int | tuple[int, int] = (1, 1),
        padding_mode: str = "zeros",
        device: Any = None,
        dtype: Any = None,
    ) -> None:
        super().__init__(
            in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
            dilation=dilation, groups=groups, bias=use_bias, padding_mode=padding_mode, device=device, dtype=dtype,
        )
        self.use_bias = use_bias


class LayerNorm(nn.LayerNorm, WeightedModule):
    """reference: fluxion/layers/norm.py:13-46."""

    def __init__(self, normalized_shape: int | list[int], eps: float = 0.00001, device: Any = None, dtype: Any = None) -> None:
        super().__init__(normalized_shape=normalized_shape, eps=eps, elementwise_affine=True, device=device, dtype=dtype)


class GroupNorm(nn.GroupNorm, WeightedModule):
    """reference: fluxion/layers/norm.py:49-93 (note the argument order: channels first)."""

    def __init__(self, channels: int, num_groups: int, eps: float = 1e-5, device: Any = None, dtype: Any = None) -> None:
        super().__init__(num_groups=num_groups, num_channels=channels, eps=eps, affine=True, device=device, dtype=dtype)
        self.channels = channels
        self.num_groups = num_groups
        self.eps = eps


class LayerNorm2d(WeightedModule):
    """Channel-wise layer norm of an NCHW tensor (reference: fluxion/layers/norm.py:96-140)."""

    def __init__(self, channels: int, eps: float = 1e-6, device: Any = None, dtype: Any = None) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(channels, device=device, dtype=dtype))
        self.bias = nn.Parameter(torch.zeros(channels, device=device, dtype=dtype))
        self.eps = eps

    def forward(self, x: Tensor) -> Tensor:
        mu = x.mean(1, keepdim=True)
        var = (x - mu).pow(2).mean(1, keepdim=True)
        return self.weight[:, None, None] * ((x - mu) / torch.sqrt(var + self.eps)) + self.bias[:, None, None]


class Parameter(WeightedModule):
    def __init__(self, *dims: int, requires_grad: bool = True, device: Any = None, dtype: Any = None) -> None:
        super().__init__()
        self.dims = dims
        self.weight = nn.Parameter(torch.randn(*dims, device=device, dtype=dtype), requires_grad=requires_grad)

    def forward(self, x: Tensor) -> Tensor:
        return self.weight.expand(x.shape[0], *self.dims)


# ------------------------------------------------------------------------------------------------ activations
class Activation(Module):
    def __init__(self) -> None:
        super().__init__()


class SiLU(Activation):
    def forward(self, x: Tensor) -> Tensor:
        return F.silu(x)


class ReLU(Activation):
    def forward(self, x: Tensor) -> Tensor:
        return F.relu(x)


class Sigmoid(Activation):
    def forward(self, x: Tensor) -> Tensor:
        return torch.sigmoid(x)


class GeLUApproximation(Enum):
    NONE = "none"
    TANH = "tanh"
    SIGMOID = "sigmoid"


class GeLU(Activation):
    """reference: fluxion/layers/activations.py:83-125 (exact erf form by default)."""

    def __init__(self, approximation: GeLUApproximation = GeLUApproximation.NONE) -> None:
        super().__init__()
        self.approximation = approximation

    def forward(self, x: Tensor) -> Tensor:
        if self.approximation is GeLUApproximation.SIGMOID:
            return x * torch.sigmoid(1.702 * x)
        return F.gelu(x, approximate=self.approximation.value)


class GLU(Activation):
    """a, g = x.chunk(2, -1); a * activation(g)  (reference: fluxion/layers/activations.py:128-160)."""

    def __init__(self, activation: Activation) -> None:
        super().__init__()
        self.activation = activation

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(activation={self.activation})"

    def forward(self, x: Tensor) -> Tensor:
        assert x.shape[-1] % 2 == 0, "Non-batch input dimension must be divisible by 2"
        a, g = x.chunk(2, dim=-1)
        return a * self.activation(g)


# ------------------------------------------------------------------------------------------------ shape plumbing
class Identity(Module):
    def __init__(self) -> None:
        super().__init__()

    def forward(self, x: Tensor) -> Tensor:
        return x


class GetArg(Module):
    def __init__(self, index: int) -> None:
        super().__init__()
        self.index = index

    def forward(self, *args: Tensor) -> Tensor:
        return args[self.index]


class Flatten(Module):
    def __init__(self, start_dim: int = 0, end_dim: int = -1) -> None:
        super().__init__()
        self.start_dim = start_dim
        self.end_dim = end_dim

    def forward(self, x: Tensor) -> Tensor:
        return torch.flatten(x, self.start_dim, self.end_dim)


class Unflatten(Module):
    def __init__(self, dim: int) -> None:
        super().__init__()
        self.dim = dim

    def forward(self, x: Tensor, sizes: Size) -> Tensor:
        return torch.unflatten(x, self.dim, sizes)


class Reshape(Module):
    """Reshapes everything but the batch dimension."""

    def __init__(self, *shape: int) -> None:
        super().__init__()
        self.shape = shape

    def forward(self, x: Tensor) -> Tensor:
        return torch.reshape(x, (x.shape[0], *self.shape))


class Transpose(Module):
    def __init__(self, dim0: int, dim1: int) -> None:
        super().__init__()
        self.dim0 = dim0
        self.dim1 = dim1

    def forward(self, x: Tensor) -> Tensor:
        return torch.transpose(x, self.dim0, self.dim1)


class Permute(Module):
    def __init__(self, *dims: int) -> None:
        super().__init__()
        self.dims = dims

    def forward(self, x: Tensor) -> Tensor:
        return torch.permute(x, self.dims)


class Squeeze(Module):
    def __init__(self, dim: int) -> None:
        super().__init__()
        self.dim = dim

    def forward(self, x: Tensor) -> Tensor:
        return torch.squeeze(x, self.dim)


class Unsqueeze(Module):
    def __init__(self, dim: int) -> None:
        super().__init__()
        self.dim = dim

    def forward(self, x: Tensor) -> Tensor:
        return torch.unsqueeze(x, self.dim)


class Slicing(Module):
    def __init__(self, dim: int = 0, start: int = 0, end: int | None = None, step: int = 1) -> None:
        super().__init__()
        self.dim, self.start, self.end, self.step = dim, start, end, step

    def forward(self, x: Tensor) -> Tensor:
        n = x.shape[self.dim]
        lo = self.start if self.start >= 0 else n + self.start
        lo = max(min(lo, n), 0)
        hi = self.end or n
        hi = hi if hi >= 0 else n + hi
        hi = max(min(hi, n), 0)
        if lo >= hi:
            shape = list(x.shape)
            shape[self.dim] = 0
            return torch.empty(*shape, device=x.device)
        return torch.index_select(x, self.dim, torch.arange(lo, hi, self.step, device=x.device))


class Multiply(Module):
    """scale * x + bias with Python-float scale/bias (reference: fluxion/layers/basics.py:385-405).

    `scale` is a live attribute (LoRA and IP-Adapter strengths are changed through it), so assigning it bumps the tree
    epoch and a compiled MI355X plan that baked the old value is rebuilt.
    """

    def __init__(self, scale: float = 1.0, bias: float = 0.0) -> None:
        super().__init__()
        self.scale = scale
        self.bias = bias

    def __setattr__(self, name: str, value: Any) -> None:
        if name in ("scale", "bias"):
            bump_epoch()
        super().__setattr__(name, value)

    def forward(self, x: Tensor) -> Tensor:
        return self.scale * x + self.bias


class Converter(ContextModule):
    """Casts its inputs to the parent Chain's device and/or dtype (reference: fluxion/layers/converter.py)."""

    def __init__(self, set_device: bool = True, set_dtype: bool = True) -> None:
        super().__init__()
        self.set_device = set_device
        self.set_dtype = set_dtype

    def forward(self, *inputs: Tensor) -> tuple[Tensor, ...]:
        parent = self.ensure_parent
        out = []
        for x in inputs:
            if self.set_device:
                assert parent.device is not None, "parent has no device"
                x = x.to(device=parent.device)
            if self.set_dtype:
                assert parent.dtype is not None, "parent has no dtype"
                x = x.to(dtype=parent.dtype)
            out.append(x)
        return tuple(out)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(set_device={self.set_device}, set_dtype={self.set_dtype})"


# ------------------------------------------------------------------------------------------------ attention
def scaled_dot_product_attention(query: Tensor, key: Tensor, value: Tensor, is_causal: bool = False) -> Tensor:
    return F.scaled_dot_product_attention(query, key, value, is_causal=is_causal)


def scaled_dot_product_attention_non_optimized(query: Tensor, key: Tensor, value: Tensor, is_causal: bool = False) -> Tensor:
    if is_causal:
        raise NotImplementedError("Causal attention for `scaled_dot_product_attention_non_optimized` is not yet implemented")
    w = torch.softmax(query @ key.transpose(-1, -2) / math.sqrt(query.shape[-1]), dim=-1)
    return w @ value


class ScaledDotProductAttention(Module):
    """Multi-head softmax(QK^T/sqrt(d))V on (B, L, H*d) tensors (reference: fluxion/layers/attentions.py:60-202)."""

    def __init__(self, num_heads: int = 1, is_causal: bool = False, is_optimized: bool = True, slice_size: int | None = None) -> None:
        super().__init__()
        self.num_heads = num_heads
        self.is_causal = is_causal
        self.is_optimized = is_optimized
        self.slice_size = slice_size
        self.dot_product = scaled_dot_product_attention if is_optimized else scaled_dot_product_attention_non_optimized

    def _heads(self, x: Tensor) -> Tensor:
        assert x.ndim == 3, f"Expected input tensor with shape (batch_size sequence_length embedding_dim), got {x.shape}"
        assert x.shape[-1] % self.num_heads == 0, f"embedding_dim {x.shape[-1]} not divisible by num_heads {self.num_heads}"
        return x.reshape(x.shape[0], x.shape[1], self.num_heads, x.shape[-1] // self.num_heads).transpose(1, 2)

    def _attend(self, query: Tensor, key: Tensor, value: Tensor) -> Tensor:
        y = self.dot_product(self._heads(query), self._heads(key), self._heads(value), is_causal=self.is_causal)
        return y.transpose(1, 2).reshape(y.shape[0], y.shape[2], self.num_heads * y.shape[-1])

    def forward(self, query: Tensor, key: Tensor, value: Tensor) -> Tensor:
        if not self.slice_size:
            return self._attend(query, key, value)
        out = torch.zeros_like(query)
        for lo in range(0, query.shape[1], self.slice_size):
            hi = min(lo + self.slice_size, query.shape[1])
            out[:, lo:hi, :] = self._attend(query[:, lo:hi, :], key, value)
        return out


class Attention(Chain):
    """Distribute(Wq, Wk, Wv) -> SDPA -> Wo (reference: fluxion/layers/attentions.py:205-316)."""

    def __init__(
        self,
        embedding_dim: int,
        num_heads: int = 1,
        key_embedding_dim: int | None = None,
        value_embedding_dim: int | None = None,
        inner_dim: int | None = None,
        use_bias: bool = True,
        is_causal: bool = False,
        is_optimized: bool = True,
        device: Any = None,
        dtype: Any = None,
    ) -> None:
        assert embedding_dim % num_heads == 0, f"embedding_dim {embedding_dim} must be divisible by num_heads {num_heads}"
        self.embedding_dim = embedding_dim
        self.num_heads = num_heads
        self.heads_dim = embedding_dim // num_heads
        self.key_embedding_dim = key_embedding_dim or embedding_dim
        self.value_embedding_dim = value_embedding_dim or embedding_dim
        self.inner_dim = inner_dim or embedding_dim
        self.use_bias = use_bias
        self.is_causal = is_causal
        self.is_optimized = is_optimized
        kw = dict(device=device, dtype=dtype)
        super().__init__(
            Distribute(
                Linear(self.embedding_dim, self.inner_dim, bias=use_bias, **kw),
                Linear(self.key_embedding_dim, self.inner_dim, bias=use_bias, **kw),
                Linear(self.value_embedding_dim, self.inner_dim, bias=use_bias, **kw),
            ),
            ScaledDotProductAttention(num_heads=num_heads, is_causal=is_causal, is_optimized=is_optimized),
            Linear(self.inner_dim, self.embedding_dim, bias=True, **kw),
        )


class SelfAttention(Attention):
    """Attention whose q, k, v inputs are the same tensor (reference: fluxion/layers/attentions.py:319-385)."""

    def __init__(
        self,
        embedding_dim: int,
        inner_dim: int | None = None,
        num_heads: int = 1,
        use_bias: bool = True,
        is_causal: bool = False,
        is_optimized: bool = True,
        device: Any = None,
        dtype: Any = None,
    ) -> None:
        super().__init__(
            embedding_dim=embedding_dim, inner_dim=inner_dim, num_heads=num_heads, use_bias=use_bias,
            is_causal=is_causal, is_optimized=is_optimized, device=device, dtype=dtype,
        )
        self.insert(0, Parallel(Identity(), Identity(), Identity()))


class SelfAttention2d(SelfAttention):
    """SelfAttention over the pixels of an NCHW tensor (reference: fluxion/layers/attentions.py:388-470)."""

    def __init__(
        self,
        channels: int,
        num_heads: int = 1,
        use_bias: bool = True,
        is_causal: bool = False,
        is_optimized: bool = True,
        device: Any = None,
        dtype: Any = None,
    ) -> None:
        assert channels % num_heads == 0, f"channels {channels} must be divisible by num_heads {num_heads}"
        self.channels = channels
        super().__init__(
            embedding_dim=channels, num_heads=num_heads, use_bias=use_bias, is_causal=is_causal,
            is_optimized=is_optimized, device=device, dtype=dtype,
        )
        self.insert(0, Lambda(self._tensor_2d_to_sequence))
        self.append(Lambda(self._sequence_to_tensor_2d))

    def init_context(self) -> Contexts:
        return {"reshape": {"height": None, "width": None}}

    def _tensor_2d_to_sequence(self, x: Tensor) -> Tensor:
        h, w = x.shape[-2:]
        self.set_context("reshape", {"height": h, "width": w})
        return x.reshape(x.shape[0], x.shape[1], h * w).transpose(1, 2)

    def _sequence_to_tensor_2d(self, x: Tensor) -> Tensor:
        h, w = self.use_context("reshape").values()
        return x.transpose(1, 2).reshape(x.shape[0], x.shape[2], h, w)


# ------------------------------------------------------------------------------------------------ resampling
def interpolate(x: Tensor, size: Size, mode: str = "nearest", antialias: bool = False) -> Tensor:
    return F.interpolate(x, size=size, mode=mode, antialias=antialias) if mode != "nearest" else F.interpolate(x, size=size, mode=mode)


class Interpolate(Module):
    def __init__(self, mode: str = "nearest", antialias: bool = False) -> None:
        super().__init__()
        self.mode = mode
        self.antialias = antialias

    def forward(self, x: Tensor, shape: Size) -> Tensor:
        return interpolate(x, shape, self.mode, self.antialias)


class Downsample(Chain):
    """Strided 3x3 conv that remembers its input H x W in context "sampling".shapes (reference: sampling.py:41-109)."""

    def __init__(self, channels: int, scale_factor: int, padding: int = 0, register_shape: bool = True, device: Any = None, dtype: Any = None):
        self.channels = channels
        self.in_channels = channels
        self.out_channels = channels
        self.scale_factor = scale_factor
        self.padding = padding
        super().__init__(Conv2d(channels, channels, kernel_size=3, stride=scale_factor, padding=padding, device=device, dtype=dtype))
        if padding == 0:
            zero_pad: Callable[[Tensor], Tensor] = lambda x: F.pad(x, (0, 1, 0, 1))
            self.insert(0, Lambda(zero_pad))
        if register_shape:
            self.insert(0, SetContext(context="sampling", key="shapes", callback=self.register_shape))

    def register_shape(self, shapes: list[Size], x: Tensor) -> None:
        shapes.append(x.shape[2:])


class Upsample(Chain):
    """Nearest interpolation to the shape popped from "sampling".shapes (or a static factor), then a 3x3 conv
    (reference: sampling.py:112-161)."""

    def __init__(self, channels: int, upsample_factor: int | None = None, device: Any = None, dtype: Any = None):
        self.channels = channels
        self.upsample_factor = upsample_factor
        shape_source = (
            Lambda(self._get_static_shape)
            if upsample_factor is not None
            else UseContext(context="sampling", key="shapes").compose(lambda shapes: shapes.pop())
        )
        super().__init__(
            Parallel(Identity(), shape_source),
            Interpolate(),
            Conv2d(channels, channels, kernel_size=3, padding=1, device=device, dtype=dtype),
        )

    def _get_static_shape(self, x: Tensor) -> Size:
        assert self.upsample_factor is not None
        return Size([s * self.upsample_factor for s in x.shape[2:]])
