"""Host-side mirror of refiners' T2I-Adapter (`foundationals/latent_diffusion/t2i_adapter.py:20-220`,
`stable_diffusion_xl/t2i_adapter.py:9-49`) -- SURVEY.md section 8(f) next-4.

Two halves with very different costs:
  * `ConditionEncoderXL`: a small conv net that turns the conditioning picture (depth, canny, ...) into four feature maps,
    ONCE per image (`compute_condition_features`); it stays on torch.
  * `T2IFeatures`: four `x + scale * features[i]` nodes inside the UNet (three down blocks and the middle block), hit on
    EVERY denoising step; the engine lowers them to one fused add each (refiners_amd/engine/unet_lowering.py).
State-dict keys of the encoder equal the reference's (tests/golden/t2i_keys.json).
"""
from __future__ import annotations

from typing import Any, Optional

from torch import Tensor, nn

from ..fluxion import layers as fl
from ..fluxion.adapters import Adapter
from ..fluxion.tree import Module, bump_epoch
from .blocks import ResidualAccumulator


class PixelUnshuffle(nn.PixelUnshuffle, Module):
    """fl.PixelUnshuffle (`fluxion/layers/pixelshuffle.py:6-20`)."""

    def __init__(self, downscale_factor: int) -> None:
        nn.PixelUnshuffle.__init__(self, downscale_factor=downscale_factor)


class Downsample2d(nn.AvgPool2d, Module):
    def __init__(self, scale_factor: int) -> None:
        nn.AvgPool2d.__init__(self, kernel_size=scale_factor, stride=scale_factor)


class ResidualBlock(fl.Residual):
    """x + conv1x1(relu(conv3x3(x)))"""

    def __init__(self, channels: int, device: Any = None, dtype: Any = None) -> None:
        super().__init__(
            fl.Conv2d(channels, channels, kernel_size=3, padding=1, device=device, dtype=dtype),
            fl.ReLU(),
            fl.Conv2d(channels, channels, kernel_size=1, device=device, dtype=dtype),
        )


class ResidualBlocks(fl.Chain):
    def __init__(self, in_channels: int, out_channels: int, num_residual_blocks: int = 2, downsample: bool = False, device: Any = None, dtype: Any = None) -> None:
        pre = Downsample2d(scale_factor=2) if downsample else fl.Identity()
        shortcut = fl.Conv2d(in_channels, out_channels, kernel_size=1, device=device, dtype=dtype) if in_channels != out_channels else fl.Identity()
        super().__init__(pre, shortcut, fl.Chain(ResidualBlock(out_channels, device=device, dtype=dtype) for _ in range(num_residual_blocks)))


class StatefulResidualBlocks(fl.Chain):
    """ResidualBlocks whose output is also appended to context "t2iadapter".features."""

    def __init__(self, in_channels: int, out_channels: int, num_residual_blocks: int = 2, downsample: bool = False, device: Any = None, dtype: Any = None) -> None:
        super().__init__(
            ResidualBlocks(in_channels, out_channels, num_residual_blocks, downsample=downsample, device=device, dtype=dtype),
            fl.SetContext(context="t2iadapter", key="features", callback=self.push),
        )

    def push(self, features: list[Tensor], x: Tensor) -> None:
        features.append(x)


class ConditionEncoder(fl.Chain):
    """SD1.5 geometry: pixel-unshuffle by 8, four stages, each later one halving the resolution."""

    def __init__(self, in_channels: int = 3, channels: tuple[int, int, int, int] = (320, 640, 1280, 1280), num_residual_blocks: int = 2,
                 downscale_factor: int = 8, scale: float = 1.0, device: Any = None, dtype: Any = None) -> None:
        self.scale = scale
        kw = dict(device=device, dtype=dtype)
        super().__init__(
            PixelUnshuffle(downscale_factor=downscale_factor),
            fl.Conv2d(in_channels * downscale_factor ** 2, channels[0], kernel_size=3, padding=1, **kw),
            StatefulResidualBlocks(channels[0], channels[0], num_residual_blocks, **kw),
            *(StatefulResidualBlocks(channels[i - 1], channels[i], num_residual_blocks, downsample=True, **kw) for i in range(1, len(channels))),
            fl.UseContext(context="t2iadapter", key="features"),
        )

    def init_context(self) -> dict[str, dict[str, Any]]:
        return {"t2iadapter": {"features": []}}


class ConditionEncoderXL(ConditionEncoder, fl.Chain):
    """SDXL geometry: pixel-unshuffle by 16; only the third stage downsamples (`t2i_adapter.py:132-163`)."""

    def __init__(self, in_channels: int = 3, channels: tuple[int, int, int, int] = (320, 640, 1280, 1280), num_residual_blocks: int = 2,
                 downscale_factor: int = 16, scale: float = 1.0, device: Any = None, dtype: Any = None) -> None:
        self.scale = scale
        kw = dict(device=device, dtype=dtype)
        fl.Chain.__init__(
            self,
            PixelUnshuffle(downscale_factor=downscale_factor),
            fl.Conv2d(in_channels * downscale_factor ** 2, channels[0], kernel_size=3, padding=1, **kw),
            StatefulResidualBlocks(channels[0], channels[0], num_residual_blocks, **kw),
            StatefulResidualBlocks(channels[0], channels[1], num_residual_blocks, **kw),
            StatefulResidualBlocks(channels[1], channels[2], num_residual_blocks, downsample=True, **kw),
            StatefulResidualBlocks(channels[2], channels[3], num_residual_blocks, **kw),
            fl.UseContext(context="t2iadapter", key="features"),
        )


class T2IFeatures(fl.Residual):
    """x + scale * features[index], features = context "t2iadapter".condition_features_<name> (a tuple of NCHW maps)."""

    def __init__(self, name: str, index: int, scale: float = 1.0) -> None:
        self.name = name
        self.index = index
        self.scale = scale
        super().__init__(fl.UseContext(context="t2iadapter", key=f"condition_features_{name}").compose(func=lambda features: self.scale * features[self.index]))

    def __setattr__(self, key: str, value: Any) -> None:
        if key == "scale" and getattr(self, "scale", value) != value:
            bump_epoch()  # a compiled program bakes the scale in: changing it re-lowers, like Multiply.scale / Lora.scale
        super().__setattr__(key, value)


class T2IAdapter(fl.Chain, Adapter[fl.Chain]):
    def __init__(self, target: fl.Chain, name: str, condition_encoder: ConditionEncoder, weights: Optional[dict[str, Tensor]] = None) -> None:
        self.name = name
        if weights is not None:
            condition_encoder.load_state_dict(weights)
        self._condition_encoder = [condition_encoder]
        with self.setup_adapter(target):
            super().__init__(target)

    @property
    def condition_encoder(self) -> ConditionEncoder:
        return self._condition_encoder[0]

    def compute_condition_features(self, condition: Tensor) -> tuple[Tensor, ...]:
        return self.condition_encoder(condition)

    def set_condition_features(self, features: tuple[Tensor, ...]) -> None:
        self.set_context("t2iadapter", {f"condition_features_{self.name}": features})

    @property
    def scale(self) -> float:
        return self._features[0].scale

    @scale.setter
    def scale(self, value: float) -> None:
        for f in self._features:
            f.scale = value

    def init_context(self) -> dict[str, dict[str, Any]]:
        return {"t2iadapter": {f"condition_features_{self.name}": None}}

    def structural_copy(self) -> "T2IAdapter":
        raise RuntimeError("T2I-Adapter cannot be copied, eject it first.")


class SDXLT2IAdapter(T2IAdapter):
    """Features 0..2 go in front of the ResidualAccumulator of DownBlocks 3 / 5 / 8, feature 3 at the end of the MiddleBlock."""

    def __init__(self, target: fl.Chain, name: str, condition_encoder: Optional[ConditionEncoderXL] = None, scale: float = 1.0,
                 weights: Optional[dict[str, Tensor]] = None) -> None:
        self.residual_indices = (3, 5, 8)
        self._features = [T2IFeatures(name=name, index=i, scale=scale) for i in range(4)]
        super().__init__(target=target, name=name, condition_encoder=condition_encoder or ConditionEncoderXL(device=target.device, dtype=target.dtype),
                         weights=weights)

    def _check(self, block: fl.Chain) -> None:
        for layer in block.layers(T2IFeatures):
            assert layer.name != self.name, f"T2I-Adapter named {self.name} is already injected"

    def inject(self, parent: Optional[fl.Chain] = None) -> "SDXLT2IAdapter":
        for n, feat in zip(self.residual_indices, self._features):
            block = self.target.layer(("DownBlocks", n), fl.Chain)
            self._check(block)
            block.insert_before_type(ResidualAccumulator, feat)
        mid = self.target.layer("MiddleBlock", fl.Chain)
        self._check(mid)
        mid.append(self._features[-1])
        return super().inject(parent)  # type: ignore[return-value]

    def eject(self) -> None:
        for n, feat in zip(self.residual_indices, self._features):
            self.target.layer(("DownBlocks", n), fl.Chain).remove(feat)
        self.target.layer("MiddleBlock", fl.Chain).remove(self._features[-1])
        super().eject()
