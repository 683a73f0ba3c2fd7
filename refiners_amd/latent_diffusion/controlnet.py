"""Host-side mirror of refiners' SD1.5 ControlNet (`foundationals/latent_diffusion/stable_diffusion_1/controlnet.py:17-233`)
-- SURVEY.md section 8(f) next-4.  A second, separately weighted copy of the UNet's encoder half (plus a small conv net on
the conditioning picture) runs in front of the UNet and adds its thirteen scaled block outputs into "unet".residuals.

The engine lowers it (refiners_amd/engine/unet_lowering.py: the `Controlnet` branch next to control_lora -- the copied encoder runs as part of
the step program, its residual taps are GEMM launches that accumulate into the UNet's residual slots); goldens from the real reference:
tests/test_controlnet_golden.py, tests/test_engine_gpu.py.
"""
from __future__ import annotations

from typing import Any, Callable, Optional

from torch import Tensor

from ..fluxion import layers as fl
from ..fluxion.adapters import Adapter
from ..fluxion.tree import bump_epoch
from .adapters import ConditionEncoder
from .blocks import RangeAdapter2d, ResidualBlock
from .sd1 import DownBlocks, MiddleBlock, TimestepEncoder


class Controlnet(fl.Passthrough):
    """Passthrough(TimestepEncoder', Slicing(:4), DownBlocks', MiddleBlock'): returns its input untouched; its effect is
    residuals[n] += scale * scale_decay^(12 - n) * zero_conv_n(block_n output) for the 12 down blocks and the middle block."""

    def __init__(self, name: str, scale: float = 1.0, scale_decay: float = 1.0, device: Any = None, dtype: Any = None) -> None:
        self.name = name
        self.scale = scale
        self._scale_decay = scale_decay
        self.compute_scale_decays()
        kw = dict(device=device, dtype=dtype)
        super().__init__(
            TimestepEncoder(context_key=f"timestep_embedding_{name}", **kw),
            fl.Slicing(dim=1, end=4),  # inpainting UNets carry 9 channels; the control branch sees the 4 latent ones
            DownBlocks(in_channels=4, **kw),
            MiddleBlock(**kw),
        )
        self.layer(("DownBlocks", 0), fl.Chain).append(fl.Residual(fl.UseContext("controlnet", f"condition_{name}"), ConditionEncoder(**kw)))
        for block in self.layers(ResidualBlock):
            chain = block.layer("Chain", fl.Chain)
            RangeAdapter2d(target=chain.layer("Conv2d_1", fl.Conv2d), channels=block.out_channels, embedding_dim=1280,
                           context_key=f"timestep_embedding_{name}", **kw).inject(chain)
        for n, block in enumerate(self.layer("DownBlocks", DownBlocks)):
            width = block[0].out_channels
            assert isinstance(width, int), f"{block[0]} has no out_channels"
            block.append(fl.Passthrough(fl.Conv2d(width, width, kernel_size=1, **kw), fl.Lambda(self._store_nth_residual(n))))
        self.layer("MiddleBlock", MiddleBlock).append(fl.Passthrough(fl.Conv2d(1280, 1280, kernel_size=1, **kw), fl.Lambda(self._store_nth_residual(12))))

    def _store_nth_residual(self, n: int) -> Callable[[Tensor], Tensor]:
        def _store_residual(x: Tensor) -> Tensor:
            residuals = self.use_context("unet")["residuals"]
            residuals[n] = residuals[n] + x * self.scale * self.scale_decays[n]
            return x

        return _store_residual

    @property
    def scale_decay(self) -> float:
        return self._scale_decay

    @scale_decay.setter
    def scale_decay(self, value: float) -> None:
        self._scale_decay = value
        self.compute_scale_decays()
        bump_epoch()

    def compute_scale_decays(self) -> None:
        self.scale_decays = [self._scale_decay ** float(12 - i) for i in range(13)]


class SD1ControlnetAdapter(fl.Chain, Adapter[fl.Chain]):
    """Owns a Controlnet and, on inject, inserts it at index 0 of the SD1UNet (several may be stacked under distinct names)."""

    def __init__(self, target: fl.Chain, name: str, scale: float = 1.0, scale_decay: float = 1.0, weights: Optional[dict[str, Tensor]] = None) -> None:
        self.name = name
        controlnet = Controlnet(name=name, scale=scale, scale_decay=scale_decay, device=target.device, dtype=target.dtype)
        if weights is not None:
            controlnet.load_state_dict(weights)
        self._controlnet = [controlnet]  # kept out of torch's module registry, like the reference
        with self.setup_adapter(target):
            super().__init__(target)

    @property
    def controlnet(self) -> Controlnet:
        return self._controlnet[0]

    def inject(self, parent: Optional[fl.Chain] = None) -> "SD1ControlnetAdapter":
        present = [m for m in self.target if isinstance(m, Controlnet)]
        assert self.controlnet not in present, f"{self.controlnet} is already injected"
        assert all(m.name != self.name for m in present), f"Controlnet named {self.name} is already injected"
        self.target.insert(0, self.controlnet)
        return super().inject(parent)  # type: ignore[return-value]

    def eject(self) -> None:
        self.target.remove(self.controlnet)
        super().eject()

    def init_context(self) -> dict[str, dict[str, Any]]:
        return {"controlnet": {f"condition_{self.name}": None}}

    @property
    def scale(self) -> float:
        return self.controlnet.scale

    @scale.setter
    def scale(self, value: float) -> None:
        self.controlnet.scale = value
        bump_epoch()

    @property
    def scale_decay(self) -> float:
        return self.controlnet.scale_decay

    @scale_decay.setter
    def scale_decay(self, value: float) -> None:
        self.controlnet.scale_decay = value

    def set_controlnet_condition(self, condition: Tensor) -> None:
        self.set_context("controlnet", {f"condition_{self.name}": condition})

    def structural_copy(self) -> "SD1ControlnetAdapter":
        raise RuntimeError("Controlnet cannot be copied, eject it first.")
