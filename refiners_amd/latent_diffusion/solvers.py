"""Host-side mirror of refiners' other deterministic solvers -- SURVEY.md section 8(f) next-4:

* `Euler`      `latent_diffusion/solvers/euler.py:13-100`  (k-diffusion Euler, noise prediction; scales the model input)
* `DPMSolver`  `latent_diffusion/solvers/dpm.py:36-329`    (DPM-Solver++ 2M, `sde_variance = 0`; SD1.5's default solver)
* `LCMSolver`  `latent_diffusion/solvers/lcm.py:15-150`    (consistency jump + re-noising to an emulated DPM schedule; stochastic)

on the reference's default schedule (`solvers/solver.py:96-123, 386-416`: 1000 train steps, quadratic betas
8.5e-4 .. 1.2e-2, noise prediction).  Both are LINEAR in (x, eps, previous data estimate), so each exposes
`linear_step(step)` -- eight float coefficients -- and the whole guidance + solver update runs as one kernel
(`mi355x_cfg_linear_step`, see `refiners_amd.engine.compiled.CompiledSDXL`):

    eps  = u + cfg * (c - u)                       classifier-free guidance on the UNet's two halves
    d    = hx * x + he * eps                       the quantity the solver keeps (DPM++: the data estimate x0; Euler: eps)
    x'   = kx * x + ke * eps + kd * d + kp * hist  the update
    hist = d ;  model input of the next step = s_next * x'

`__call__` is the unfused torch path with the reference's own formulas and operation order.
"""
from __future__ import annotations

from collections import deque
from typing import Any

import numpy as np
import torch
from torch import Tensor


class _Schedule:
    """Solver.__init__ of the reference with its default parameters (solvers/solver.py:96-147)."""

    def __init__(self, num_inference_steps: int, first_inference_step: int, num_train_timesteps: int, initial_diffusion_rate: float,
                 final_diffusion_rate: float) -> None:
        self.num_inference_steps = num_inference_steps
        self.first_inference_step = first_inference_step
        self.num_train_timesteps = num_train_timesteps
        betas = torch.linspace(initial_diffusion_rate ** 0.5, final_diffusion_rate ** 0.5, num_train_timesteps) ** 2
        self.scale_factors = 1 - betas
        self.cumulative_scale_factors = torch.sqrt(self.scale_factors.cumprod(dim=0))
        self.noise_std = torch.sqrt(1.0 - self.scale_factors.cumprod(dim=0))

    @property
    def inference_steps(self) -> list[int]:
        return list(range(self.num_inference_steps))[self.first_inference_step:]

    # -- Solver.add_noise / remove_noise (solvers/solver.py:244-319): what Self-Attention Guidance degrades the latents through -----------
    def _noise_index(self, step: int) -> Any:
        """The base class indexes its train-time tables with the step's TIMESTEP (solver.py:261-263, 312-314)."""
        return self.timesteps[step]

    def sag_coefficients(self, step: int) -> tuple[float, float]:
        """(cumulative scale factor, noise std) that add_noise / remove_noise use at `step`: the two numbers mi355x_sag_degrade needs."""
        i = self._noise_index(step)
        return float(self.cumulative_scale_factors[i]), float(self.noise_std[i])

    def add_noise(self, x: Tensor, noise: Tensor, step: int) -> Tensor:
        i = self._noise_index(step)
        return self.cumulative_scale_factors[i] * x + self.noise_std[i] * noise

    def remove_noise(self, x: Tensor, noise: Tensor, step: int) -> Tensor:
        i = self._noise_index(step)
        return (x - self.noise_std[i] * noise) / self.cumulative_scale_factors[i]

    def _move(self, names: tuple[str, ...], device: Any, dtype: Any) -> None:
        for n in names:
            setattr(self, n, getattr(self, n).to(device=device, dtype=dtype))
        self.timesteps = self.timesteps.to(device=device)  # type: ignore[has-type]
        self.device, self.dtype = torch.device(device), dtype


class Euler(_Schedule):
    """x' = x + eps * (sigma_next - sigma), model input x / sqrt(sigma^2 + 1); LINSPACE timesteps (float), sigmas
    interpolated at them, a final 0 appended (euler.py:46-100)."""

    def __init__(self, num_inference_steps: int, first_inference_step: int = 0, device: Any = "cpu", dtype: torch.dtype = torch.float32,
                 num_train_timesteps: int = 1000, initial_diffusion_rate: float = 8.5e-4, final_diffusion_rate: float = 1.2e-2) -> None:
        super().__init__(num_inference_steps, first_inference_step, num_train_timesteps, initial_diffusion_rate, final_diffusion_rate)
        self.timesteps = torch.tensor(np.linspace(0, num_train_timesteps - 1, num_inference_steps), dtype=torch.float32).flip(0)
        sig = self.noise_std / self.cumulative_scale_factors
        sig = torch.tensor(np.interp(self.timesteps.numpy(), np.arange(0, len(sig)), sig.numpy()))
        self.sigmas = torch.cat([sig, torch.tensor([0.0])])
        self._move(("scale_factors", "cumulative_scale_factors", "noise_std", "sigmas"), device, dtype)

    @property
    def init_noise_sigma(self) -> Tensor:
        return self.sigmas.max()

    def _noise_index(self, step: int) -> Any:
        # Euler's timesteps are FLOATS (linspace, euler.py:46-50) and the reference's add_noise / remove_noise index integer tables with them:
        # refiners itself raises here (so Self-Attention Guidance cannot be combined with Euler in the reference either)
        raise IndexError("tensors used as indices must be long, int, byte or bool tensors (Euler's float timesteps: the reference's "
                         "Solver.add_noise / remove_noise, and with them Self-Attention Guidance, do not work with this solver)")

    def scale_model_input(self, x: Tensor, step: int) -> Tensor:
        if step == -1:
            return x * self.init_noise_sigma
        return x / ((self.sigmas[step] ** 2 + 1) ** 0.5)

    def input_scale(self, step: int) -> float:
        s = float(self.sigmas[step].double()) if step < self.num_inference_steps else 0.0
        return 1.0 / (s * s + 1.0) ** 0.5

    def linear_step(self, step: int) -> tuple[float, ...]:
        """(hx, he, kx, ke, kd, kp, s_next): d = eps is not needed by Euler, so hist stays unused (kd = kp = 0)."""
        s = self.sigmas.double().cpu()
        return (0.0, 1.0, 1.0, float(s[step + 1] - s[step]), 0.0, 0.0, self.input_scale(step + 1))

    def __call__(self, x: Tensor, predicted_noise: Tensor, step: int, generator: Any = None) -> Tensor:
        assert self.first_inference_step <= step < self.num_inference_steps, f"invalid step {step}"
        return x + predicted_noise * (self.sigmas[step + 1] - self.sigmas[step])


class DPMSolver(_Schedule):
    """DPM-Solver++ (2M), deterministic: first-order update on the first (and optionally last) step, second-order
    multistep otherwise; CUSTOM timestep spacing, sigmas interpolated at the timesteps with sigma_min appended, constants
    computed in float64 (dpm.py:58-120, 224-329)."""

    def __init__(self, num_inference_steps: int, first_inference_step: int = 0, last_step_first_order: bool = False, device: Any = "cpu",
                 dtype: torch.dtype = torch.float32, num_train_timesteps: int = 1000, initial_diffusion_rate: float = 8.5e-4,
                 final_diffusion_rate: float = 1.2e-2, timesteps_spacing: str = "custom") -> None:
        super().__init__(num_inference_steps, first_inference_step, num_train_timesteps, initial_diffusion_rate, final_diffusion_rate)
        assert timesteps_spacing in ("custom", "trailing")
        self.last_step_first_order = last_step_first_order
        # (the reference builds the train-time schedule in float32 and only then switches to float64: dpm.py:74-81)
        csf, nstd = self.cumulative_scale_factors.double(), self.noise_std.double()
        if timesteps_spacing == "custom":  # DPM's own spacing (dpm.py:112-123)
            spaced = torch.tensor(np.linspace(0, num_train_timesteps - 1, num_inference_steps + 1).round().astype(int)[1:]).flip(0)
        else:  # TRAILING (solver.py:229-232): what LCMSolver asks of the DPM solver it emulates
            spaced = torch.arange(num_train_timesteps - 1, 0, -(num_train_timesteps // num_inference_steps))
        sigmas_all = nstd / csf
        sig = torch.tensor(np.interp(spaced.numpy(), np.arange(0, len(sigmas_all)), sigmas_all.numpy()))
        self.sigmas = torch.cat([sig, sigmas_all[0:1]])
        self.cumulative_scale_factors = 1 / torch.sqrt(self.sigmas ** 2 + 1)
        self.noise_std = self.sigmas * self.cumulative_scale_factors
        self.signal_to_noise_ratios = torch.log(self.cumulative_scale_factors) - torch.log(self.noise_std)
        self.timesteps = self._timesteps_from_sigmas(sigmas_all)
        self._tables64 = (self.cumulative_scale_factors.clone(), self.noise_std.clone(), self.signal_to_noise_ratios.clone())
        self.estimated_data: deque[Tensor] = deque([torch.tensor([])] * 2, maxlen=2)
        self._move(("scale_factors", "cumulative_scale_factors", "noise_std", "sigmas", "signal_to_noise_ratios"), device, dtype)

    def _timesteps_from_sigmas(self, sigmas_all: Tensor) -> Tensor:
        """The (fractional, then rounded) train timestep whose log-sigma matches each inference sigma (dpm.py:122-141)."""
        log_all = torch.log(sigmas_all)
        out = []
        for sigma in self.sigmas[:-1]:
            dist = torch.log(sigma) - log_all.unsqueeze(1)
            low = (dist >= 0).cumsum(dim=0).argmax(dim=0).clip(max=sigmas_all.size(0) - 2)
            high = low + 1
            w = ((log_all[low] - torch.log(sigma)) / (log_all[low] - log_all[high])).clamp(0, 1)
            out.append((1 - w) * low + w * high)
        return torch.cat(out).round().int()

    def scale_model_input(self, x: Tensor, step: int) -> Tensor:
        return x

    def input_scale(self, step: int) -> float:
        return 1.0

    def _noise_index(self, step: int) -> Any:
        """DPMSolver._add_noise / remove_noise (dpm.py:171-202): the tables are indexed by inference STEP here, not by train timestep."""
        return step

    def _first_order(self, step: int) -> bool:
        return step == self.first_inference_step or (self.last_step_first_order and step == self.num_inference_steps - 1)

    def linear_step(self, step: int) -> tuple[float, ...]:
        """(hx, he, kx, ke, kd, kp, s_next) with d = x0 = (x - noise_std * eps) / scale, hist = the previous step's x0."""
        a, n, lam = (t.tolist() for t in self._tables64)
        hx, he = 1.0 / a[step], -n[step] / a[step]
        delta = lam[step] - lam[step + 1]
        f = 1.0 - float(np.exp(delta))
        kx = n[step + 1] / n[step]
        if self._first_order(step):
            return (hx, he, kx, 0.0, f * a[step + 1], 0.0, 1.0)
        r = (lam[step] - lam[step - 1]) / (lam[step + 1] - lam[step])
        half = 0.5 * a[step + 1] * f / r
        return (hx, he, kx, 0.0, a[step + 1] * f + half, -half, 1.0)

    def __call__(self, x: Tensor, predicted_noise: Tensor, step: int, generator: Any = None) -> Tensor:
        assert self.first_inference_step <= step < self.num_inference_steps, f"invalid step {step}"
        x0 = (x - self.noise_std[step] * predicted_noise) / self.cumulative_scale_factors[step]
        self.estimated_data.append(x0)
        lam, a, n = self.signal_to_noise_ratios, self.cumulative_scale_factors, self.noise_std
        delta = lam[step] - lam[step + 1]
        if self._first_order(step):
            return (n[step + 1] / n[step]) * x + (1.0 - torch.exp(delta)) * a[step + 1] * x0
        cur, prev = self.estimated_data[-1], self.estimated_data[-2]
        est_delta = (cur - prev) / ((lam[step] - lam[step - 1]) / (lam[step + 1] - lam[step]))
        f = 1.0 - torch.exp(delta)
        return (n[step + 1] / n[step]) * x + a[step + 1] * f * cur + 0.5 * a[step + 1] * f * est_delta


class LCMSolver(_Schedule):
    """Latent Consistency Model solver (solvers/lcm.py:15-150): every step jumps to the consistency function's data estimate
    and, except on the last one, re-noises it to the next timestep of an emulated `num_orig_steps`-step DPM schedule
    (TRAILING spacing).  The update is linear in (x, eps, fresh noise):
        x' = a2 (c_skip + c_out / a) x  -  a2 c_out n / a  eps  +  n2 noise
    so it runs on the same fused guidance + update kernel as Euler / DPM++ with the kernel's history buffer pre-loaded with
    the step's noise draw (`needs_noise`); `__call__` is the unfused path with the reference's formulas and operation order."""

    def __init__(self, num_inference_steps: int, first_inference_step: int = 0, num_orig_steps: int = 50, device: Any = "cpu",
                 dtype: torch.dtype = torch.float32, num_train_timesteps: int = 1000, initial_diffusion_rate: float = 8.5e-4,
                 final_diffusion_rate: float = 1.2e-2) -> None:
        assert num_orig_steps >= num_inference_steps, f"num_orig_steps ({num_orig_steps}) < num_inference_steps ({num_inference_steps})"
        super().__init__(num_inference_steps, first_inference_step, num_train_timesteps, initial_diffusion_rate, final_diffusion_rate)
        self.dpm = DPMSolver(num_orig_steps, device=device, dtype=dtype, num_train_timesteps=num_train_timesteps, initial_diffusion_rate=initial_diffusion_rate,
                             final_diffusion_rate=final_diffusion_rate, timesteps_spacing="trailing")
        self.timestep_indices: list[int] = torch.floor(torch.linspace(start=0, end=num_orig_steps, steps=num_inference_steps + 1)[:-1]).int().tolist()
        self.timesteps = self.dpm.timesteps[self.timestep_indices]
        self._move(("scale_factors", "cumulative_scale_factors", "noise_std"), device, dtype)

    def scale_model_input(self, x: Tensor, step: int) -> Tensor:
        return x

    def input_scale(self, step: int) -> float:
        return 1.0

    def needs_noise(self, step: int) -> bool:
        return step != self.num_inference_steps - 1

    def _consistency(self, step: int) -> tuple[Tensor, Tensor, Tensor, Tensor]:
        t_now = self.timesteps[step]
        sigma = 0.5  # assumed standard deviation of the data distribution (lcm.py:137)
        t = t_now * 10
        return self.cumulative_scale_factors[t_now], self.noise_std[t_now], sigma ** 2 / (t ** 2 + sigma ** 2), t / torch.sqrt(sigma ** 2 + t ** 2)

    def __call__(self, x: Tensor, predicted_noise: Tensor, step: int, generator: Any = None) -> Tensor:
        assert self.first_inference_step <= step < self.num_inference_steps, f"invalid step {step}"
        a, n, c_skip, c_out = self._consistency(step)
        data = (x - n * predicted_noise) / a
        denoised = c_skip * x + c_out * data
        if not self.needs_noise(step):
            return denoised
        noise = torch.randn(predicted_noise.shape, generator=generator, device=self.device, dtype=self.dtype)
        return self.dpm.add_noise(x=denoised, noise=noise, step=int(self.timestep_indices[step + 1]))

    def linear_step(self, step: int) -> tuple[float, ...]:
        """(hx, he, kx, ke, kd, kp, s_next); the history slot holds this step's NOISE draw (kp = its coefficient)."""
        a, n, c_skip, c_out = (float(v) for v in self._consistency(step))
        a2, n2 = 1.0, 0.0
        if self.needs_noise(step):
            nxt = int(self.timestep_indices[step + 1])
            a2, n2 = float(self.dpm.cumulative_scale_factors[nxt]), float(self.dpm.noise_std[nxt])
        return (0.0, 0.0, a2 * (c_skip + c_out / a), -a2 * c_out * n / a, 0.0, n2, 1.0)
