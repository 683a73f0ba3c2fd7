"""The sampling loop around the UNet: DDIM solver and the classifier-free-guidance step.

* `DDIM`            reference latent_diffusion/solvers/ddim.py:14-95 + the schedule of solvers/solver.py:151-228, 386-416
* `SDXLDenoiser`    reference latent_diffusion/model.py:128-159 (CFG: cat(x, x) -> UNet -> u + s (c - u) -> solver) and
                    stable_diffusion_xl/model.py:113-141 (default_time_ids, set_unet_context)

Text encoders and the VAE are out of scope: embeddings and latents are tensors in and out.
This file is the unfused torch path; refiners_amd.engine.CompiledSDXL runs the same step as one HIP graph.
"""
from __future__ import annotations

from typing import Any

import torch
from torch import Tensor

import refiners_amd.fluxion.layers as fl


class DDIM:
    """Deterministic DDIM, noise prediction, quadratic beta schedule 8.5e-4..1.2e-2 over 1000 train steps,
    LEADING timestep spacing with offset 1 (50 steps -> 981, 961, ..., 1)."""

    def __init__(self, num_inference_steps: int, first_inference_step: int = 0, device: Any = "cpu", dtype: torch.dtype = torch.float32,
                 num_train_timesteps: int = 1000, initial_diffusion_rate: float = 8.5e-4, final_diffusion_rate: float = 1.2e-2) -> None:
        self.num_inference_steps = num_inference_steps
        self.first_inference_step = first_inference_step
        self.num_train_timesteps = num_train_timesteps
        betas = torch.linspace(initial_diffusion_rate ** 0.5, final_diffusion_rate ** 0.5, num_train_timesteps) ** 2
        self.scale_factors = 1 - betas
        self.cumulative_scale_factors = torch.sqrt(self.scale_factors.cumprod(dim=0))  # sqrt(alpha_bar_t)
        self.noise_std = torch.sqrt(1.0 - self.scale_factors.cumprod(dim=0))
        ratio = num_train_timesteps // num_inference_steps
        self.timesteps = (torch.arange(0, num_inference_steps, 1) * ratio + 1).flip(0)
        self.device, self.dtype = torch.device(device), dtype
        self.to(device, dtype)

    def to(self, device: Any = None, dtype: torch.dtype | None = None) -> "DDIM":
        for name in ("scale_factors", "cumulative_scale_factors", "noise_std"):
            setattr(self, name, getattr(self, name).to(device=device, dtype=dtype))
        self.timesteps = self.timesteps.to(device=device)
        if device is not None:
            self.device = torch.device(device)
        if dtype is not None:
            self.dtype = dtype
        return self

    @property
    def inference_steps(self) -> list[int]:
        return list(range(self.num_inference_steps))[self.first_inference_step:]

    def coefficients(self, step: int) -> tuple[float, float, float, float]:
        """(sqrt(a_t), sqrt(1 - a_t), sqrt(a_prev), noise factor) as Python floats: the host-side, sync-free form of
        the update below (used to fill the device coefficient table of the fused CFG+DDIM kernel)."""
        csf = self.cumulative_scale_factors.double().cpu()
        t = int(self.timesteps[step])
        prev_t = int(self.timesteps[step + 1]) if step < self.num_inference_steps - 1 else 0
        cur = float(csf[t])
        prev = float(csf[prev_t]) if prev_t > 0 else float(csf[0])
        noise = 0.0 if step == self.num_inference_steps - 1 else (1 - prev * prev) ** 0.5
        return cur, (1 - cur * cur) ** 0.5, prev, noise

    def scale_model_input(self, x: Tensor, step: int) -> Tensor:
        return x

    def sag_coefficients(self, step: int) -> tuple[float, float]:
        """(cumulative scale factor, noise std) that add_noise / remove_noise use at `step` (see refiners_amd.latent_diffusion.solvers)."""
        t = self.timesteps[step]
        return float(self.cumulative_scale_factors[t]), float(self.noise_std[t])

    def add_noise(self, x: Tensor, noise: Tensor, step: int) -> Tensor:
        t = self.timesteps[step]
        return self.cumulative_scale_factors[t] * x + self.noise_std[t] * noise

    def remove_noise(self, x: Tensor, noise: Tensor, step: int) -> Tensor:
        """The data estimate (x - noise_std * noise) / scale at this step's timestep (solvers/solver.py:298-319)."""
        t = self.timesteps[step]
        return (x - self.noise_std[t] * noise) / self.cumulative_scale_factors[t]

    def __call__(self, x: Tensor, predicted_noise: Tensor, step: int, generator: Any = None) -> Tensor:
        assert self.first_inference_step <= step < self.num_inference_steps, f"invalid step {step}"
        t = self.timesteps[step]
        prev_t = self.timesteps[step + 1] if step < self.num_inference_steps - 1 else torch.tensor([0], device=self.device)
        cur = self.cumulative_scale_factors[t]
        prev = self.cumulative_scale_factors[prev_t] if prev_t > 0 else self.cumulative_scale_factors[0]
        x0 = (x - torch.sqrt(1 - cur ** 2) * predicted_noise) / cur
        noise_factor = torch.sqrt(1 - prev ** 2) if step != self.num_inference_steps - 1 else 0
        return prev * x0 + noise_factor * predicted_noise


class SDXLDenoiser:
    """UNet + solver with classifier-free guidance: the part of refiners' StableDiffusion_XL the 50-step loop calls."""

    def __init__(self, unet: fl.Chain, solver: DDIM | None = None, classifier_free_guidance: bool = True) -> None:
        self.unet = unet
        self.solver = solver or DDIM(num_inference_steps=30, device=unet.device, dtype=unet.dtype)
        self.classifier_free_guidance = classifier_free_guidance

    @property
    def device(self) -> torch.device:
        return self.unet.device

    @property
    def dtype(self) -> torch.dtype:
        return self.unet.dtype

    @property
    def steps(self) -> list[int]:
        return self.solver.inference_steps

    def set_inference_steps(self, num_steps: int, first_step: int = 0) -> None:
        self.solver = DDIM(num_inference_steps=num_steps, first_inference_step=first_step, device=self.device, dtype=self.dtype)

    @property
    def default_time_ids(self) -> Tensor:
        ids = torch.tensor([1024, 1024, 0, 0, 1024, 1024], device=self.device)
        return ids.repeat(2 if self.classifier_free_guidance else 1, 1)

    def init_latents(self, size: tuple[int, int], batch: int = 1, generator: torch.Generator | None = None) -> Tensor:
        h, w = size
        x = torch.randn(batch, 4, h // 8, w // 8, generator=generator, dtype=torch.float32)
        return x.to(device=self.device, dtype=self.dtype)

    def set_unet_context(self, *, timestep: Tensor, clip_text_embedding: Tensor, pooled_text_embedding: Tensor, time_ids: Tensor, **_: Tensor) -> None:
        self.unet.set_timestep(timestep=timestep)
        self.unet.set_clip_text_embedding(clip_text_embedding=clip_text_embedding)
        self.unet.set_pooled_text_embedding(pooled_text_embedding=pooled_text_embedding)
        self.unet.set_time_ids(time_ids=time_ids)

    def __call__(self, x: Tensor, step: int, *, clip_text_embedding: Tensor, pooled_text_embedding: Tensor, time_ids: Tensor,
                 condition_scale: float = 5.0, **kwargs: Tensor) -> Tensor:
        if self.classifier_free_guidance:
            assert clip_text_embedding.shape[0] % 2 == 0, f"invalid batch size: {clip_text_embedding.shape[0]}"
        timestep = self.solver.timesteps[step].unsqueeze(dim=0)
        self.set_unet_context(timestep=timestep, clip_text_embedding=clip_text_embedding,
                              pooled_text_embedding=pooled_text_embedding, time_ids=time_ids, **kwargs)
        latents = torch.cat((x, x)) if self.classifier_free_guidance else x
        latents = self.solver.scale_model_input(latents, step=step)
        if self.classifier_free_guidance:
            uncond, cond = self.unet(latents).chunk(2)
            noise = uncond + condition_scale * (cond - uncond)
            sag = self._find_sag_adapter()
            if sag is not None:  # model.py:147-155
                noise = noise + self.compute_self_attention_guidance(sag, x.narrow(1, 0, 4), uncond, step, clip_text_embedding=clip_text_embedding,
                                                                     pooled_text_embedding=pooled_text_embedding, time_ids=time_ids)
        else:
            noise = self.unet(latents)
        return self.solver(x.narrow(1, 0, 4), predicted_noise=noise, step=step)

    # -- self-attention guidance (xl/model.py:164-250) -----------------------------------------------------------------------
    def _find_sag_adapter(self) -> Any:
        from .sag import SAGAdapter

        return next((p for p in self.unet.get_parents() if isinstance(p, SAGAdapter)), None)

    def set_self_attention_guidance(self, enable: bool, scale: float = 1.0) -> None:
        from .sag import SDXLSAGAdapter

        sag = self._find_sag_adapter()
        if enable:
            if sag is not None:
                sag.scale = scale
            else:
                SDXLSAGAdapter(target=self.unet, scale=scale).inject()
        elif sag is not None:
            sag.eject()

    def has_self_attention_guidance(self) -> bool:
        return self._find_sag_adapter() is not None

    def compute_self_attention_guidance(self, sag: Any, x: Tensor, noise: Tensor, step: int, *, clip_text_embedding: Tensor, pooled_text_embedding: Tensor,
                                        time_ids: Tensor) -> Tensor:
        degraded = sag.compute_degraded_latents(solver=self.solver, latents=x, noise=noise, step=step, classifier_free_guidance=True)
        neg_text, _ = clip_text_embedding.chunk(2)
        neg_pooled, _ = pooled_text_embedding.chunk(2)
        neg_ids, _ = time_ids.chunk(2)
        self.set_unet_context(timestep=self.solver.timesteps[step].unsqueeze(dim=0), clip_text_embedding=neg_text, pooled_text_embedding=neg_pooled, time_ids=neg_ids)
        if "ip_adapter" in self.unet.provider.contexts:
            ctx = self.unet.use_context("ip_adapter")
            keep = ctx["clip_image_embedding"].clone()
            ctx["clip_image_embedding"], _ = ctx["clip_image_embedding"].chunk(2)
            degraded_noise = self.unet(degraded)
            ctx["clip_image_embedding"] = keep
        else:
            degraded_noise = self.unet(degraded)
        return sag.scale * (noise - degraded_noise)
