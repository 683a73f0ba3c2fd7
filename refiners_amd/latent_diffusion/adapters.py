"""Model-level adapters of the SDXL hot path: IP-Adapter (UNet side), ControlLora, and the LoRA manager.

* ImageProjection / ImageCrossAttention / CrossAttentionAdapter / IPAdapter / SDXLIPAdapter
      reference latent_diffusion/image_prompt.py:24-45, 237-564 and stable_diffusion_xl/image_prompt.py:9-65
* ConditionEncoder / ZeroConvolution / ControlLora / ControlLoraAdapter
      reference latent_diffusion/stable_diffusion_xl/control_lora.py:14-411
* SDLoraManager   reference latent_diffusion/lora.py:10-330

Out of scope here: the CLIP image encoder that produces the image embedding; the
adapters take the embedding as a tensor (`set_clip_image_embedding`), exactly as the sampling loop does.
"""
from __future__ import annotations

from typing import Any

import torch
from torch import Tensor, nn

import refiners_amd.fluxion.layers as fl
from refiners_amd.fluxion.adapters import Adapter, Lora, LoraAdapter, auto_attach_loras
from refiners_amd.fluxion.tree import Contexts, bump_epoch

from .perceiver import PerceiverResampler, convert_to_grid_features  # noqa: F401  (re-exported: the fine-grained image projection)
from .blocks import CrossAttentionBlock2d, RangeAdapter2d, ResidualAccumulator, ResidualBlock


# ------------------------------------------------------------------------------------------------ IP-Adapter
class ImageProjection(fl.Chain):
    """CLIP image embedding (B, 1024) -> `num_tokens` text-space tokens (B, num_tokens, text_dim)."""

    def __init__(self, clip_image_embedding_dim: int = 1024, clip_text_embedding_dim: int = 768, num_tokens: int = 4, device: Any = None, dtype: Any = None) -> None:
        self.clip_image_embedding_dim = clip_image_embedding_dim
        self.clip_text_embedding_dim = clip_text_embedding_dim
        self.num_tokens = num_tokens
        super().__init__(
            fl.Linear(clip_image_embedding_dim, clip_text_embedding_dim * num_tokens, device=device, dtype=dtype),
            fl.Reshape(num_tokens, clip_text_embedding_dim),
            fl.LayerNorm(clip_text_embedding_dim, device=device, dtype=dtype),
        )


class ImageCrossAttention(fl.Chain):
    """scale * SDPA(q, Wk' img, Wv' img): the image-token stream added to a text cross-attention's SDPA."""

    def __init__(self, text_cross_attention: fl.Attention, scale: float = 1.0) -> None:
        self._multiply = [fl.Multiply(scale)]
        att = text_cross_attention
        kw = dict(bias=att.use_bias, device=att.device, dtype=att.dtype)

        def image_tokens(width: int) -> fl.Chain:
            return fl.Chain(fl.UseContext(context="ip_adapter", key="clip_image_embedding"), fl.Linear(width, att.inner_dim, **kw))

        super().__init__(
            fl.Distribute(fl.Identity(), image_tokens(att.key_embedding_dim), image_tokens(att.value_embedding_dim)),
            fl.ScaledDotProductAttention(num_heads=att.num_heads, is_causal=att.is_causal),
            self.multiply,
        )

    @property
    def multiply(self) -> fl.Multiply:
        return self._multiply[0]

    @property
    def scale(self) -> float:
        return self.multiply.scale

    @scale.setter
    def scale(self, value: float) -> None:
        self.multiply.scale = value


class CrossAttentionAdapter(fl.Chain, Adapter[fl.Attention]):
    """Wraps a text cross-attention and turns its SDPA into Sum(SDPA, ImageCrossAttention)."""

    def __init__(self, target: fl.Attention, scale: float = 1.0) -> None:
        with self.setup_adapter(target):
            super().__init__(target)
        self._image_cross_attention = [ImageCrossAttention(text_cross_attention=target, scale=scale)]

    def inject(self, parent: fl.Chain | None = None) -> "CrossAttentionAdapter":
        sdpa = self.target.ensure_find(fl.ScaledDotProductAttention)
        self.target.replace(sdpa, fl.Sum(sdpa, self.image_cross_attention))
        return super().inject(parent)

    def eject(self) -> None:
        both = self.target.ensure_find_parent(self.image_cross_attention)
        both.remove(self.image_cross_attention)
        self.target.replace(both, both.layer("ScaledDotProductAttention", fl.ScaledDotProductAttention))
        super().eject()

    # -- the reference's accessor surface (image_prompt.py: CrossAttentionAdapter), one implementation behind it -----------------------------------
    def _image_branch(self) -> ImageCrossAttention:
        (branch,) = self._image_cross_attention  # (kept in a list so that it is not registered as a child: the target owns it once injected)
        return branch

    def _image_projection(self, slot: int) -> fl.Linear:
        """The Linear of Distribute slot `slot` inside the image branch: 1 = keys, 2 = values (slot 0 passes the query through)."""
        return self._image_branch().layer(("Distribute", slot, "Linear"), fl.Linear)

    image_cross_attention = property(_image_branch)
    image_key_projection = property(lambda self: self._image_projection(1))
    image_value_projection = property(lambda self: self._image_projection(2))
    scale = property(lambda self: self._image_branch().scale, lambda self, value: setattr(self._image_branch(), "scale", value))

    def load_weights(self, key_tensor: Tensor, value_tensor: Tensor) -> None:
        """Adopt the checkpoint's to_k_ip / to_v_ip matrices (moved to this adapter's device / dtype with the rest of the image branch)."""
        for slot, tensor in ((1, key_tensor), (2, value_tensor)):
            self._image_projection(slot).weight = nn.Parameter(tensor)
        self._image_branch().to(self.device, self.dtype)
        bump_epoch()  # packed copies of the old matrices are stale


class IPAdapter(fl.Chain, Adapter[fl.Chain]):
    """One CrossAttentionAdapter per text cross-attention of the UNet; image tokens come from context
    "ip_adapter".clip_image_embedding.  `clip_image_encoder` is kept only as an opaque handle."""

    def __init__(
        self,
        target: fl.Chain,
        clip_image_encoder: Any,
        image_proj: fl.Module,
        scale: float = 1.0,
        fine_grained: bool = False,
        weights: dict[str, Tensor] | None = None,
    ) -> None:
        with self.setup_adapter(target):
            super().__init__(target)
        self.fine_grained = fine_grained
        self._clip_image_encoder = [clip_image_encoder]
        if fine_grained and isinstance(clip_image_encoder, fl.Chain):
            self._grid_image_encoder = [convert_to_grid_features(clip_image_encoder)]
        self._image_proj = [image_proj]
        self.sub_adapters = [
            CrossAttentionAdapter(target=att, scale=scale)
            for att in target.layers(fl.Attention)
            if type(att) is not fl.SelfAttention
        ]
        if weights is not None:
            self.image_proj.load_state_dict({k.removeprefix("image_proj."): v for k, v in weights.items() if k.startswith("image_proj.")})
            for i, sub in enumerate(self.sub_adapters):
                pair = [v for k, v in weights.items() if k.startswith(f"ip_adapter.{i:03d}.")]
                assert len(pair) == 2
                sub.load_weights(*pair)

    @property
    def clip_image_encoder(self) -> Any:
        return self._clip_image_encoder[0]

    @property
    def image_proj(self) -> fl.Module:
        return self._image_proj[0]

    def inject(self, parent: fl.Chain | None = None) -> "IPAdapter":
        for sub in self.sub_adapters:
            sub.inject()
        return super().inject(parent)

    def eject(self) -> None:
        for sub in self.sub_adapters:
            sub.eject()
        super().eject()

    @property
    def scale(self) -> float:
        return self.sub_adapters[0].scale

    @scale.setter
    def scale(self, value: float) -> None:
        for sub in self.sub_adapters:
            sub.scale = value

    def set_clip_image_embedding(self, image_embedding: Tensor) -> None:
        self.set_context("ip_adapter", {"clip_image_embedding": image_embedding})

    @property
    def grid_image_encoder(self) -> fl.Chain:
        assert hasattr(self, "_grid_image_encoder"), "fine_grained needs a real CLIPImageEncoder to derive the grid encoder from"
        return self._grid_image_encoder[0]

    def compute_clip_image_embedding(self, clip_embedding: Tensor) -> Tensor:
        """[negative ; conditional] image tokens from an already-encoded CLIP embedding (reference image_prompt.py:497-525 for
        the non fine-grained path; `compute_image_tokens` below runs the encoder too)."""
        assert not self.fine_grained, "the fine-grained path encodes a ZERO IMAGE for the negative prompt: use compute_image_tokens(image)"
        cond = self.image_proj(clip_embedding)
        neg = self.image_proj(torch.zeros_like(clip_embedding))
        return torch.cat((neg, cond))

    def compute_image_tokens(self, image_prompt: Tensor) -> Tensor:
        """IPAdapter._compute_clip_image_embedding + the final cat (image_prompt.py:497-525) on a preprocessed (B, 3, 224, 224)
        tensor: plain adapters project the class embedding and an all-zero embedding; fine-grained ("plus") adapters resample
        the penultimate layer's token grid of the image and of an ALL-ZERO IMAGE."""
        if not self.fine_grained:
            return self.compute_clip_image_embedding(self.clip_image_encoder(image_prompt))
        enc = self.grid_image_encoder
        cond = self.image_proj(enc(image_prompt))
        neg = self.image_proj(enc(torch.zeros_like(image_prompt)))
        return torch.cat((neg, cond))


class SDXLIPAdapter(IPAdapter):
    def __init__(
        self,
        target: fl.Chain,
        clip_image_encoder: Any = None,
        image_proj: fl.Module | None = None,
        scale: float = 1.0,
        fine_grained: bool = False,
        weights: dict[str, Tensor] | None = None,
    ) -> None:
        if image_proj is None:
            xattn = target.ensure_find(CrossAttentionBlock2d)
            if not fine_grained:
                image_proj = ImageProjection(
                    clip_image_embedding_dim=getattr(clip_image_encoder, "output_dim", 1024),
                    clip_text_embedding_dim=xattn.context_embedding_dim, device=target.device, dtype=target.dtype,
                )
            else:  # xl/image_prompt.py:43-53
                image_proj = PerceiverResampler(
                    latents_dim=1280, num_attention_layers=4, num_attention_heads=20, head_dim=64, num_tokens=16,
                    input_dim=getattr(clip_image_encoder, "embedding_dim", 1280), output_dim=xattn.context_embedding_dim,
                    device=target.device, dtype=target.dtype,
                )
        super().__init__(target=target, clip_image_encoder=clip_image_encoder, image_proj=image_proj, scale=scale,
                         fine_grained=fine_grained, weights=weights)


# ------------------------------------------------------------------------------------------------ ControlLora
class ConditionEncoder(fl.Chain):
    """(B, 3, 8h, 8w) control image -> (B, 320, h, w): eight 3x3 convs with SiLU, three of them stride 2."""

    def __init__(self, in_channels: int = 3, out_channels: int = 320, intermediate_channels: tuple[int, ...] = (16, 32, 96, 256), device: Any = None, dtype: Any = None) -> None:
        kw = dict(device=device, dtype=dtype)
        c = intermediate_channels
        stages = [
            fl.Chain(
                fl.Conv2d(c[i], c[i], kernel_size=3, padding=1, **kw), fl.SiLU(),
                fl.Conv2d(c[i], c[i + 1], kernel_size=3, stride=2, padding=1, **kw), fl.SiLU(),
            )
            for i in range(len(c) - 1)
        ]
        super().__init__(
            fl.Chain(fl.Conv2d(in_channels, c[0], kernel_size=3, stride=1, padding=1, **kw), fl.SiLU()),
            *stages,
            fl.Conv2d(c[-1], out_channels, kernel_size=3, padding=1, **kw),
        )


class ZeroConvolution(fl.Passthrough):
    """residuals[i] += scale * conv1x1(x); passes x through."""

    def __init__(self, in_channels: int, out_channels: int, residual_index: int, scale: float = 1.0, device: Any = None, dtype: Any = None) -> None:
        self._scale = scale
        super().__init__(
            fl.Conv2d(in_channels, out_channels, kernel_size=1, device=device, dtype=dtype),
            fl.Multiply(scale=scale),
            ResidualAccumulator(n=residual_index),
        )

    @property
    def scale(self) -> float:
        return self._scale

    @scale.setter
    def scale(self, value: float) -> None:
        self._scale = value
        self.ensure_find(fl.Multiply).scale = value


class ControlLora(fl.Passthrough):
    """A weight-sharing copy of the UNet's encoder half that adds its block outputs into "unet".residuals before the
    UNet proper runs (it sits at index 0 of the UNet)."""

    def __init__(self, name: str, unet: fl.Chain, scale: float = 1.0, condition_channels: int = 3) -> None:
        self.name = name
        timestep_encoder = unet.layer("TimestepEncoder", fl.Chain).structural_copy()
        downblocks = unet.layer("DownBlocks", fl.Chain).structural_copy()
        middle_block = unet.layer("MiddleBlock", fl.Chain).structural_copy()
        super().__init__(timestep_encoder, downblocks, middle_block)
        key = f"timestep_embedding_control_lora_{name}"
        timestep_encoder.context_key = key
        for ra in self.layers(RangeAdapter2d):
            ra.context_key = key
        stem = downblocks.layer(0, fl.Chain)
        kw = dict(device=unet.device, dtype=unet.dtype)
        stem.append(
            fl.Residual(
                fl.UseContext(f"control_lora_{name}", "condition"),
                ConditionEncoder(in_channels=condition_channels, out_channels=stem.layer(0, fl.Conv2d).out_channels, **kw),
            )
        )
        for acc in list(self.layers(ResidualAccumulator)):
            stage = self.ensure_find_parent(acc)
            width = stage[0].out_channels
            assert isinstance(width, int), f"{stage[0]} has no out_channels attribute"
            stage.replace(acc, ZeroConvolution(in_channels=width, out_channels=width, residual_index=acc.n, scale=scale, **kw))
        mid = middle_block.layer(0, ResidualBlock).out_channels
        middle_block.append(ZeroConvolution(in_channels=mid, out_channels=mid, residual_index=len(downblocks), scale=scale, **kw))

    @property
    def scale(self) -> float:
        return self.ensure_find(ZeroConvolution).scale

    @scale.setter
    def scale(self, value: float) -> None:
        for zc in self.layers(ZeroConvolution):
            zc.scale = value


class ControlLoraAdapter(fl.Chain, Adapter[fl.Chain]):
    def __init__(self, name: str, target: fl.Chain, scale: float = 1.0, condition_channels: int = 3, weights: dict[str, Tensor] | None = None) -> None:
        with self.setup_adapter(target):
            self.name = name
            self._control_lora = [ControlLora(name=name, unet=target, scale=scale, condition_channels=condition_channels)]
            super().__init__(target)
        if weights:
            self.load_weights(weights)

    @property
    def control_lora(self) -> ControlLora:
        return self._control_lora[0]

    def init_context(self) -> Contexts:
        return {f"control_lora_{self.name}": {"condition": None}}

    def inject(self, parent: fl.Chain | None = None) -> "ControlLoraAdapter":
        self.target.insert(0, self.control_lora)
        return super().inject(parent)

    def eject(self) -> None:
        self.target.remove(self.control_lora)
        super().eject()

    def structural_copy(self) -> "ControlLoraAdapter":
        raise RuntimeError("ControlLoraAdapter cannot be copied, eject it first.")

    @property
    def scale(self) -> float:
        return self.control_lora.scale

    @scale.setter
    def scale(self, value: float) -> None:
        self.control_lora.scale = value

    def set_condition(self, condition: Tensor) -> None:
        self.set_context(f"control_lora_{self.name}", {"condition": condition})

    def load_weights(self, state_dict: dict[str, Tensor]) -> None:
        """Keys: `ControlLora.<path>.{down,up}` LoRA pairs, `ZeroConvolution_NN.*`, `ConditionEncoder.*`."""
        cl = self.control_lora
        pairs = {f"{k.removeprefix('ControlLora.')}.weight": v.to(dtype=cl.dtype, device=cl.device) for k, v in state_dict.items() if "ControlLora" in k}
        adapters = []
        for key, lora in Lora.from_dict(self.name, state_dict=pairs).items():
            leaf = cl.layer(key.split("."), fl.WeightedModule)
            assert lora.is_compatible(leaf)
            adapters.append(LoraAdapter(leaf, lora))
        for a in adapters:
            a.inject(cl)
        for i, zc in enumerate(cl.layers(ZeroConvolution)):
            tag = f"ZeroConvolution_{i + 1:02d}"
            zc.load_state_dict({k.removeprefix(f"{tag}."): v for k, v in state_dict.items() if tag in k})
        enc = cl.ensure_find(ConditionEncoder)
        enc.load_state_dict({k.removeprefix("ConditionEncoder."): v for k, v in state_dict.items() if "ConditionEncoder" in k})
        bump_epoch()


# ------------------------------------------------------------------------------------------------ LoRA manager
class SDLoraManager:
    """Attaches named LoRA files to a denoiser's UNet (text-encoder LoRAs are out of scope and ignored).

    `target` is anything with a `.unet` Chain and `.device` / `.dtype` (our sampling.SDXLDenoiser, or a bare UNet
    wrapped by `SDLoraManager.for_unet`)."""

    def __init__(self, target: Any) -> None:
        self.target = target

    @property
    def unet(self) -> fl.Chain:
        unet = self.target.unet
        assert isinstance(unet, fl.Chain)
        return unet

    def add_loras(
        self,
        name: str,
        /,
        tensors: dict[str, Tensor],
        scale: float = 1.0,
        unet_inclusions: list[str] | None = None,
        unet_exclusions: list[str] | None = None,
        unet_preprocess: dict[str, str] | None = None,
    ) -> None:
        assert name not in self.names, f"LoRA {name} already exists"
        loras = Lora.from_dict(name, state_dict={k: v.to(device=self.target.device, dtype=self.target.dtype) for k, v in tensors.items()})
        loras = {k: loras[k] for k in sorted(loras, key=SDLoraManager.sort_keys)}
        if all("unet" not in k and "text" not in k for k in loras):
            loras = {f"unet_{k}": v for k, v in loras.items()}
        self.add_loras_to_unet(loras, include=unet_inclusions, exclude=unet_exclusions, preprocess=unet_preprocess)
        self.set_scale(name, scale)

    def add_loras_to_unet(
        self,
        loras: dict[str, Lora],
        /,
        include: list[str] | None = None,
        exclude: list[str] | None = None,
        preprocess: dict[str, str] | None = None,
        debug_map: list[tuple[str, str]] | None = None,
    ) -> None:
        """Keys containing res / downsample / upsample are attached first, restricted to ResidualBlock / Downsample /
        Upsample ancestors; the rest goes everywhere else (reference lora.py:150-195)."""
        mine = {k: v for k, v in loras.items() if "unet" in k}
        exclude = ["TimestepEncoder"] if exclude is None else exclude
        preprocess = {"res": "ResidualBlock", "downsample": "Downsample", "upsample": "Upsample"} if preprocess is None else preprocess
        if include is not None:
            preprocess = {k: v for k, v in preprocess.items() if v in include}
        preprocess = {k: v for k, v in preprocess.items() if v not in exclude}
        special = {k: v for k, v in mine.items() if any(tag in k for tag in preprocess)}
        rest = {k: v for k, v in mine.items() if k not in special}
        for tag, cls_name in preprocess.items():
            group = {k: v for k, v in special.items() if tag in k}
            auto_attach_loras(group, self.unet, include=[cls_name], exclude=exclude, debug_map=debug_map)
        auto_attach_loras(rest, self.unet, exclude=[*exclude, *preprocess.values()], include=include, debug_map=debug_map)

    def remove_loras(self, *names: str) -> None:
        for adapter in self.lora_adapters:
            for name in names:
                adapter.remove_lora(name)
            if len(adapter.loras) == 0:
                adapter.eject()

    def remove_all(self) -> None:
        for adapter in self.lora_adapters:
            adapter.eject()

    def get_loras_by_name(self, name: str, /) -> list[Lora]:
        return [lora for lora in self.loras if lora.name == name]

    def get_scale(self, name: str, /) -> float:
        loras = self.get_loras_by_name(name)
        assert all(lora.scale == loras[0].scale for lora in loras), "lora scales are not all the same"
        return loras[0].scale

    def set_scale(self, name: str, scale: float, /) -> None:
        self.update_scales({name: scale})

    def update_scales(self, scales: dict[str, float], /) -> None:
        assert all(name in self.names for name in scales), f"Scales keys must be a subset of {self.names}"
        for name, scale in scales.items():
            for lora in self.get_loras_by_name(name):
                lora.scale = scale

    @property
    def loras(self) -> list[Lora]:
        return list(self.unet.layers(Lora))

    @property
    def names(self) -> list[str]:
        return list({lora.name for lora in self.loras})

    @property
    def lora_adapters(self) -> list[LoraAdapter]:
        return list(self.unet.layers(LoraAdapter))

    @property
    def scales(self) -> dict[str, float]:
        return {name: self.get_scale(name) for name in self.names}

    @staticmethod
    def _pad(text: str, /, padding_length: int = 2) -> str:
        return "_".join(part.zfill(padding_length) if part.isdigit() else part for part in text.split("_"))

    @staticmethod
    def sort_keys(key: str, /) -> tuple[str, int]:
        """q < k < v/in < out within a layer; numeric path components compared as zero-padded (lora.py:301-330)."""
        rank = {"q": 1, "k": 2, "v": 3, "in": 3, "out": 4, "out0": 4, "out_0": 4}
        table = {fmt.format(s): score for s, score in rank.items() for fmt in ("_{}", "_{}_lora")}
        suffix, score = next(((s, v) for s, v in table.items() if key.endswith(s)), ("", 5))
        return SDLoraManager._pad(key.removesuffix(suffix)), score
