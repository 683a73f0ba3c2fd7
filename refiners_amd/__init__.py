"""refiners_amd: MI355X-native hot path for refiners' SDXL UNet step behind the fluxion Chain / Adapter API."""


def namespace():
    """The mirror's public classes under the names `synth.apply_adapters` expects (same as refiners' own)."""
    from types import SimpleNamespace

    from .fluxion import layers as fl
    from .fluxion.adapters import Conv2dLora, LinearLora, LoraAdapter
    from .latent_diffusion.adapters import ConditionEncoder, ControlLoraAdapter, SDXLIPAdapter, ZeroConvolution

    return SimpleNamespace(fl=fl, LinearLora=LinearLora, Conv2dLora=Conv2dLora, LoraAdapter=LoraAdapter, SDXLIPAdapter=SDXLIPAdapter,
                           ControlLoraAdapter=ControlLoraAdapter, ConditionEncoder=ConditionEncoder, ZeroConvolution=ZeroConvolution)
