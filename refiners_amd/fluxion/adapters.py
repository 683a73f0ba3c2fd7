"""Same names as `refiners.fluxion.adapters`."""
from .adapt import Adapter, Conv2dLora, LinearLora, Lora, LoraAdapter, auto_attach_loras, lookup_top_adapter  # noqa: F401
