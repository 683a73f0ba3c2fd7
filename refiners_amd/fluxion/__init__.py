"""Host-side mirror of refiners' `fluxion` package (Chain tree, context store, leaf layers, adapters).

`import refiners_amd.fluxion.layers as fl` and `from refiners_amd.fluxion.adapters import Adapter, LoraAdapter` work
exactly like their `refiners.fluxion` counterparts; see tree.py / leaves.py / adapt.py for the reference citations.
"""
from . import adapters, layers  # noqa: F401
from .tree import ContextProvider, tree_epoch  # noqa: F401
