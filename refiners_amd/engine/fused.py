"""Node-level drop-ins: fused MI355X replacements for `ResidualBlock` and `CrossAttentionBlock2d` as ordinary adapters.

This is the boundary in refiners' own terms: a replacement is a `fl.Chain` + `Adapter` whose
`forward` calls native code instead of iterating its children.  `inject()` splices it where the block was, `eject()`
restores the tree byte-for-byte (`repr` identical), the wrapped block stays a registered child (state-dict keys gain the
adapter's name like for any refiners adapter), and the node still takes its side inputs from the context store:
  FusedResidualBlock        reads "range_adapter".<key>            (reference latent_diffusion/unet.py:6-51, range_adapter.py:64-86)
  FusedCrossAttentionBlock2d reads "cross_attention_block".<key>, "ip_adapter".clip_image_embedding
                                                                    (reference cross_attention.py:92-175, image_prompt.py:244)
Input / output are NCHW like the blocks they replace (one layout kernel each way); everything in between is the same
lowering the whole-UNet engine uses.  LoRAs / IP-Adapter injected *inside* the wrapped block, before or after fusing,
are honoured (the program is re-lowered when the tree epoch moves).

`fuse(unet)` / `unfuse(unet)` wrap / unwrap every such block of a UNet.  The whole-UNet path (CompiledUNet) is faster
(no layout round trips, no Python between blocks); these adapters are for trees that contain nodes the UNet-level
lowering does not know.
"""
from __future__ import annotations

from typing import Any, Optional

import torch
from torch import Tensor

from .. import native
from ..fluxion import layers as fl
from ..fluxion.adapters import Adapter
from ..fluxion.tree import tree_epoch
from .lowering_blocks import BlockLowering as Lowering
from .packing import Act, PackCache, isa, kids
from .unet_lowering import UNetContext


class _FusedBlock(fl.Chain):
    """Shared machinery: per-(epoch, shape) program, static I/O buffers, replay."""

    def _init_runtime(self) -> None:
        self._rt: dict[str, Any] = {"key": None, "cache": PackCache()}

    def _lower(self, low: Lowering, ctx: UNetContext, a: Act) -> Act:
        raise NotImplementedError

    def _side_inputs(self) -> dict[str, Tensor]:
        raise NotImplementedError

    def _run(self, x: Tensor) -> Tensor:
        native.load()
        rt = self._rt
        side = self._side_inputs()
        key = (tree_epoch(), tuple(x.shape), x.dtype, x.device, tuple((k, tuple(v.shape)) for k, v in side.items()))
        if key != rt["key"]:
            B, C, H, W = x.shape
            low = Lowering(x.device, x.dtype, rt["cache"])
            ctx = UNetContext(low, B)
            x_in = torch.empty_like(x, memory_format=torch.contiguous_format)
            bufs: dict[str, Tensor] = {}
            with low.in_step():
                a = Act(low.pool.get(B * H * W, C), B, H, W)
                native.nchw_to_nhwc(x_in, a.tokens())
                for name, v in side.items():
                    if name.startswith("temb:"):
                        bufs[name] = torch.empty(B, v.shape[-1], device=x.device, dtype=x.dtype)
                        ts = low.pool.get(B, v.shape[-1])
                        low.pool.pin(ts)
                        native.silu(bufs[name], ts)
                        ctx.temb_silu[name[5:]] = ts
                    else:  # token inputs, padded to a multiple of 64 keys per sample
                        L, width = v.shape[1], v.shape[2]
                        bufs[name] = torch.zeros(B * ((L + 63) // 64 * 64), width, device=x.device, dtype=x.dtype)
                        c, k = name.split(":")[1:]
                        ctx.text[(c, k)] = (bufs[name], L)
                out_act = self._lower(low, ctx, a)
                out = torch.empty(B, out_act.C, out_act.H, out_act.W, device=x.device, dtype=x.dtype)
                native.nhwc_to_nchw(out_act.tokens(), out, out_act.C)
            rt["cache"].sweep()
            rt.update(key=key, low=low, x_in=x_in, bufs=bufs, out=out, side_id=None)
        rt["x_in"].copy_(x)
        B = x.shape[0]
        side_id = tuple((v.data_ptr(), v._version) for v in side.values())
        fresh = side_id != rt["side_id"]
        for name, v in side.items():
            buf = rt["bufs"][name]
            if name.startswith("temb:"):
                buf.copy_(v.reshape(-1, v.shape[-1]).expand(B, -1))
            elif fresh:
                buf.view(B, -1, v.shape[2])[:, : v.shape[1]].copy_(v)
        if fresh:
            native.replay(rt["low"].prologue)
            rt["side_id"] = side_id
        native.replay(rt["low"].step)
        rt["calls"] = rt.get("calls", 0) + 1
        if fresh or rt["calls"] % 16 == 0:  # a lost in-launch hand-over raises an error word on the device: look at it now and then (CompiledUNet.check_handovers)
            bad = rt["low"].handover_pending()
            if bad and bool((bad[0] if len(bad) == 1 else torch.stack(bad).any()).item()):
                rt["low"].handover_raise()
        return rt["out"].clone()


class FusedResidualBlock(_FusedBlock, Adapter[fl.Chain]):
    def __init__(self, target: fl.Chain) -> None:
        assert isa(target, "ResidualBlock"), f"FusedResidualBlock wraps a ResidualBlock, not {type(target).__name__}"
        with self.setup_adapter(target):
            super().__init__(target)
        self._init_runtime()

    def _time_key(self) -> Optional[str]:
        body = kids(kids(self.target)[0])
        ra = body[2]
        if isa(ra, "RangeAdapter2d"):
            return kids(kids(ra)[1])[0].key
        return None

    def _side_inputs(self) -> dict[str, Tensor]:
        key = self._time_key()
        if key is None:
            return {}
        temb = self.use_context("range_adapter").get(key)
        assert temb is not None, f"context range_adapter.{key} is unset"
        return {f"temb:{key}": temb}

    def _lower(self, low: Lowering, ctx: UNetContext, a: Act) -> Act:
        return low.residual_block(self.target, a, ctx)

    def forward(self, x: Tensor) -> Tensor:  # type: ignore[override]
        return self._run(x)


class FusedCrossAttentionBlock2d(_FusedBlock, Adapter[fl.Chain]):
    def __init__(self, target: fl.Chain) -> None:
        assert isa(target, "CrossAttentionBlock2d"), f"FusedCrossAttentionBlock2d wraps a CrossAttentionBlock2d, not {type(target).__name__}"
        with self.setup_adapter(target):
            super().__init__(target)
        self._init_runtime()

    def _side_inputs(self) -> dict[str, Tensor]:
        out: dict[str, Tensor] = {}
        seen = set()
        for m in self.target.modules():
            if isa(m, "UseContext") and m.context in ("cross_attention_block", "ip_adapter") and (m.context, m.key) not in seen:
                seen.add((m.context, m.key))
                v = self.use_context(m.context).get(m.key)
                assert v is not None, f"context {m.context}.{m.key} is unset"
                out[f"tok:{m.context}:{m.key}"] = v
        return out

    def _lower(self, low: Lowering, ctx: UNetContext, a: Act) -> Act:
        return low.cross_attention_2d(self.target, a, ctx)

    def forward(self, x: Tensor) -> Tensor:  # type: ignore[override]
        return self._run(x)


def fuse(root: fl.Chain) -> list[Any]:
    """Wrap every ResidualBlock / CrossAttentionBlock2d below `root` in its fused adapter; returns the adapters."""
    made: list[Any] = []
    targets = [(m, p) for m, p in root.walk(lambda m, p: isa(m, "ResidualBlock", "CrossAttentionBlock2d") and not isa(p, "_FusedBlock"))]
    for m, parent in targets:
        cls = FusedResidualBlock if isa(m, "ResidualBlock") else FusedCrossAttentionBlock2d
        made.append(cls(m).inject(parent))
    return made


def unfuse(root: fl.Chain) -> int:
    """Eject every fused adapter below `root` (the tree is restored exactly)."""
    n = 0
    for m in [m for m, _ in root.walk(lambda m, p: isa(m, "_FusedBlock"), recurse=True)]:
        m.eject()
        n += 1
    return n
