"""The UNet walker: SDXLUNet / SD1UNet trees (with ControlLora / ControlNet / T2I-Adapter taps) -> `prologue` + `step` programs."""
from __future__ import annotations

import os

import math
from dataclasses import dataclass, field
from typing import Any, Callable, Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .. import native

from .lowering_blocks import BlockLowering
from .packing import Act, CatAct, ConvSpec, LinSpec, LoraPack, PackCache, Pool, Unsupported, _expect, cname, isa, kids, launches  # noqa: F401


@dataclass
class UNetContext:
    """Compile-time stand-in for the reference's context store during one UNet forward."""

    low: Lowering
    B: int
    text: dict[tuple[str, str], tuple[Tensor, int]] = field(default_factory=dict)  # padded token buffers + true length
    temb_silu: dict[str, Tensor] = field(default_factory=dict)  # context key -> SiLU(timestep embedding) [B, 1280]
    temb_table: dict[str, Tensor] = field(default_factory=dict)  # table mode: context key -> the [B*S, 1280] prologue table (rows b*S + s); the [B, 1280] rows are gathered on first use
    residuals: list[Any] = field(default_factory=list)
    shapes: list[tuple[int, int]] = field(default_factory=list)
    time_table: dict[tuple[str, int], Tensor] = field(default_factory=dict)  # (context key, id(packed weight)) -> [B, cout] view of the batched launch

    def tokens(self, context: str, key: str) -> tuple[Tensor, int]:
        got = self.text.get((context, key))
        if got is None:
            raise Unsupported(f"context {context}.{key} is not a registered token input")
        return got

    def time_bias(self, spec: ConvSpec) -> Tensor:
        key, lin = spec.time  # type: ignore[misc]
        got = self.time_table.get((key, id(lin.w)))
        if got is not None:  # a column slice of the one launch UNetLowering.batch_time_biases issued for every RangeAdapter2d of this key
            return got
        src = self.temb_silu.get(key)
        if src is None and key in self.temb_table:  # table mode, and a site outside the batched launch (its Linear carries LoRAs): this step's rows
            src = self.low.pool.get(self.B, self.temb_table[key].shape[1])
            self.low.pool.pin(src)
            native.gather_rows(self.temb_table[key], self.low.io.step_rows, src)
            self.temb_silu[key] = src
        if src is None:
            raise Unsupported(f"timestep embedding '{key}' has not been produced yet")
        out = self.low.pool.get(self.B, lin.N)
        self.low.pool.pin(out)
        self.low.linear(src, lin, out=out)
        return out


# ------------------------------------------------------------------------------------------------ whole-UNet lowering
def sinusoid_rows(x: Tensor, dim: int) -> Tensor:
    """range_adapter.py:11-22 on a 1-D float tensor: [cos | sin] of x * 10000^(-i/half), float32."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32, device=x.device) / half
    angle = x.float().unsqueeze(1) * torch.exp(exponent).unsqueeze(0)
    return torch.cat([torch.cos(angle), torch.sin(angle)], dim=-1)


@dataclass
class UNetIO:
    """Static input / output buffers of a lowered UNet (filled by CompiledUNet before a replay)."""

    x: Tensor  # [B, Cin, H, W] compute dtype, NCHW (the reference's input layout)
    timestep: Tensor  # [B] float32
    out: Tensor  # [B, Cout, H, W]
    pooled: Optional[Tensor] = None  # [B, 1280]
    time_ids: Optional[Tensor] = None  # [B, 6] float32
    # table mode (CompiledSDXL knows the solver's S timesteps): the whole timestep-embedding chain -- sinusoid, two Linears, SiLU, the 17
    # RangeAdapter2d projections -- is a function of (timestep, batch row) alone, so it runs ONCE per prompt in the prologue for all S
    # timesteps and a step only gathers its rows
    timesteps: Optional[Tensor] = None  # [B*S] float32, row b*S + s = timestep s
    step_rows: Optional[Tensor] = None  # [B] int32 on the device: b*S + (current step)
    tokens: dict[tuple[str, str], tuple[Tensor, int]] = field(default_factory=dict)  # (context, key) -> ([B*Lp, width], L)
    conditions: dict[str, Tensor] = field(default_factory=dict)  # control context name -> [B, 3, 8H, 8W]
    t2i: dict[str, list[Tensor]] = field(default_factory=dict)  # T2I-Adapter name -> its feature maps, NCHW, batch 1 or B


class UNetLowering(BlockLowering):
    """Lowers SDXLUNet / SD1UNet trees (reference xl/unet.py:258-351, sd1/unet.py:165-249), with ControlLoras at
    index 0 (xl/control_lora.py:144-248), into `prologue` + `step`."""

    def lower(self, unet: Any, io: UNetIO) -> None:
        B, _, H, W = io.x.shape
        self.io = io
        ctx = UNetContext(self, B)
        ctx.text = dict(io.tokens)
        n_slots = len(unet.init_context()["unet"]["residuals"])
        ctx.residuals = [None] * n_slots
        cur: Any = None
        with self.in_step():
            for child in kids(unet):
                if isa(child, "ControlLora"):
                    self.control_lora(child, ctx, H, W)
                elif isa(child, "Controlnet"):
                    self.controlnet(child, ctx, H, W)
                elif isa(child, "TimestepEncoder"):
                    self.timestep_encoder(child, ctx, scope=unet)
                elif cname(child) in ("DownBlocks", "UpBlocks"):
                    for stage in kids(child):
                        _expect(isa(stage, "Chain"), "UNet stages must be Chains")
                        for piece in kids(stage):
                            cur = self.piece(piece, cur, ctx, H, W)
                elif cname(child) == "MiddleBlock":
                    for piece in kids(child):
                        cur = self.piece(piece, cur, ctx, H, W)
                elif isa(child, "Residual") and len(kids(child)) == 1 and self._reads_residuals(kids(child)[0]):
                    cur = self.add_last_residual(cur, ctx)  # xl/unet.py:282
                elif isa(child, "Sum") and len(kids(child)) == 2 and self._reads_residuals(kids(child)[0]) and cname(kids(child)[1]) == "MiddleBlock":
                    for piece in kids(kids(child)[1]):  # sd1/unet.py:193-196: residuals[-1] + MiddleBlock(x)
                        cur = self.piece(piece, cur, ctx, H, W)
                    cur = self.add_last_residual(cur, ctx)
                elif isa(child, "Chain") and [cname(k) for k in kids(child)] == ["GroupNorm", "SiLU", "Conv2d"]:
                    cur = self.output_block(child, cur)
                else:
                    raise Unsupported(f"unexpected top-level UNet child {cname(child)}")
            _expect(isinstance(cur, Tensor), "UNet did not end with an output block")

    @staticmethod
    def _reads_residuals(m: Any) -> bool:
        return isa(m, "UseContext") and m.context == "unet" and m.key == "residuals"

    # -- timestep ----------------------------------------------------------------------------------------------
    def _range_encoder(self, enc: Any, res: Optional[Tensor], table: bool = False) -> Tensor:
        """RangeEncoder (range_adapter.py:25-44) of the step's timestep rows [B]; `table`: of all S timesteps x B rows (row b*S + s), `res`
        ([B, C], one row per batch row) then enters as a per-group row bias."""
        ch = kids(enc)
        _expect(len(ch) == 5 and isa(ch[0], "Lambda") and isa(ch[1], "Converter") and isa(ch[3], "SiLU"), "unexpected RangeEncoder layout")
        l1, l2 = self.linear_spec(ch[2]), self.linear_spec(ch[4])
        ts = self.io.timesteps if table else self.io.timestep
        R = ts.shape[0]
        sin = self.pool.get(R, enc.sinusoidal_embedding_dim)
        native.sinusoidal(ts, enc.sinusoidal_embedding_dim, sin)
        e1 = self.linear(sin, l1)
        self.pool.put(sin)
        e1s = self.pool.get(R, l1.N)
        native.silu(e1, e1s)
        if table and res is not None and l2.lora is None:
            te = self.pool.get(R, l2.N)
            native.gemm([(e1s, self.kblocked(l2.w))], te, bias=l2.b, rowbias=res, rows_per_group=R // res.shape[0])
        elif table and res is not None:
            # a LoRA on the encoder's second Linear (TimestepEncoder taken out of the loader's exclusions): the adapted launch has no row-bias slot, so the
            # per-batch-row TextTimeEmbedding enters as a residual expanded to the table's rows -- prologue work, once per prompt (round-4 advisor: this used to
            # raise Unsupported and send the whole UNet to the stock Chain forward)
            rep = R // res.shape[0]
            resx = self.pool.get(R, l2.N)
            self.python(lambda: resx.copy_(res.repeat_interleave(rep, dim=0)), "time-table row bias -> residual rows")
            te = self.linear(e1s, l2, res=resx)
            self.pool.put(resx)
        else:
            te = self.linear(e1s, l2, res=res)
        self.pool.put(e1)
        self.pool.put(e1s)
        return te

    def timestep_encoder(self, node: Any, ctx: UNetContext, scope: Any = None) -> None:
        """`scope`: the sub-tree whose RangeAdapter2d's read this encoder's context key (their projections are then batched)."""
        ch = kids(node)
        B = ctx.B
        if len(ch) == 2 and isa(ch[0], "Sum"):  # SDXL: Sum(Chain(UseContext timestep, RangeEncoder), TextTimeEmbedding)
            sc = kids(ch[0])
            _expect(len(sc) == 2 and isa(sc[1], "TextTimeEmbedding") and isa(kids(sc[0])[1], "RangeEncoder"), "unexpected SDXL TimestepEncoder layout")
            tt = kids(sc[1])
            _expect(len(tt) == 5 and isa(tt[0], "Concatenate") and isa(tt[1], "Converter") and isa(tt[3], "SiLU"), "unexpected TextTimeEmbedding layout")
            l1, l2 = self.linear_spec(tt[2]), self.linear_spec(tt[4])
            _expect(self.io.pooled is not None and self.io.time_ids is not None, "SDXL needs pooled_text_embedding and time_ids")
            with self.in_prologue():  # constant over the sampling loop
                pooled, ids, dim = self.io.pooled, self.io.time_ids, sc[1].time_ids_embedding_dim
                _expect(pooled.shape[1] + ids.shape[1] * dim == l1.K, "TextTimeEmbedding width mismatch")
                emb = self.pool.get(B, ids.shape[1] * dim)
                native.sinusoidal(ids, dim, emb, group=ids.shape[1])
                cat = self.pool.get(B, l1.K)
                native.concat2(pooled, emb, cat)
                t1 = self.linear(cat, l1)
                self.pool.put(emb)
                self.pool.put(cat)
                t1s = self.pool.get(B, l1.N)
                native.silu(t1, t1s)
                tte = self.pool.get(B, l2.N)
                self.pool.pin(tte)
                self.linear(t1s, l2, out=tte)
                self.pool.put(t1)
                self.pool.put(t1s)
            enc, res, writer = kids(sc[0])[1], tte, ch[1]
        else:  # SD1.5: Passthrough(UseContext timestep, RangeEncoder, SetContext)
            _expect(len(ch) == 3 and isa(ch[1], "RangeEncoder"), "unexpected TimestepEncoder layout")
            enc, res, writer = ch[1], None, ch[2]
        _expect(isa(writer, "SetContext") and writer.context == "range_adapter", "TimestepEncoder must write context range_adapter")
        if self.io.timesteps is not None and self.io.step_rows is not None and self.device.type != "meta" and os.environ.get("REFINERS_AMD_TIME_TABLE", "1") != "0":
            with self.in_prologue():
                temb = self._range_encoder(enc, res, table=True)
                tab = self.pool.get(temb.shape[0], temb.shape[1])
                self.pool.pin(tab)
                native.silu(temb, tab)
                self.pool.put(temb)
            ctx.temb_table[writer.key] = tab
            self.stats["time_table_rows"] = tab.shape[0]
            if scope is not None:
                self.batch_time_biases(scope, writer.key, tab, ctx, table=True)
            return
        temb = self._range_encoder(enc, res)
        ts = self.pool.get(B, temb.shape[1])
        self.pool.pin(ts)
        native.silu(temb, ts)
        self.pool.put(temb)
        ctx.temb_silu[writer.key] = ts
        if scope is not None:
            self.batch_time_biases(scope, writer.key, ts, ctx)

    def batch_time_biases(self, scope: Any, key: str, src: Tensor, ctx: UNetContext, table: bool = False) -> None:
        """Every RangeAdapter2d below `scope` computes Linear_i(SiLU(timestep embedding)) from the same [B, 1280] row pair
        (range_adapter.py:47-86): one GEMM against the row-concatenated weights instead of one 12-14 us, 2-row launch per ResidualBlock
        (19 per SDXL step); each block's conv then reads its [B, cout] column slice as `rowbias` (ld_rowbias = total width).
        Sites whose Linear carries run-time LoRAs keep their own launch."""
        if os.environ.get("REFINERS_AMD_TIME_BATCH", "1") == "0":
            return
        specs: list[LinSpec] = []

        def visit(m: Any) -> None:
            if isa(m, "RangeAdapter2d"):
                ch = kids(m)
                tc = kids(ch[1]) if len(ch) == 2 and isa(ch[1], "Chain") else []
                if len(tc) == 4 and isa(tc[0], "UseContext") and tc[0].context == "range_adapter" and tc[0].key == key and isa(tc[1], "SiLU"):
                    sp = self.linear_spec(tc[2])
                    if sp.lora is None and not sp.geglu and sp.K == src.shape[1] and (sp.N * self.es) % 16 == 0 and all(sp.w is not o.w for o in specs):
                        specs.append(sp)
                return
            for c in kids(m):
                visit(c)

        visit(scope)
        if len(specs) < 2:
            return
        ck = ("time_cat",) + PackCache.ident(*[sp.w for sp in specs], *[sp.b for sp in specs])

        def make() -> tuple[Tensor, Tensor]:
            w = torch.cat([sp.w for sp in specs], dim=0).contiguous()
            b = torch.cat([sp.b if sp.b is not None else torch.zeros(sp.N, device=sp.w.device, dtype=sp.w.dtype) for sp in specs]).contiguous()
            return w, b

        w, b = self.cache.get(ck, make)
        total = w.shape[0]
        out = self.pool.get(ctx.B, total)
        self.pool.pin(out)
        if table:  # `src` = the [B*S, 1280] table of the prologue: the projections of every (timestep, batch row) once per prompt, a step gathers its B rows
            with self.in_prologue():
                tb = self.pool.get(src.shape[0], total)
                self.pool.pin(tb)
                self.linear(src, LinSpec(w, b), out=tb)
            native.gather_rows(tb, self.io.step_rows, out)
        else:
            self.linear(src, LinSpec(w, b), out=out)
        off = 0
        for sp in specs:
            ctx.time_table[(key, id(sp.w))] = out[:, off:off + sp.N]
            off += sp.N
        self.stats["time_bias_batched"] = self.stats.get("time_bias_batched", 0) + len(specs)

    # -- stage pieces ------------------------------------------------------------------------------------------
    def stem(self, conv: Any) -> Act:
        """First convolution, straight from the NCHW latents (im2col of the tiny-channel image, then one GEMM)."""
        _expect(isa(conv, "Conv2d") and conv.kernel_size == (3, 3) and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (1, 1), "unsupported stem conv")
        B, cin, H, W = self.io.x.shape
        _expect(conv.in_channels == cin, "stem conv channel mismatch")
        kp = (9 * cin + self.kblk - 1) // self.kblk * self.kblk

        def pack() -> Tensor:
            wp = torch.zeros(conv.out_channels, kp, device=self.device, dtype=self.dtype)
            wp[:, : 9 * cin] = native.pack_conv_weight(self.cvt(conv.weight))
            return wp

        wp = self.cache.get(("stem", kp) + PackCache.ident(conv.weight), pack)
        cols = self.pool.get(B * H * W, kp)
        native.im2col3x3_nchw(self.io.x, cols)
        out = self.pool.get(B * H * W, conv.out_channels)
        cs = self.colstats_for(B * H * W, conv.out_channels, H * W)
        native.gemm([(cols, wp)], out, bias=self._w(conv.bias), colstats_out=cs)
        self.pool.put(cols)
        return Act(out, B, H, W, cs)

    def _release(self, a: Any) -> None:
        if a is not None and not isinstance(a, CatAct):  # (the ResidualBlock that consumes a CatAct releases its first part itself)
            self.pool.put(a.t)

    def piece(self, m: Any, cur: Optional[Act], ctx: UNetContext, H: int, W: int) -> Optional[Act]:
        if cur is None:
            return self.stem(m)
        if isinstance(cur, CatAct) and not isa(m, "ResidualBlock"):  # only a ResidualBlock reads a concatenation in place
            cat = cur
            cur = self.materialise(cat)
            self.pool.put(cat.a.t)
            self.stats["concat_materialised"] = self.stats.get("concat_materialised", 0) + 1
        if isa(m, "ResidualBlock"):
            out = self.residual_block(m, cur, ctx)
        elif isa(m, "CrossAttentionBlock2d"):
            out = self.cross_attention_2d(m, cur, ctx)
        elif isa(m, "Downsample"):
            ch = kids(m)
            conv = ch[-1]
            _expect(all(isa(c, "SetContext", "Lambda") for c in ch[:-1]), "unexpected Downsample layout")
            if any(isa(c, "SetContext") for c in ch[:-1]):
                ctx.shapes.append((cur.H, cur.W))
            explicit_pad = any(isa(c, "Lambda") for c in ch[:-1])  # padding=0 variant: F.pad(x, (0, 1, 0, 1)) + unpadded conv
            _expect(explicit_pad == (m.padding == 0), "Downsample padding attribute and layout disagree")
            spec = self.conv_spec(conv, asym=explicit_pad)
            _expect(spec.stride == 2 and spec.ksize == 3, "unexpected Downsample convolution")
            out = self.conv(cur, spec)
        elif isa(m, "Upsample"):
            ch = kids(m)
            _expect(len(ch) == 3 and isa(ch[0], "Parallel") and isa(ch[1], "Interpolate") and ch[1].mode == "nearest", "unexpected Upsample layout")
            src = kids(ch[0])[1]
            th, tw = ctx.shapes.pop() if isa(src, "UseContext") else (cur.H * m.upsample_factor, cur.W * m.upsample_factor)
            _expect((th, tw) == (2 * cur.H, 2 * cur.W), "only exact 2x nearest upsampling is lowered")
            out = self.conv(cur, self.conv_spec(ch[2]), ups=2)
        elif isa(m, "ResidualAccumulator"):
            self.accumulate(ctx, m.n, cur)
            return cur
        elif isa(m, "ResidualConcatenator"):
            skip = self.slot(ctx, m.n)
            _expect(skip is not None and (skip.B, skip.H, skip.W) == (cur.B, cur.H, cur.W), "skip tensor missing or of another size")
            return CatAct(cur, skip)  # (cur is released by the ResidualBlock that consumes the pair)
        elif isa(m, "ZeroConvolution"):
            self.zero_convolution(m, cur, ctx)
            return cur
        elif isa(m, "Residual") and len(kids(m)) == 2 and isa(kids(m)[0], "UseContext") and isa(kids(m)[1], "ConditionEncoder"):
            out = self.add_condition(m, cur)
        elif isa(m, "T2IFeatures"):
            out = self.add_t2i_features(m, cur)
        elif isa(m, "SelfAttentionShape"):  # SAG: remembers the (H, W) of the feature map the tapped attention runs on
            self.sag_shape = (cur.H, cur.W)
            return cur
        else:
            out = self.torch_node(m, cur)
        self._release(cur)
        return out

    def slot(self, ctx: UNetContext, n: int) -> Optional[Act]:
        return ctx.residuals[n]

    def accumulate(self, ctx: UNetContext, n: int, a: Act) -> None:
        """residuals[n] <- a + residuals[n]   (unet.py:54-66); aliasing `a` when the slot still holds its initial 0.0."""
        prev = ctx.residuals[n]
        if prev is None:
            self.pool.pin(a.t)
            ctx.residuals[n] = a
            return
        _expect((prev.B, prev.H, prev.W, prev.C) == (a.B, a.H, a.W, a.C), "residual slot shape mismatch")
        s = self.pool.get(a.M, a.C)
        self.pool.pin(s)
        native.axpby(a.t, 1.0, prev.t, 1.0, s)
        ctx.residuals[n] = Act(s, a.B, a.H, a.W)

    def add_last_residual(self, cur: Act, ctx: UNetContext) -> Act:
        last = ctx.residuals[-1]
        if last is None:  # 0.0 in the reference: x + 0.0
            return cur
        out = self.pool.get(cur.M, cur.C)
        native.axpby(cur.t, 1.0, last.t, 1.0, out)
        self._release(cur)
        return Act(out, cur.B, cur.H, cur.W)

    def output_block(self, node: Any, cur: Act) -> Tensor:
        gn, _, conv = kids(node)
        g = self.groupnorm(cur, gn, silu=True)
        self._release(cur)
        y = self.conv(g, self.conv_spec(conv))
        self.pool.put(g.t)
        native.nhwc_to_nchw(y.tokens(), self.io.out, y.C)
        self.pool.put(y.t)
        return self.io.out

    # -- ControlLora -------------------------------------------------------------------------------------------
    def zero_convolution(self, m: Any, cur: Act, ctx: UNetContext) -> None:
        """residuals[n] += scale * conv1x1(x)   (control_lora.py:90-141): the scale is folded into the packed weights."""
        ch = kids(m)
        _expect(len(ch) == 3 and isa(ch[0], "Conv2d") and isa(ch[1], "Multiply") and isa(ch[2], "ResidualAccumulator") and ch[1].bias == 0.0, "unexpected ZeroConvolution layout")
        conv, scale = ch[0], float(ch[1].scale)
        _expect(conv.kernel_size == (1, 1), "ZeroConvolution must be 1x1")
        w = self.cache.get(("zc_w", scale) + PackCache.ident(conv.weight), lambda: (conv.weight.detach().to(self.device, torch.float32).reshape(conv.out_channels, conv.in_channels) * scale).to(self.dtype).contiguous())
        b = self.cache.get(("zc_b", scale) + PackCache.ident(conv.bias), lambda: (conv.bias.detach().to(self.device, torch.float32) * scale).to(self.dtype).contiguous())
        prev = ctx.residuals[ch[2].n]
        z = self.pool.get(cur.M, conv.out_channels)
        self.pool.pin(z)
        native.gemm([(cur.t, w)], z, bias=b, res=None if prev is None else prev.t)
        ctx.residuals[ch[2].n] = Act(z, cur.B, cur.H, cur.W)

    def _padded_conv_spec(self, conv: Any, cin_pad: int, cout_pad: int) -> ConvSpec:
        """Conv2d whose channel counts are below the kernel's 128-byte granularity: zero-pad input channels (weight
        columns) and output channels (weight rows, bias) so that padded activations stay exactly zero."""
        _expect(isa(conv, "Conv2d") and conv.kernel_size == (3, 3) and tuple(conv.padding) == (1, 1) and conv.stride[0] == conv.stride[1], "unexpected ConditionEncoder conv")
        o, i = conv.out_channels, conv.in_channels

        def pack() -> tuple[Tensor, Tensor]:
            w = torch.zeros(cout_pad, cin_pad, 3, 3, device=self.device, dtype=self.dtype)
            w[:o, :i] = conv.weight.detach().to(device=self.device, dtype=self.dtype)
            b = torch.zeros(cout_pad, device=self.device, dtype=self.dtype)
            b[:o] = conv.bias.detach().to(device=self.device, dtype=self.dtype)
            return native.pack_conv_weight(w), b

        wp, bp = self.cache.get(("padconv", cin_pad, cout_pad) + PackCache.ident(conv.weight, conv.bias), pack)
        return ConvSpec(wp, bp, cin_pad, cout_pad, 3, conv.stride[0])

    def condition_encoder(self, enc: Any, cond: Tensor) -> Act:
        """ConditionEncoder (control_lora.py:14-87): (B, 3, 8H, 8W) -> (B, 320, H, W), eight 3x3 convs with SiLU.  Runs in
        the prologue (the control image is constant over the sampling loop) on channel-padded NHWC activations."""
        convs = [m for m in enc.modules() if isa(m, "Conv2d")]
        order = [m for m in enc.modules() if isa(m, "Conv2d", "SiLU")]
        _expect(len(convs) == 8 and isa(order[-1], "Conv2d"), "unexpected ConditionEncoder layout")
        pad = lambda c: (c + self.kblk - 1) // self.kblk * self.kblk
        B, C, H, W = cond.shape
        x = torch.zeros(B * H * W, pad(C), device=self.device, dtype=self.dtype)  # padding channels stay zero for ever
        self.prologue_keep = getattr(self, "prologue_keep", []) + [x]
        a = Act(x, B, H, W)
        native.nchw_to_nhwc(cond, a.tokens())
        for k, m in enumerate(order):
            if isa(m, "SiLU"):
                native.silu(a.t, a.t)
                continue
            last = m is order[-1]
            spec = self._padded_conv_spec(m, a.C, m.out_channels if last else pad(m.out_channels))
            nxt = self.conv(a, spec)
            if a.t is not x:
                self.pool.put(a.t)
            a = nxt
        return a

    def add_condition(self, m: Any, cur: Act) -> Act:
        """x + ConditionEncoder(condition)   (control_lora.py:190-202), encoder output produced in the prologue."""
        reader, enc = kids(m)
        # ControlLora: one context per adapter, key "condition"; SD1.5 Controlnet: shared context "controlnet", key "condition_<name>"
        cname_ = reader.context if reader.key == "condition" else f"{reader.context}.{reader.key}"
        cond = self.io.conditions.get(cname_)
        _expect(cond is not None, f"no condition image registered for {cname_}")
        with self.in_prologue():
            e = self.condition_encoder(enc, cond)
            if e.B == 1 and cur.B > 1:
                # ONE control picture for the whole batch: the reference's Sum broadcasts it (and it is the only form its Self-Attention
                # Guidance pass, n rows after a 2n-row CFG pass, can take): encode once, repeat the rows once per prompt, here
                _expect((e.H, e.W, e.C) == (cur.H, cur.W, cur.C), "ConditionEncoder output does not match the UNet stem")
                full = self.pool.get(cur.M, cur.C)
                hw = cur.H * cur.W
                for b in range(cur.B):
                    native.axpby(e.t, 1.0, e.t, 0.0, full[b * hw : (b + 1) * hw])
                self.pool.put(e.t)
                e = Act(full, cur.B, cur.H, cur.W)
            self.pool.pin(e.t)
        _expect((e.B, e.H, e.W, e.C) == (cur.B, cur.H, cur.W, cur.C), "ConditionEncoder output does not match the UNet stem")
        out = self.pool.get(cur.M, cur.C)
        native.axpby(cur.t, 1.0, e.t, 1.0, out)
        return Act(out, cur.B, cur.H, cur.W)

    def add_t2i_features(self, m: Any, cur: Act) -> Act:
        """x + scale * features[index]   (latent_diffusion/t2i_adapter.py:166-177): the feature map comes from the
        T2I-Adapter's condition encoder, once per image, as NCHW; it is turned token-major in the prologue (broadcast over
        the CFG batch when it has batch 1) and added with the node's live scale in one launch per step."""
        feats = self.io.t2i.get(m.name)
        _expect(feats is not None and 0 <= m.index < len(feats), f"no T2I-Adapter features registered for '{m.name}'")
        f = feats[m.index]
        fb, fc, fh, fw = f.shape
        _expect((fc, fh, fw) == (cur.C, cur.H, cur.W) and fb in (1, cur.B), f"T2I feature {m.index} of '{m.name}' is {tuple(f.shape)}, the UNet has {(cur.B, cur.C, cur.H, cur.W)} here")
        with self.in_prologue():
            tok = self.pool.get(cur.M, cur.C)
            self.pool.pin(tok)
            hw = fh * fw
            if fb == cur.B:
                native.nchw_to_nhwc(f, tok.view(cur.B, hw, cur.C))
            else:
                for b in range(cur.B):
                    native.nchw_to_nhwc(f, tok[b * hw : (b + 1) * hw].view(1, hw, cur.C))
        out = self.pool.get(cur.M, cur.C)
        native.axpby(cur.t, 1.0, tok, float(m.scale), out)
        self.stats["t2i_sites"] = self.stats.get("t2i_sites", 0) + 1
        return Act(out, cur.B, cur.H, cur.W)

    # -- SD1.5 ControlNet -----------------------------------------------------------------------------------------
    def controlnet(self, node: Any, ctx: UNetContext, H: int, W: int) -> None:
        """Controlnet = Passthrough(TimestepEncoder', Slicing(:4), DownBlocks', MiddleBlock') (stable_diffusion_1/controlnet.py:72-166):
        a second, separately weighted encoder half in front of the UNet; after every one of its 12 down blocks and after its
        middle block, residuals[n] += scale * scale_decay^(12 - n) * conv1x1_n(x) (:152-166).  Same shape of work as ControlLora:
        each tap is one GEMM (scale folded into the packed 1x1 weights, previous slot value as the residual operand)."""
        ch = kids(node)
        _expect(len(ch) == 4 and isa(ch[0], "TimestepEncoder") and isa(ch[1], "Slicing") and cname(ch[2]) == "DownBlocks" and cname(ch[3]) == "MiddleBlock",
                "unexpected Controlnet layout")
        _expect(ch[1].dim == 1 and ch[1].start == 0 and ch[1].end == 4 and self.io.x.shape[1] == 4, "Controlnet on a UNet input with more than 4 channels is not lowered")
        sub = UNetContext(self, ctx.B, text=ctx.text, temb_silu=ctx.temb_silu, residuals=ctx.residuals, shapes=[])
        self.timestep_encoder(ch[0], sub, scope=node)
        cur: Optional[Act] = None
        stages = [(n, kids(stage)) for n, stage in enumerate(kids(ch[2]))] + [(12, kids(ch[3]))]
        _expect(len(stages) == 13, "Controlnet must have 12 down blocks and a middle block")
        for n, pieces in stages:
            for piece in pieces:
                if isa(piece, "Passthrough") and len(kids(piece)) == 2 and isa(kids(piece)[0], "Conv2d") and isa(kids(piece)[1], "Lambda"):
                    self.controlnet_tap(node, kids(piece)[0], n, cur, sub)
                else:
                    cur = self.piece(piece, cur, sub, H, W)
        self._release(cur)
        self.stats["controlnets"] = self.stats.get("controlnets", 0) + 1

    def controlnet_tap(self, node: Any, conv: Any, n: int, cur: Optional[Act], ctx: UNetContext) -> None:
        _expect(cur is not None and conv.kernel_size == (1, 1) and conv.in_channels == cur.C, "unexpected Controlnet residual tap")
        scale = float(node.scale) * float(node.scale_decays[n])
        w = self.cache.get(("cn_w", scale) + PackCache.ident(conv.weight), lambda: (conv.weight.detach().to(self.device, torch.float32).reshape(conv.out_channels, conv.in_channels) * scale).to(self.dtype).contiguous())
        b = self.cache.get(("cn_b", scale) + PackCache.ident(conv.bias), lambda: (conv.bias.detach().to(self.device, torch.float32) * scale).to(self.dtype).contiguous())
        prev = ctx.residuals[n]
        z = self.pool.get(cur.M, conv.out_channels)
        self.pool.pin(z)
        native.gemm([(cur.t, w)], z, bias=b, res=None if prev is None else prev.t)
        ctx.residuals[n] = Act(z, cur.B, cur.H, cur.W)

    def control_lora(self, node: Any, ctx: UNetContext, H: int, W: int) -> None:
        """Passthrough(TimestepEncoder', DownBlocks', MiddleBlock'): fills ctx.residuals, returns nothing."""
        ch = kids(node)
        _expect(len(ch) == 3 and isa(ch[0], "TimestepEncoder") and cname(ch[1]) == "DownBlocks" and cname(ch[2]) == "MiddleBlock", "unexpected ControlLora layout")
        sub = UNetContext(self, ctx.B, text=ctx.text, temb_silu=ctx.temb_silu, residuals=ctx.residuals, shapes=[])
        self.timestep_encoder(ch[0], sub, scope=node)
        cur: Optional[Act] = None
        for stage in kids(ch[1]):
            for piece in kids(stage):
                cur = self.piece(piece, cur, sub, H, W)
        for piece in kids(ch[2]):
            cur = self.piece(piece, cur, sub, H, W)
        self._release(cur)
