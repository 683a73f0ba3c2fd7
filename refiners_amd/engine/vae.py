"""MI355X lowering of the latent-diffusion autoencoder's Decoder (latents -> image, the step
right after the sampling loop; reference auto_encoder.py:143-207, 322-325).

Same engine and kernels as the UNet (token-major activations, GroupNorm+SiLU, implicit-GEMM 3x3 with the shortcut fused
into the second conv's K loop, nearest-2x upsampling as address arithmetic):
  latents / encoder_scale -> Conv1x1(4->4)     one tiny NCHW kernel, 1/encoder_scale folded into its weights
  Conv3x3(4->512)                                im2col + GEMM (as the UNet stem)
  15 x Resnet, 3 x Upsample, GN+SiLU+Conv3x3(128->3)   native
  the mid-block SelfAttention2d (ONE head of 512 over H*W tokens)   q/k/v/out projections native; softmax(QK^T)V itself
        through torch SDPA: head_dim 512 is outside the flash kernel (listed in stats["fallback_nodes"])
"""
from __future__ import annotations

from typing import Any

import torch
from torch import Tensor

from .. import native
from .packing import Act, PackCache, _expect, cname, isa, kids, launches
from .unet_lowering import UNetContext, UNetLowering


class VAEDecoderLowering(UNetLowering):
    def lower_decoder(self, dec: Any, latents: Tensor, out: Tensor, encoder_scale: float) -> None:
        ch = kids(dec)
        _expect(len(ch) == 4 and isa(ch[0], "Conv2d") and isa(ch[1], "Conv2d") and isa(ch[2], "Chain") and isa(ch[3], "Chain"), "unexpected Decoder layout")
        B, C, H, W = latents.shape
        ctx = UNetContext(self, B)
        with self.in_step():
            c0 = ch[0]
            _expect(c0.kernel_size == (1, 1) and c0.in_channels == C and c0.in_channels <= 8 and c0.out_channels <= 8, "unexpected post-quantisation conv")
            w0 = self.cache.get(("vae_pq", encoder_scale) + PackCache.ident(c0.weight),
                                lambda: (c0.weight.detach().to(self.device, torch.float32).reshape(c0.out_channels, c0.in_channels) / encoder_scale).to(self.dtype).contiguous())
            y0 = torch.empty(B, c0.out_channels, H, W, device=self.device, dtype=self.dtype)
            self.keepalive = [y0]
            native.pointwise_nchw(latents, w0, self._w(c0.bias), y0)
            cur = self._stem_from(ch[1], y0)
            for stage in kids(ch[2]):
                for m in kids(stage):
                    if isa(m, "Resnet"):
                        nxt = self.resnet(m, cur)
                    elif isa(m, "Residual") and len(kids(m)) == 2 and isa(kids(m)[1], "SelfAttention2d"):
                        nxt = self.attention_2d(m, cur)
                    elif isa(m, "Upsample"):
                        nxt = self.piece(m, cur, ctx, H, W)
                        cur = None  # piece() released the input already
                    else:
                        nxt = self.torch_node(m, cur)
                    if cur is not None:
                        self.pool.put(cur.t)
                    cur = nxt
            gn, act, conv = kids(ch[3])
            _expect(isa(gn, "GroupNorm") and isa(act, "SiLU") and isa(conv, "Conv2d"), "unexpected Decoder output block")
            g = self.groupnorm(cur, gn, silu=True)
            self.pool.put(cur.t)
            y = self.conv(g, self.conv_spec(conv))
            self.pool.put(g.t)
            native.nhwc_to_nchw(y.tokens(), out, y.C)
            self.pool.put(y.t)

    def lower_encoder(self, enc: Any, image: Tensor, out: Tensor, encoder_scale: float) -> None:
        """Encoder (auto_encoder.py:83-141) + encoder_scale (317-320): (B, 3, 8h, 8w) in [-1, 1] -> latents (B, 4, h, w)."""
        ch = kids(enc)
        _expect(len(ch) == 4 and isa(ch[0], "Conv2d") and all(isa(c, "Chain") for c in ch[1:]), "unexpected Encoder layout")
        B, C, H, W = image.shape
        ctx = UNetContext(self, B)
        with self.in_step():
            cpad = (C + self.kblk - 1) // self.kblk * self.kblk  # 3 input channels -> one 128-byte block of zero-padded channels
            x0 = torch.zeros(B * H * W, cpad, device=self.device, dtype=self.dtype)
            self.keepalive = [x0]
            a0 = Act(x0, B, H, W)
            native.nchw_to_nhwc(image, a0.tokens())
            cur = self.conv(a0, self._padded_conv_spec(ch[0], cpad, ch[0].out_channels))
            for stage in kids(ch[1]):
                for m in kids(stage):
                    if isa(m, "Resnet"):
                        nxt = self.resnet(m, cur)
                    elif isa(m, "Residual") and len(kids(m)) == 2 and isa(kids(m)[1], "SelfAttention2d"):
                        nxt = self.attention_2d(m, cur)
                    elif isa(m, "Downsample"):
                        nxt = self.piece(m, cur, ctx, H, W)
                        cur = None
                    else:
                        nxt = self.torch_node(m, cur)
                    if cur is not None:
                        self.pool.put(cur.t)
                    cur = nxt
            gn, act, conv = kids(ch[2])
            _expect(isa(gn, "GroupNorm") and isa(act, "SiLU") and isa(conv, "Conv2d"), "unexpected Encoder output block")
            g = self.groupnorm(cur, gn, silu=True)
            self.pool.put(cur.t)
            y = self.conv(g, self.conv_spec(conv))
            self.pool.put(g.t)
            quant, cut = kids(ch[3])
            _expect(isa(quant, "Conv2d") and quant.kernel_size == (1, 1) and isa(cut, "Slicing") and cut.dim == 1 and cut.start == 0 and cut.step == 1, "unexpected quantisation tail")
            keep = cut.end or quant.out_channels
            _expect(out.shape[1] == keep and quant.in_channels <= 8, "unexpected latent width")
            moments = torch.empty(B, y.C, y.H, y.W, device=self.device, dtype=self.dtype)
            self.keepalive.append(moments)
            native.nhwc_to_nchw(y.tokens(), moments, y.C)
            self.pool.put(y.t)
            # Conv1x1(8 -> 8), Slicing(end=4) and the encoder_scale factor as one tiny kernel on the 4 kept output channels
            wq = self.cache.get(("vae_q_w", encoder_scale, keep) + PackCache.ident(quant.weight),
                                lambda: (quant.weight.detach().to(self.device, torch.float32).reshape(quant.out_channels, quant.in_channels)[:keep] * encoder_scale).to(self.dtype).contiguous())
            bq = self.cache.get(("vae_q_b", encoder_scale, keep) + PackCache.ident(quant.bias),
                                lambda: (quant.bias.detach().to(self.device, torch.float32)[:keep] * encoder_scale).to(self.dtype).contiguous())
            native.pointwise_nchw(moments, wq, bq, out)

    def _stem_from(self, conv: Any, x: Tensor) -> Act:
        io_x, self.io = getattr(self, "io", None), type("IO", (), {"x": x})()
        try:
            return self.stem(conv)
        finally:
            self.io = io_x

    def resnet(self, node: Any, a: Act) -> Act:
        """Sum(shortcut, Chain(GN, SiLU, conv, GN, SiLU, conv)) (auto_encoder.py:83-140): the ResidualBlock lowering without the
        time embedding; the 1x1 shortcut rides in the second conv's K loop."""
        ch = kids(node)
        _expect(len(ch) == 2 and isa(ch[1], "Chain"), "unexpected Resnet layout")
        body = kids(ch[1])
        _expect(len(body) == 6 and isa(body[0], "GroupNorm") and isa(body[1], "SiLU") and isa(body[3], "GroupNorm") and isa(body[4], "SiLU"), "unexpected Resnet body")
        c1, c2 = self.conv_spec(body[2]), self.conv_spec(body[5])
        g1 = self.groupnorm(a, body[0], silu=True)
        h1 = self.conv(g1, c1)
        self.pool.put(g1.t)
        g2 = self.groupnorm(h1, body[3], silu=True)
        self.pool.put(h1.t)
        if isa(ch[0], "Identity"):
            out = self.conv(g2, c2, res=a.t)
        else:
            sc = self.conv_spec(ch[0])
            _expect(sc.ksize == 1 and sc.lora is None and c2.lora is None, "unexpected Resnet shortcut")
            both = self.cache.get(("bias_sum",) + PackCache.ident(c2.b, sc.b), lambda: (c2.b.float() + sc.b.float()).to(self.dtype))
            out = self.conv(g2, c2, shortcut=(a, sc), bias=both)
        self.pool.put(g2.t)
        return out

    def attention_2d(self, node: Any, a: Act) -> Act:
        """x + SelfAttention2d(GroupNorm(x)) with a single head of `channels` (auto_encoder.py:221-225)."""
        gn, att = kids(node)
        ch = kids(att)
        _expect(isa(gn, "GroupNorm") and len(ch) == 6 and isa(ch[0], "Lambda") and isa(ch[1], "Parallel") and isa(ch[2], "Distribute") and isa(ch[3], "ScaledDotProductAttention")
                and isa(ch[5], "Lambda") and not ch[3].is_causal, "unexpected SelfAttention2d layout")
        heads = ch[3].num_heads
        g = self.groupnorm(a, gn, silu=False)
        q = self.linear(g.t, self.linear_spec(kids(ch[2])[0]))
        k = self.linear(g.t, self.linear_spec(kids(ch[2])[1]))
        v = self.linear(g.t, self.linear_spec(kids(ch[2])[2]))
        self.pool.put(g.t)
        _expect((a.C // heads) != 64, "head_dim 64 SelfAttention2d should use the flash path")
        o = self.sdpa(q, a.B, heads, [(k, v, a.HW, 1.0)], v_plain=[v])
        for t in (q, k, v):
            self.pool.put(t)
        out = self.linear(o, self.linear_spec(ch[4]), res=a.t)
        self.pool.put(o)
        return Act(out, a.B, a.H, a.W)


class CompiledVAEDecoder:
    """`image = CompiledVAEDecoder(vae)(latents)` == `vae.decode(latents)` (vae: LatentDiffusionAutoencoder-like Chain)."""

    def __init__(self, vae: Any) -> None:
        native.load()
        self.vae = vae
        self.cache = PackCache()
        self.key: Any = None
        self.stats: dict[str, Any] = {}

    @torch.no_grad()
    def __call__(self, latents: Tensor) -> Tensor:
        from ..fluxion.tree import tree_epoch

        dec = kids(self.vae)[1]
        dtype = dec.dtype
        key = (tree_epoch(), tuple(latents.shape), dtype, latents.device, float(self.vae.encoder_scale))
        if key != self.key:
            B, _, H, W = latents.shape
            self.x = torch.empty(tuple(latents.shape), device=latents.device, dtype=dtype)
            self.out = torch.empty(B, dec.output_channels, 8 * H, 8 * W, device=latents.device, dtype=dtype)
            low = VAEDecoderLowering(latents.device, dtype, self.cache)
            low.lower_decoder(dec, self.x, self.out, float(self.vae.encoder_scale))
            self.cache.sweep()
            self.low, self.key = low, key
            self.stats = dict(low.stats, step_ops=launches(low.step), pool_bytes=low.step_pool.bytes())
        self.x.copy_(latents)
        native.replay(self.low.step)
        return self.out.clone()


class CompiledVAEEncoder:
    """`latents = CompiledVAEEncoder(vae)(image)` == `vae.encode(image)` (image in [-1, 1], (B, 3, 8h, 8w))."""

    def __init__(self, vae: Any) -> None:
        native.load()
        self.vae = vae
        self.cache = PackCache()
        self.key: Any = None
        self.stats: dict[str, Any] = {}

    @torch.no_grad()
    def __call__(self, image: Tensor) -> Tensor:
        from ..fluxion.tree import tree_epoch

        enc = kids(self.vae)[0]
        dtype = enc.dtype
        key = (tree_epoch(), tuple(image.shape), dtype, image.device, float(self.vae.encoder_scale))
        if key != self.key:
            B, _, H, W = image.shape
            assert H % 8 == 0 and W % 8 == 0, "the autoencoder downsamples by 8"
            self.x = torch.empty(tuple(image.shape), device=image.device, dtype=dtype)
            self.out = torch.empty(B, 4, H // 8, W // 8, device=image.device, dtype=dtype)
            low = VAEDecoderLowering(image.device, dtype, self.cache)
            low.lower_encoder(enc, self.x, self.out, float(self.vae.encoder_scale))
            self.cache.sweep()
            self.low, self.key = low, key
            self.stats = dict(low.stats, step_ops=launches(low.step), pool_bytes=low.step_pool.bytes())
        self.x.copy_(image)
        native.replay(self.low.step)
        return self.out.clone()
