"""Kernel-level parity cases: every C-ABI kernel against a plain PyTorch fp32 reference of the same op.

Each case is a function returning (max_abs_err, ref_scale, tol_rel) — the caller asserts err <= tol_rel * ref_scale.
The references are computed with stock torch ops in float32 ON THE SAME DEVICE from the same (dtype-rounded) inputs,
so the float32 cases check the 1e-3 contract of BASELINE.json and the bfloat16 cases check bf16-rounding-sized errors.
Used by tests/test_kernels_gpu.py (pytest -m gpu) and tools/probe.py (first-contact diagnostics on a GPU box).
"""
from __future__ import annotations

import ctypes
import math

import torch
import torch.nn.functional as F

from refiners_amd import native

DEV = "cuda"


def _tol(dtype: torch.dtype) -> float:
    # relative to the reference's max-abs; f32: the north-star 1e-3 bar (we are far inside it); bf16: output rounding
    return 1e-3 if dtype == torch.float32 else 1.6e-2


def _rand(*shape, dtype, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(device=DEV, dtype=dtype)


def _cmp(out: torch.Tensor, ref: torch.Tensor, dtype) -> tuple[float, float, float]:
    out = out.float()
    assert torch.isfinite(out).all(), "non-finite values in kernel output"
    err = (out - ref).abs().max().item()
    return err, max(ref.abs().max().item(), 1e-6), _tol(dtype)


# ------------------------------------------------------------------------------------------------ GEMM
def gemm_case(M, K, N, dtype, *, bias=False, res=False, rowbias=0, seed=0):
    x = _rand(M, K, dtype=dtype, seed=seed)
    w = _rand(N, K, dtype=dtype, seed=seed + 1, scale=K ** -0.5)
    b = _rand(N, dtype=dtype, seed=seed + 2) if bias else None
    r = _rand(M, N, dtype=dtype, seed=seed + 3) if res else None
    rb = _rand(M // rowbias, N, dtype=dtype, seed=seed + 4) if rowbias else None
    out = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    native.gemm([(x, w)], out, bias=b, res=r, rowbias=rb, rows_per_group=rowbias or 1)
    ref = x.float() @ w.float().t()
    if bias:
        ref = ref + b.float()
    if rowbias:
        ref = ref + rb.float().repeat_interleave(rowbias, dim=0)
    if res:
        ref = ref + r.float()
    return _cmp(out, ref, dtype)


def gemm_splitk_case(M, K, N, dtype, ksplit, tile=0, seed=5):
    """Split-K: ksplit workgroups per tile + deterministic float32 reduction with the full epilogue (bias, residual)."""
    x = _rand(M, K, dtype=dtype, seed=seed)
    w = _rand(N, K, dtype=dtype, seed=seed + 1, scale=K ** -0.5)
    b = _rand(N, dtype=dtype, seed=seed + 2)
    r = _rand(M, N, dtype=dtype, seed=seed + 3)
    ws = torch.empty(ksplit * M * N, dtype=torch.float32, device=DEV)
    out = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    native.gemm([(x, w)], out, bias=b, res=r, tile=tile, ksplit=ksplit, ws=ws)
    out2 = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    native.gemm([(x, w)], out2, bias=b, res=r, tile=tile, ksplit=ksplit, ws=ws)
    assert torch.equal(out, out2), "split-K must be bit-reproducible"
    ref = x.float() @ w.float().t() + b.float() + r.float()
    return _cmp(out, ref, dtype)


def conv_splitk_case(B, Cin, Cout, H, W, dtype, ksplit, seed=55):
    x = _rand(B, Cin, H, W, dtype=dtype, seed=seed)
    w = _rand(Cout, Cin, 3, 3, dtype=dtype, seed=seed + 1, scale=(Cin * 9) ** -0.5)
    b = _rand(Cout, dtype=dtype, seed=seed + 2)
    rb = _rand(B, Cout, dtype=dtype, seed=seed + 3)
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=1) + rb.float()[:, :, None, None]
    x_nhwc = x.permute(0, 2, 3, 1).contiguous()
    out = torch.empty(B * H * W, Cout, dtype=dtype, device=DEV)
    ws = torch.empty(ksplit * B * H * W * Cout, dtype=torch.float32, device=DEV)
    native.conv_gemm([(x_nhwc, native.pack_conv_weight(w), 3, 1, 1)], out, B, H, W, bias=b, rowbias=rb, rows_per_group=H * W, ksplit=ksplit, ws=ws)
    got = out.float().reshape(B, H, W, Cout).permute(0, 3, 1, 2)
    return _cmp(got, ref, dtype)


def gemm_lora_case(M, K, N, dtype, ranks=(16, 16), seed=10):
    """y = x W^T + b + sum_i s_i (x A_i^T) B_i^T as ONE fused launch after the skinny down-projection launch."""
    x = _rand(M, K, dtype=dtype, seed=seed)
    w = _rand(N, K, dtype=dtype, seed=seed + 1, scale=K ** -0.5)
    b = _rand(N, dtype=dtype, seed=seed + 2)
    scales = [1.0, 0.8, 0.5][: len(ranks)]
    downs = [_rand(r, K, dtype=dtype, seed=seed + 10 + i, scale=1.0 / r) for i, r in enumerate(ranks)]
    ups = [_rand(N, r, dtype=dtype, seed=seed + 20 + i, scale=0.05) for i, r in enumerate(ranks)]
    rt = sum(ranks)
    es = 4 if dtype == torch.float32 else 2
    bk = 128 // es
    rpad = (rt + bk - 1) // bk * bk
    a_cat = torch.zeros(rpad, K, dtype=dtype, device=DEV)
    bs_cat = torch.zeros(N, rpad, dtype=dtype, device=DEV)
    o = 0
    for d, u, s in zip(downs, ups, scales):
        a_cat[o : o + d.shape[0]] = d
        bs_cat[:, o : o + d.shape[0]] = (u.float() * s).to(dtype)
        o += d.shape[0]
    t = torch.empty(M, rpad, dtype=dtype, device=DEV)
    native.gemm([(x, a_cat)], t)
    out = torch.empty(M, N, dtype=dtype, device=DEV)
    native.gemm([(x, w), (t, bs_cat)], out, bias=b)
    ref = x.float() @ w.float().t() + b.float()
    for d, u, s in zip(downs, ups, scales):
        ref = ref + s * ((x.float() @ d.float().t()) @ u.float().t())
    return _cmp(out, ref, dtype)


def gemm_geglu_case(M, K, n_out, dtype, seed=30, tile=0):
    x = _rand(M, K, dtype=dtype, seed=seed)
    w = _rand(2 * n_out, K, dtype=dtype, seed=seed + 1, scale=K ** -0.5)
    b = _rand(2 * n_out, dtype=dtype, seed=seed + 2)
    idx = native.geglu_pack_index(n_out, device=DEV)
    wp, bp = w[idx].contiguous(), b[idx].contiguous()
    out = torch.empty(M, n_out, dtype=dtype, device=DEV)
    native.gemm([(x, wp)], out, bias=bp, geglu=True, tile=tile)
    y = x.float() @ w.float().t() + b.float()
    a, g = y.chunk(2, dim=-1)
    ref = a * F.gelu(g, approximate="none")
    return _cmp(out, ref, dtype)


def gemm_geglu_kblocked_chain_case(M, K, n_out, N2, dtype, seed=31):
    """FeedForward as the engine runs it: GEGLU epilogue storing its [M, n_out] result K-blocked, consumed as the K-blocked x
    operand of the second GEMM (K-blocked weights too) -- bit-identical to the row-major chain."""
    x = _rand(M, K, dtype=dtype, seed=seed)
    w = _rand(2 * n_out, K, dtype=dtype, seed=seed + 1, scale=K ** -0.5)
    b = _rand(2 * n_out, dtype=dtype, seed=seed + 2)
    idx = native.geglu_pack_index(n_out, device=DEV)
    wp, bp = w[idx].contiguous(), b[idx].contiguous()
    w2 = _rand(N2, n_out, dtype=dtype, seed=seed + 3, scale=n_out ** -0.5)
    r = _rand(M, N2, dtype=dtype, seed=seed + 4)
    mid0 = torch.empty(M, n_out, dtype=dtype, device=DEV)
    out0 = torch.empty(M, N2, dtype=dtype, device=DEV)
    native.gemm([(x, wp)], mid0, bias=bp, geglu=True)
    native.gemm([(mid0, w2)], out0, res=r)
    mid1 = torch.full((M, n_out), float("nan"), dtype=dtype, device=DEV)
    out1 = torch.full((M, N2), float("nan"), dtype=dtype, device=DEV)
    native.gemm([(x, native.KBlocked(wp))], mid1, bias=bp, geglu=True, out_kblocked=True)
    blocked = native.KBlocked.adopt(mid1.view(-1), M, n_out)
    assert torch.equal(blocked.dense(), mid0), "K-blocked GEGLU store differs from the row-major one"
    native.gemm([(blocked, native.KBlocked(w2))], out1, res=r)
    assert torch.equal(out0, out1), "K-blocked chain changed the result"
    y = x.float() @ w.float().t() + b.float()
    a, g = y.chunk(2, dim=-1)
    ref = (a * F.gelu(g, approximate="none")).to(dtype).float() @ w2.float().t() + r.float()
    return _cmp(out1, ref, dtype)


def gemm_vt_case(L, K, Cc, dtype, B=2, seed=40):
    """V^T projection: out[Cc, B*L] = W_v @ x^T (operands swapped) -- the layout mi355x_attention consumes."""
    x = _rand(B * L, K, dtype=dtype, seed=seed)
    w = _rand(Cc, K, dtype=dtype, seed=seed + 1, scale=K ** -0.5)
    out = torch.empty(Cc, B * L, dtype=dtype, device=DEV)
    native.gemm([(w, x)], out)
    ref = w.float() @ x.float().t()
    return _cmp(out, ref, dtype)


# ------------------------------------------------------------------------------------------------ conv
def conv_case(B, Cin, Cout, H, W, dtype, *, ksize=3, stride=1, ups=1, split=0, bias=True, rowbias=False, res=False, seed=50, tile=0):
    x = _rand(B, Cin, H, W, dtype=dtype, seed=seed)
    w = _rand(Cout, Cin, ksize, ksize, dtype=dtype, seed=seed + 1, scale=(Cin * ksize * ksize) ** -0.5)
    b = _rand(Cout, dtype=dtype, seed=seed + 2) if bias else None
    xin = F.interpolate(x.float(), scale_factor=ups, mode="nearest") if ups > 1 else x.float()
    ref = F.conv2d(xin, w.float(), b.float() if bias else None, stride=stride, padding=ksize // 2)
    OH, OW = ref.shape[2], ref.shape[3]
    x_nhwc = x.permute(0, 2, 3, 1).contiguous()
    if split:
        xa, xb = x_nhwc[..., :split].contiguous(), x_nhwc[..., split:].contiguous()
        segs = [
            (xa, native.pack_conv_weight(w[:, :split]), ksize, stride, ups),
            (xb, native.pack_conv_weight(w[:, split:]), ksize, stride, ups),
        ]
    else:
        segs = [(x_nhwc, native.pack_conv_weight(w), ksize, stride, ups)]
    rb = r = None
    if rowbias:
        rb = _rand(B, Cout, dtype=dtype, seed=seed + 3)
        ref = ref + rb.float()[:, :, None, None]
    if res:
        r = _rand(B * OH * OW, Cout, dtype=dtype, seed=seed + 4)
        ref = ref + r.float().reshape(B, OH, OW, Cout).permute(0, 3, 1, 2)
    out = torch.empty(B * OH * OW, Cout, dtype=dtype, device=DEV)
    native.conv_gemm(segs, out, B, OH, OW, bias=b, rowbias=rb, rows_per_group=OH * OW, res=r, tile=tile)
    got = out.float().reshape(B, OH, OW, Cout).permute(0, 3, 1, 2)
    return _cmp(got, ref, dtype)


def conv_asym_case(B, Cin, Cout, H, W, dtype, seed=58):
    """fl.Downsample(padding=0): F.pad(x, (0, 1, 0, 1)) + 3x3 stride-2 conv without padding."""
    x = _rand(B, Cin, H, W, dtype=dtype, seed=seed)
    w = _rand(Cout, Cin, 3, 3, dtype=dtype, seed=seed + 1, scale=(Cin * 9) ** -0.5)
    b = _rand(Cout, dtype=dtype, seed=seed + 2)
    ref = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w.float(), b.float(), stride=2)
    OH, OW = ref.shape[2], ref.shape[3]
    out = torch.empty(B * OH * OW, Cout, dtype=dtype, device=DEV)
    native.conv_gemm([(x.permute(0, 2, 3, 1).contiguous(), native.pack_conv_weight(w), 3, 2, 1, 1)], out, B, OH, OW, bias=b)
    return _cmp(out.float().reshape(B, OH, OW, Cout).permute(0, 3, 1, 2), ref, dtype)


def conv_first_case(B, H, W, dtype, seed=60):
    """4 -> 320 input conv through im2col (K = 36 padded to one 128-byte block)."""
    Cin, Cout = 4, 320
    x = _rand(B, Cin, H, W, dtype=dtype, seed=seed)
    w = _rand(Cout, Cin, 3, 3, dtype=dtype, seed=seed + 1, scale=1 / 6)
    b = _rand(Cout, dtype=dtype, seed=seed + 2)
    es = 4 if dtype == torch.float32 else 2
    kp = 128 // es * ((36 * es + 127) // 128)
    cols = torch.empty(B * H * W, kp, dtype=dtype, device=DEV)
    native.im2col3x3_nchw(x, cols)
    wp = torch.zeros(Cout, kp, dtype=dtype, device=DEV)
    wp[:, :36] = native.pack_conv_weight(w)
    out = torch.empty(B * H * W, Cout, dtype=dtype, device=DEV)
    native.gemm([(cols, wp)], out, bias=b)
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=1)
    got = out.float().reshape(B, H, W, Cout).permute(0, 3, 1, 2)
    return _cmp(got, ref, dtype)


# ------------------------------------------------------------------------------------------------ q-projection + cross-attention in one launch
def _vt_from_v(v: torch.Tensor, Lkp: int) -> torch.Tensor:
    B, Lk, Cc = v.shape
    vt = torch.zeros(Cc, B, Lkp, dtype=v.dtype, device=v.device)
    vt[:, :, :Lk] = v.permute(2, 0, 1)
    return vt


def _sdpa_ref(q, k, v, H):
    B, Lq, Cc = q.shape
    D = Cc // H
    qh = q.float().reshape(B, Lq, H, D).transpose(1, 2)
    kh = k.float().reshape(B, -1, H, D).transpose(1, 2)
    vh = v.float().reshape(B, -1, H, D).transpose(1, 2)
    att = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(D), dim=-1)
    return (att @ vh).transpose(1, 2).reshape(B, Lq, Cc)


def attention_case(B, H, Lq, Lk, dtype, *, ip_tokens=0, ip_scale=0.7, spike=False, seed=70, pipe=None):
    """pipe = mi355x_attention_set_pipeline code (K/V tiles in flight | OPT bits << 4) of a non-default kernel variant to run."""
    D = 64
    Cc = H * D
    q = _rand(B, Lq, Cc, dtype=dtype, seed=seed)
    k = _rand(B, Lk, Cc, dtype=dtype, seed=seed + 1)
    v = _rand(B, Lk, Cc, dtype=dtype, seed=seed + 2)
    if spike:  # force large running-max jumps late in the key sequence (online-softmax rescale path)
        k[:, Lk - 3] *= 6.0
        k[:, Lk // 2] *= 4.0
    Lkp = (Lk + 63) // 64 * 64
    streams = [(k, _vt_from_v(v, Lkp), Lk, 1.0)]
    ref = _sdpa_ref(q, k, v, H)
    if ip_tokens:
        k2 = _rand(B, ip_tokens, Cc, dtype=dtype, seed=seed + 3)
        v2 = _rand(B, ip_tokens, Cc, dtype=dtype, seed=seed + 4)
        streams.append((k2, _vt_from_v(v2, (ip_tokens + 63) // 64 * 64), ip_tokens, ip_scale))
        ref = ref + ip_scale * _sdpa_ref(q, k2, v2, H)
    out = torch.full((B, Lq, Cc), float("nan"), dtype=dtype, device=DEV)
    if pipe is not None:
        native.load().mi355x_attention_set_pipeline(pipe, -1)
    try:
        native.attention(q, out, H, streams)
    finally:
        if pipe is not None:
            native.attention_pipeline_from_env()
    return _cmp(out, ref, dtype)


def attention_general_case(B, H, Lq, Lk, Dqk, Dv, dtype, *, causal=False, spike=False, seed=170, fast=None):
    """softmax(Q K^T / sqrt(Dqk) [causal]) V with Dqk != Dv allowed, against torch in fp32.  fast = 0: the bf16 launch takes the round-5 instance
    (per-tile maximum, vector row sums) instead of the lazy one."""
    q = _rand(B, Lq, H * Dqk, dtype=dtype, seed=seed)
    k = _rand(B, Lk, H * Dqk, dtype=dtype, seed=seed + 1)
    v = _rand(B, Lk, H * Dv, dtype=dtype, seed=seed + 2)
    if spike:
        k[:, Lk - 3] *= 6.0
        k[:, Lk // 2] *= 4.0
    Lkp = (Lk + 63) // 64 * 64
    qh = q.float().reshape(B, Lq, H, Dqk).transpose(1, 2)
    kh = k.float().reshape(B, Lk, H, Dqk).transpose(1, 2)
    vh = v.float().reshape(B, Lk, H, Dv).transpose(1, 2)
    logits = qh @ kh.transpose(-1, -2) / math.sqrt(Dqk)
    if causal:
        keep = torch.ones(Lq, Lk, dtype=torch.bool, device=q.device).tril()
        logits = logits.masked_fill(~keep, float("-inf"))
    ref = (torch.softmax(logits, dim=-1) @ vh).transpose(1, 2).reshape(B, Lq, H * Dv)
    out = torch.full((B, Lq, H * Dv), float("nan"), dtype=dtype, device=DEV)
    if fast is not None:
        native.load().mi355x_attention_general_set_fast(int(fast))
    try:
        native.attention_general(q, k, _vt_from_v(v, Lkp), out, H, Lk, causal=causal)
    finally:
        if fast is not None:
            native.load().mi355x_attention_general_set_fast(1)
    return _cmp(out, ref, dtype)


def attention_relpos_case(B, H, gh, gw, D, dtype, seed=180):
    """SegmentAnything-style decomposed relative position bias folded into extra QK columns:
    logits[q, (kh, kw)] = scale * q.k + rel_h[q, kh] + rel_w[q, kw]   (image_encoder.py:82-127)."""
    L = gh * gw
    scale = D ** -0.5
    q = _rand(B, L, H * D, dtype=dtype, seed=seed)
    k = _rand(B, L, H * D, dtype=dtype, seed=seed + 1)
    v = _rand(B, L, H * D, dtype=dtype, seed=seed + 2)
    rel_h = _rand(B, H, L, gh, dtype=dtype, seed=seed + 3)
    rel_w = _rand(B, H, L, gw, dtype=dtype, seed=seed + 4)
    qh = q.float().reshape(B, L, H, D).transpose(1, 2)
    kh = k.float().reshape(B, L, H, D).transpose(1, 2)
    vh = v.float().reshape(B, L, H, D).transpose(1, 2)
    bias = (rel_h.float()[..., :, None] + rel_w.float()[..., None, :]).reshape(B, H, L, L)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * scale + bias, dim=-1) @ vh).transpose(1, 2).reshape(B, L, H * D)
    Dq = (D + gh + gw + 7) // 8 * 8
    qa = torch.zeros(B, L, H, Dq, dtype=dtype, device=DEV)
    ka = torch.zeros(B, L, H, Dq, dtype=dtype, device=DEV)
    qa[..., :D] = (q.reshape(B, L, H, D).float() * scale).to(dtype)
    qa[..., D : D + gh] = rel_h.transpose(1, 2)
    qa[..., D + gh : D + gh + gw] = rel_w.transpose(1, 2)
    ka[..., :D] = k.reshape(B, L, H, D)
    idx = torch.arange(L, device=DEV)
    ka[:, idx, :, D + idx // gw] = 1
    ka[:, idx, :, D + gh + idx % gw] = 1
    out = torch.full((B, L, H * D), float("nan"), dtype=dtype, device=DEV)
    Lkp = (L + 63) // 64 * 64
    native.attention_general(qa.reshape(B, L, H * Dq), ka.reshape(B, L, H * Dq), _vt_from_v(v, Lkp), out, H, L, scale=1.0)
    if dtype == torch.bfloat16:  # q * scale is rounded to bf16 once more than in the reference: compare with that rounding applied
        qs = (q.reshape(B, L, H, D).float() * scale).to(dtype).float().transpose(1, 2)
        ref = (torch.softmax(qs @ kh.transpose(-1, -2) + bias, dim=-1) @ vh).transpose(1, 2).reshape(B, L, H * D)
    return _cmp(out, ref, dtype)


# ------------------------------------------------------------------------------------------------ norms
def layernorm_case(M, Cc, dtype, seed=80):
    x = _rand(M, Cc, dtype=dtype, seed=seed) * 2 + 0.5
    g = (1 + 0.1 * _rand(Cc, dtype=torch.float32, seed=seed + 1)).to(dtype)
    b = (0.1 * _rand(Cc, dtype=torch.float32, seed=seed + 2)).to(dtype)
    out = torch.empty_like(x)
    native.layernorm(x, g, b, 1e-5, out)
    ref = F.layer_norm(x.float(), (Cc,), g.float(), b.float(), 1e-5)
    return _cmp(out, ref, dtype)


def groupnorm_case(B, Cc, HW, dtype, silu=True, eps=1e-5, seed=90):
    x = _rand(B, HW, Cc, dtype=dtype, seed=seed) * 1.5 + 3.0  # large mean: stresses the variance computation
    g = (1 + 0.1 * _rand(Cc, dtype=torch.float32, seed=seed + 1)).to(dtype)
    b = (0.1 * _rand(Cc, dtype=torch.float32, seed=seed + 2)).to(dtype)
    out = torch.empty_like(x)
    native.groupnorm_nhwc(x, g, b, 32, eps, silu, out)
    xr = x.float().permute(0, 2, 1)  # [B, C, HW]
    ref = F.group_norm(xr, 32, g.float(), b.float(), eps)
    if silu:
        ref = F.silu(ref)
    return _cmp(out.float().permute(0, 2, 1), ref, dtype)


def colstats_case(M, K, N, dtype, *, tile=0, ksplit=1, res=True, seed=310):
    """mi355x_gemm_args.colstats_out: (sum, sum of squares) per (32-row block, column) of the output AS STORED, from the epilogue of every 4-wave
    tile and from the split-K reduction pass -- checked against the sums of the stored tensor itself, rows beyond M excluded."""
    x = _rand(M, K, dtype=dtype, seed=seed)
    w = _rand(N, K, dtype=dtype, seed=seed + 1, scale=K ** -0.5)
    b = _rand(N, dtype=dtype, seed=seed + 2) + 2.0
    r = _rand(M, N, dtype=dtype, seed=seed + 3) if res else None
    out = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    cs = torch.full(native.colstats_shape(M, N), float("nan"), dtype=torch.float32, device=DEV)
    ws = torch.empty(ksplit * M * N, dtype=torch.float32, device=DEV) if ksplit > 1 else None
    native.gemm([(x, native.KBlocked(w))], out, bias=b, res=r, tile=tile, ksplit=ksplit, ws=ws, colstats_out=cs)
    ref = x.float() @ w.float().t() + b.float() + (r.float() if res else 0)
    e_out = _cmp(out, ref, dtype)
    nb = (M + 31) // 32
    pad = torch.zeros(nb * 32, N, dtype=torch.float64, device=DEV)
    pad[:M] = out.double()
    blocks = pad.view(nb, 32, N)
    want = torch.stack([blocks.sum(1), blocks.square().sum(1)], dim=-1)
    assert torch.isfinite(cs).all(), "colstats has unwritten entries"
    e_cs = ((cs.double() - want).abs() / (want.abs() + 1.0)).max().item()
    assert e_cs < 2e-6, f"column statistics off by {e_cs:.2e}"
    cs2 = torch.zeros_like(cs)
    native.gemm([(x, native.KBlocked(w))], out, bias=b, res=r, tile=tile, ksplit=ksplit, ws=ws, colstats_out=cs2)
    assert torch.equal(cs, cs2), "column statistics must be bit-reproducible"
    return e_out


def conv_groupnorm_chain_case(B, Cin, Cout, H, W, dtype, *, ksplit=1, tile=0, silu=True, seed=320, dc=1.5, wscale=1.0, chain_tol=None):
    """Conv2d -> GroupNorm (-> SiLU) with the statistics taken from the convolution's epilogue (two GroupNorm launches instead of three)
    against torch's conv2d + group_norm, and against the three-kernel GroupNorm on the same stored tensor."""
    x = _rand(B, Cin, H, W, dtype=dtype, seed=seed)
    w = _rand(Cout, Cin, 3, 3, dtype=dtype, seed=seed + 1, scale=wscale * (Cin * 9) ** -0.5)
    b = _rand(Cout, dtype=dtype, seed=seed + 2) * wscale + dc  # (dc: the channels' common offset -- raw-moment statistics lose mean^2 / var of their digits)
    g = (1 + 0.1 * _rand(Cout, dtype=torch.float32, seed=seed + 3)).to(dtype)
    be = (0.1 * _rand(Cout, dtype=torch.float32, seed=seed + 4)).to(dtype)
    M = B * H * W
    y = torch.empty(M, Cout, dtype=dtype, device=DEV)
    cs = torch.full(native.colstats_shape(M, Cout), float("nan"), dtype=torch.float32, device=DEV)
    ws = torch.empty(ksplit * M * Cout, dtype=torch.float32, device=DEV) if ksplit > 1 else None
    native.conv_gemm([(x.permute(0, 2, 3, 1).contiguous(), native.KBlocked(native.pack_conv_weight(w)), 3, 1, 1)], y, B, H, W, bias=b, tile=tile, ksplit=ksplit, ws=ws, colstats_out=cs)
    o1, o2 = torch.empty_like(y), torch.empty_like(y)
    native.groupnorm_nhwc(y.view(B, H * W, Cout), g, be, 32, 1e-5, silu, o1.view(B, H * W, Cout), colstats=cs)
    native.groupnorm_nhwc(y.view(B, H * W, Cout), g, be, 32, 1e-5, silu, o2.view(B, H * W, Cout))
    d = (o1.float() - o2.float()).abs().max().item()
    assert d <= (chain_tol if chain_tol is not None else 2e-2 if dtype == torch.bfloat16 else 2e-5), f"producer statistics vs statistics pass: {d:.2e}"
    ref = F.group_norm(y.float().view(B, H * W, Cout).permute(0, 2, 1), 32, g.float(), be.float(), 1e-5)
    if silu:
        ref = F.silu(ref)
    return _cmp(o1.float().view(B, H * W, Cout).permute(0, 2, 1), ref, dtype)


def groupnorm_two_source_case(B, C1, C2, HW, dtype, *, stats=False, silu=True, seed=330):
    """GroupNorm over Concatenate(x, x2) that never exists (ResidualConcatenator -> ResidualBlock): two-source kernels, optionally with each
    part's column statistics (as its producer's epilogue would have written them) -- against torch.group_norm of the real concatenation.
    The group width (C1 + C2) / 32 need not divide C1: groups straddle the two sources (1280 + 640 channels: 60 per group)."""
    x = _rand(B, HW, C1, dtype=dtype, seed=seed) * 1.3 + 2.0
    x2 = _rand(B, HW, C2, dtype=dtype, seed=seed + 1) * 0.7 - 1.0
    Cc = C1 + C2
    g = (1 + 0.1 * _rand(Cc, dtype=torch.float32, seed=seed + 2)).to(dtype)
    b = (0.1 * _rand(Cc, dtype=torch.float32, seed=seed + 3)).to(dtype)
    out = torch.full((B, HW, Cc), float("nan"), dtype=dtype, device=DEV)
    cs = cs2 = None
    if stats:
        def colstats(t):
            blocks = t.float().reshape(B * HW // 32, 32, t.shape[2])
            return torch.stack([blocks.sum(1), blocks.square().sum(1)], dim=-1).contiguous()
        cs, cs2 = colstats(x), colstats(x2)
    native.groupnorm_nhwc(x, g, b, 32, 1e-5, silu, out, x2=x2, colstats=cs, colstats2=cs2)
    ref = F.group_norm(torch.cat([x.float(), x2.float()], dim=2).permute(0, 2, 1), 32, g.float(), b.float(), 1e-5)
    if silu:
        ref = F.silu(ref)
    return _cmp(out.float().permute(0, 2, 1), ref, dtype)


# ------------------------------------------------------------------------------------------------ glue
def layout_case(B, Cc, H, W, dtype, seed=100):
    x = _rand(B, Cc, H, W, dtype=dtype, seed=seed)
    nhwc = torch.empty(B, H * W, Cc, dtype=dtype, device=DEV)
    native.nchw_to_nhwc(x, nhwc)
    e1 = (nhwc.float() - x.float().permute(0, 2, 3, 1).reshape(B, H * W, Cc)).abs().max().item()
    back = torch.empty_like(x)
    native.nhwc_to_nchw(nhwc, back, Cc)
    e2 = (back.float() - x.float()).abs().max().item()
    return max(e1, e2), 1.0, 0.0


def concat_axpby_case(M, C1, C2, dtype, seed=110):
    a = _rand(M, C1, dtype=dtype, seed=seed)
    b = _rand(M, C2, dtype=dtype, seed=seed + 1)
    out = torch.empty(M, C1 + C2, dtype=dtype, device=DEV)
    native.concat2(a, b, out)
    e1 = (out.float() - torch.cat([a, b], 1).float()).abs().max().item()
    c = _rand(M, C1, dtype=dtype, seed=seed + 2)
    o2 = torch.empty_like(a)
    native.axpby(a, 1.0, c, 0.55, o2)
    ref = a.float() + 0.55 * c.float()
    e2 = (o2.float() - ref).abs().max().item() / ref.abs().max().item()
    return max(e1, e2), 1.0, (0.0 if dtype == torch.float32 else 8e-3) + 1e-6


def cfg_ddim_case(n, dtype, seed=120):
    x = _rand(n, dtype=dtype, seed=seed)
    uo = _rand(2 * n, dtype=dtype, seed=seed + 1)
    coef = torch.tensor([5.0, 0.8, 0.6, 0.9, math.sqrt(1 - 0.81)], dtype=torch.float32, device=DEV)
    xf, u, c = x.float(), uo[:n].float(), uo[n:].float()
    eps = u + 5.0 * (c - u)
    x0 = (xf - 0.6 * eps) / 0.8
    ref = 0.9 * x0 + math.sqrt(1 - 0.81) * eps
    native.cfg_ddim_step(x, uo, coef)
    return _cmp(x, ref, dtype)


def gemm_prefetch_case(M, K, N, dtype, ksplit=1, seed=37):
    """mi355x_gemm_args.prefetch: extra workgroups at the head of the grid touch another buffer; the tile -> workgroup mapping is
    shifted by them, the result must be bit-identical to the launch without prefetch (also under split-K)."""
    x = _rand(M, K, dtype=dtype, seed=seed)
    w = _rand(N, K, dtype=dtype, seed=seed + 1, scale=K ** -0.5)
    b = _rand(N, dtype=dtype, seed=seed + 2)
    r = _rand(M, N, dtype=dtype, seed=seed + 3)
    other = _rand(1237, 129, dtype=dtype, seed=seed + 4)  # a span whose size is no multiple of anything
    ws = torch.empty(max(ksplit, 1) * M * N, dtype=torch.float32, device=DEV) if ksplit > 1 else None
    out0 = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    out1 = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    native.gemm([(x, w)], out0, bias=b, res=r, ksplit=ksplit, ws=ws)
    native.gemm([(x, w)], out1, bias=b, res=r, ksplit=ksplit, ws=ws, prefetch=other)
    assert torch.equal(out0, out1), "prefetch workgroups changed the result"
    ref = x.float() @ w.float().t() + b.float() + r.float()
    return _cmp(out1, ref, dtype)


def gemm_kblocked_case(M, K, N, dtype, which="w", seed=38):
    """K-blocked operand layout ([K block][row][128 B]): same arithmetic in the same order, so the result must be bit-identical
    to the row-major launch; `which` = "w" (weights in the w slot), "x" (the transposed projections) or "both"."""
    x = _rand(M, K, dtype=dtype, seed=seed)
    w = _rand(N, K, dtype=dtype, seed=seed + 1, scale=K ** -0.5)
    b = _rand(N, dtype=dtype, seed=seed + 2)
    out0 = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    out1 = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    native.gemm([(x, w)], out0, bias=b)
    xa = native.KBlocked(x) if which in ("x", "both") else x
    wa = native.KBlocked(w) if which in ("w", "both") else w
    native.gemm([(xa, wa)], out1, bias=b)
    assert torch.equal(out0, out1), "K-blocked operand changed the result"
    y = x.float() @ w.float().t() + b.float()
    return _cmp(out1, y, dtype)


def conv_kblocked_case(B, Cin, Cout, H, W, dtype, ksplit=1, seed=39):
    """3x3 convolution with K-blocked packed weights vs the same launch with row-major packed weights (bit-identical)."""
    x = _rand(B, H, W, Cin, dtype=dtype, seed=seed)
    w4 = _rand(Cout, Cin, 3, 3, dtype=dtype, seed=seed + 1, scale=(Cin * 9) ** -0.5)
    wp = native.pack_conv_weight(w4)
    b = _rand(Cout, dtype=dtype, seed=seed + 2)
    ws = torch.empty(ksplit * B * H * W * Cout, dtype=torch.float32, device=DEV) if ksplit > 1 else None
    out0 = torch.full((B * H * W, Cout), float("nan"), dtype=dtype, device=DEV)
    out1 = torch.full((B * H * W, Cout), float("nan"), dtype=dtype, device=DEV)
    native.conv_gemm([(x, wp, 3, 1, 1)], out0, B, H, W, bias=b, ksplit=ksplit, ws=ws, tile=1 if ksplit > 1 else 0)
    native.conv_gemm([(x, native.KBlocked(wp), 3, 1, 1)], out1, B, H, W, bias=b, ksplit=ksplit, ws=ws, tile=1 if ksplit > 1 else 0)
    assert torch.equal(out0, out1), "K-blocked conv weights changed the result"
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w4.float(), b.float(), padding=1).permute(0, 2, 3, 1).reshape(B * H * W, Cout)
    return _cmp(out1, ref, dtype)


def gemm_quick_gelu_case(M, K, N, dtype, seed=36):
    """CLIP-L's FeedForward activation, x * sigmoid(1.702 x) (GeLUApproximation.SIGMOID), as the GEMM epilogue."""
    x = _rand(M, K, dtype=dtype, seed=seed)
    w = _rand(N, K, dtype=dtype, seed=seed + 1, scale=K ** -0.5)
    b = _rand(N, dtype=dtype, seed=seed + 2)
    out = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    native.gemm([(x, w)], out, bias=b, gelu="quick")
    y = x.float() @ w.float().t() + b.float()
    return _cmp(out, y * torch.sigmoid(1.702 * y), dtype)


def gemm_gelu_case(M, K, N, dtype, seed=35):
    x = _rand(M, K, dtype=dtype, seed=seed)
    w = _rand(N, K, dtype=dtype, seed=seed + 1, scale=K ** -0.5)
    b = _rand(N, dtype=dtype, seed=seed + 2)
    r = _rand(M, N, dtype=dtype, seed=seed + 3)
    out = torch.empty(M, N, dtype=dtype, device=DEV)
    native.gemm([(x, w)], out, bias=b, res=r, gelu=True)
    ref = F.gelu(x.float() @ w.float().t() + b.float(), approximate="none") + r.float()
    return _cmp(out, ref, dtype)


def patchify_gather_case(dtype, seed=140):
    x = _rand(2, 3, 64, 48, dtype=dtype, seed=seed)
    P = 16
    cols = torch.empty(2 * 4 * 3, 3 * P * P, dtype=dtype, device=DEV)
    native.patchify_nchw(x, P, cols)
    ref = F.unfold(x.float(), kernel_size=P, stride=P).transpose(1, 2).reshape(-1, 3 * P * P)
    e1 = (cols.float() - ref).abs().max().item()
    src = _rand(50, 64, dtype=dtype, seed=seed + 1)
    idx = torch.tensor([3, -1, 49, 0, 7, -1, 7], dtype=torch.int32, device=DEV)
    out = torch.full((7, 64), float("nan"), dtype=dtype, device=DEV)
    native.gather_rows(src, idx, out)
    want = torch.where(idx[:, None] >= 0, src[idx.clamp(min=0).long()], torch.zeros_like(src[:7]))
    e2 = (out.float() - want.float()).abs().max().item()
    return max(e1, e2), 1.0, 0.0


def sinusoidal_case(n, dim, group, dtype, seed=130):
    x = (torch.rand(n, generator=torch.Generator().manual_seed(seed)) * 1000).to(DEV)
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32, device=DEV) / half
    ang = x.unsqueeze(1) * torch.exp(exponent).unsqueeze(0)
    ref = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1).reshape(n // group, group * dim)
    col0 = 16
    out = torch.zeros(n // group, col0 + group * dim + 8, dtype=dtype, device=DEV)
    native.sinusoidal(x, dim, out, group=group, col0=col0)
    assert float(out[:, :col0].abs().max()) == 0 and float(out[:, col0 + group * dim :].abs().max()) == 0
    # cos / sin of arguments up to 1000 rad: float32 argument rounding alone is ~6e-5 absolute
    return (out[:, col0 : col0 + group * dim].float() - ref).abs().max().item(), 1.0, (2e-4 if dtype == torch.float32 else 8e-3)


# ------------------------------------------------------------------------------------------------ round-2 kernel features
def gemm_tile_case(M, K, N, dtype, tile, stages, *, res=True, prefetch=False, seed=200):
    """A forced (tile, LDS depth) pair, incl. tile 6 = 8 waves in two K groups: every configuration the tuner may pick."""
    x = _rand(M, K, dtype=dtype, seed=seed)
    w = _rand(N, K, dtype=dtype, seed=seed + 1, scale=K ** -0.5)
    b = _rand(N, dtype=dtype, seed=seed + 2)
    r = _rand(M, N, dtype=dtype, seed=seed + 3) if res else None
    pf = _rand(1 << 20, dtype=dtype, seed=seed + 4) if prefetch else None
    out = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    native.gemm([(x, native.KBlocked(w))], out, bias=b, res=r, tile=tile, stages=stages, prefetch=pf)
    out2 = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    native.gemm([(x, native.KBlocked(w))], out2, bias=b, res=r, tile=tile, stages=stages, prefetch=pf)
    assert torch.equal(out, out2), "forced tile must be bit-reproducible"
    ref = x.float() @ w.float().t() + b.float() + (r.float() if res else 0)
    return _cmp(out, ref, dtype)


def gemm_kgroups_multiseg_case(M, K1, K2, N, dtype, seed=210):
    """Two K groups over TWO segments with an odd total block count (the odd group idles through the last trip)."""
    x1, x2 = _rand(M, K1, dtype=dtype, seed=seed), _rand(M, K2, dtype=dtype, seed=seed + 1)
    w1, w2 = _rand(N, K1, dtype=dtype, seed=seed + 2, scale=(K1 + K2) ** -0.5), _rand(N, K2, dtype=dtype, seed=seed + 3, scale=(K1 + K2) ** -0.5)
    out = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    native.gemm([(x1, w1), (x2, native.KBlocked(w2))], out, tile=6)
    ref = x1.float() @ w1.float().t() + x2.float() @ w2.float().t()
    return _cmp(out, ref, dtype)


def gemm_multiseg_case(M, K1, K2, N, dtype, tile, seed=212):
    """Two K segments (one K-blocked) with an odd total K tile count on a forced tile; bias + residual."""
    x1, x2 = _rand(M, K1, dtype=dtype, seed=seed), _rand(M, K2, dtype=dtype, seed=seed + 1)
    w1, w2 = _rand(N, K1, dtype=dtype, seed=seed + 2, scale=(K1 + K2) ** -0.5), _rand(N, K2, dtype=dtype, seed=seed + 3, scale=(K1 + K2) ** -0.5)
    b, r = _rand(N, dtype=dtype, seed=seed + 4), _rand(M, N, dtype=dtype, seed=seed + 5)
    out = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    native.gemm([(x1, w1), (x2, native.KBlocked(w2))], out, bias=b, res=r, tile=tile)
    ref = x1.float() @ w1.float().t() + x2.float() @ w2.float().t() + b.float() + r.float()
    return _cmp(out, ref, dtype)


def conv_forced_tile_case(dtype, tile, seed=222):
    """The conv loader's address arithmetic on a forced tile: stride 2, nearest-2x upsampling, and a fused 1x1 shortcut segment (three runs)."""
    worst = (0.0, 1e-6, _tol(dtype))
    for kw in ({"stride": 2}, {"ups": 2}, {"split": 640, "rowbias": True, "res": True}):
        e = conv_case(1, 960 if "split" in kw else 320, 320, 16, 16, dtype, seed=seed, tile=tile, **kw)
        if e[0] / e[1] > worst[0] / worst[1]:
            worst = e
    return worst


def conv_tile_case(B, Cin, Cout, H, W, dtype, tile, stages, ksplit=1, seed=220):
    x = _rand(B, H, W, Cin, dtype=dtype, seed=seed)
    w4 = _rand(Cout, Cin, 3, 3, dtype=dtype, seed=seed + 1, scale=(9 * Cin) ** -0.5)
    b = _rand(Cout, dtype=dtype, seed=seed + 2)
    out = torch.full((B * H * W, Cout), float("nan"), dtype=dtype, device=DEV)
    ws = torch.empty(ksplit * B * H * W * Cout, dtype=torch.float32, device=DEV) if ksplit > 1 else None
    native.conv_gemm([(x, native.KBlocked(native.pack_conv_weight(w4)), 3, 1, 1)], out, B, H, W, bias=b, tile=tile, stages=stages, ksplit=ksplit, ws=ws)
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w4.float(), b.float(), padding=1).permute(0, 2, 3, 1).reshape(B * H * W, Cout)
    return _cmp(out, ref, dtype)


def gemm_qkv_case(M, K, Cc, dtype, tile=0, bias=False, seed=230, pad=0):
    """One launch over [Wq; Wk; Wv]: Q | K row-major into `out`, V transposed into `out_t` (transposed column group);
    pad = extra elements per V^T row (a row stride that is not the row count; the padding must stay untouched)."""
    x = _rand(M, K, dtype=dtype, seed=seed)
    w = _rand(3 * Cc, K, dtype=dtype, seed=seed + 1, scale=K ** -0.5)
    b = _rand(3 * Cc, dtype=dtype, seed=seed + 2) if bias else None
    qk = torch.full((M, 2 * Cc), float("nan"), dtype=dtype, device=DEV)
    vt_full = torch.full((Cc, M + pad), 7.0, dtype=dtype, device=DEV)
    vt = vt_full[:, :M]
    native.gemm([(x, native.KBlocked(w))], qk, bias=b, out_t=vt, nt_begin=2 * Cc, tile=tile)
    ref = x.float() @ w.float().t() + (b.float() if bias else 0)
    e1 = _cmp(qk, ref[:, : 2 * Cc], dtype)
    e2 = _cmp(vt, ref[:, 2 * Cc :].t(), dtype)
    spill = float((vt_full[:, M:].float() - 7.0).abs().max().item()) if pad else 0.0
    return max(e1[0], e2[0]) + spill, max(e1[1], e2[1]), e1[2]


def gemm_t_only_case(M, K, Cc, dtype, tile=0, seed=235):
    """nt_begin = 0: the whole output transposed (out = NULL), ragged M (scalar row tail) and a column count below one tile."""
    x = _rand(M, K, dtype=dtype, seed=seed)
    w = _rand(Cc, K, dtype=dtype, seed=seed + 1, scale=K ** -0.5)
    ldt = (M + 7) // 8 * 8
    vt = torch.full((Cc, ldt), float("nan"), dtype=dtype, device=DEV)
    native.gemm([(x, w)], None, out_t=vt, nt_begin=0, tile=tile)
    return _cmp(vt[:, :M], (x.float() @ w.float().t()).t(), dtype)


def _stats_ref(y: torch.Tensor) -> torch.Tensor:
    """[N / 32][M][2] (mean, M2) of consecutive 32-column chunks, float64 -> float32."""
    M, N = y.shape
    c = y.double().view(M, N // 32, 32)
    mean = c.mean(-1)
    m2 = ((c - mean.unsqueeze(-1)) ** 2).sum(-1)
    return torch.stack([mean, m2], -1).permute(1, 0, 2).float().contiguous()


def gemm_ln_chain_case(M, Cc, N2, dtype, *, geglu=False, tile1=0, tile2=0, transposed=False, seed=240):
    """Producer GEMM (+ residual) writes its rows' statistics; the consumer applies LayerNorm + Linear in one launch.
    Reference: torch layer_norm of the STORED producer output, then the Linear (+ GEGLU / transposed store)."""
    x = _rand(M, Cc, dtype=dtype, seed=seed)
    w1 = _rand(Cc, Cc, dtype=dtype, seed=seed + 1, scale=Cc ** -0.5)
    b1 = _rand(Cc, dtype=dtype, seed=seed + 2)
    r = _rand(M, Cc, dtype=dtype, seed=seed + 3) + 3.0  # a non-zero row mean: exercises the mean * s cancellation
    gamma = (1 + 0.2 * _rand(Cc, dtype=torch.float32, seed=seed + 4)).float()
    beta = (0.3 * _rand(Cc, dtype=torch.float32, seed=seed + 5)).float()
    w2 = _rand(N2, Cc, dtype=dtype, seed=seed + 6, scale=Cc ** -0.5)
    b2 = _rand(N2, dtype=dtype, seed=seed + 7)
    y = torch.full((M, Cc), float("nan"), dtype=dtype, device=DEV)
    stats = torch.full((Cc // 32, M, 2), float("nan"), dtype=torch.float32, device=DEV)
    native.gemm([(x, w1)], y, bias=b1, res=r, stats_out=stats, tile=tile1)
    yref = x.float() @ w1.float().t() + b1.float() + r.float()
    e0 = _cmp(y, yref, dtype)
    sref = _stats_ref(y.float())
    es = (stats - sref).abs().max().item() / max(sref.abs().max().item(), 1e-6)
    assert es < 2e-5, f"row statistics off by {es:.2e}"
    eps = 1e-5
    h = torch.nn.functional.layer_norm(y.float(), (Cc,), gamma, beta, eps)
    wsrc, bsrc = w2, b2
    if geglu:
        idx = native.geglu_pack_index(N2 // 2, device=DEV)
        wsrc, bsrc = w2[idx].contiguous(), b2[idx].contiguous()
    wl = (wsrc.float() * gamma.unsqueeze(0)).to(dtype).contiguous()
    ls = wl.float().sum(1).contiguous()
    lc = (wsrc.float() @ beta + bsrc.float()).contiguous()
    full = h @ w2.float().t() + b2.float()
    if transposed:
        out_t = torch.full((N2, M), float("nan"), dtype=dtype, device=DEV)
        native.gemm([(y, wl)], None, out_t=out_t, nt_begin=0, ln=(stats, ls, lc, eps), tile=tile2)
        e1 = _cmp(out_t, full.t(), dtype)
    elif geglu:
        out = torch.full((M, N2 // 2), float("nan"), dtype=dtype, device=DEV)
        native.gemm([(y, native.KBlocked(wl))], out, geglu=True, ln=(stats, ls, lc, eps), tile=tile2)
        e1 = _cmp(out, full[:, : N2 // 2] * torch.nn.functional.gelu(full[:, N2 // 2 :]), dtype)
    else:
        out = torch.full((M, N2), float("nan"), dtype=dtype, device=DEV)
        native.gemm([(y, native.KBlocked(wl))], out, ln=(stats, ls, lc, eps), tile=tile2)
        e1 = _cmp(out, full, dtype)
    return max(e0[0], e1[0]), max(e0[1], e1[1]), e1[2] * (2.0 if dtype == torch.bfloat16 else 1.0)


def wide_head_attention_case(L, D, dtype, seed=250):
    """The VAE mid-block head (one head of 512 over H*W tokens) as S = Q K^T (float32 scores), row softmax, O = P V."""
    q = _rand(L, D, dtype=dtype, seed=seed)
    k = _rand(L, D, dtype=dtype, seed=seed + 1)
    v = _rand(L, D, dtype=dtype, seed=seed + 2)
    sc = torch.full((L, L), float("nan"), dtype=torch.float32, device=DEV)
    native.gemm([(q, k)], sc, out_f32=dtype != torch.float32)
    pr = torch.full((L, L), float("nan"), dtype=dtype, device=DEV)
    native.softmax_rows(sc, pr, L, D ** -0.5)
    o = torch.full((L, D), float("nan"), dtype=dtype, device=DEV)
    native.gemm([(pr, v.t().contiguous())], o)
    ref = torch.softmax((q.float() @ k.float().t()) * D ** -0.5, dim=-1) @ v.float()
    e0 = _cmp(sc, q.float() @ k.float().t(), torch.float32)
    assert e0[0] <= 2e-3 * e0[1] if dtype == torch.float32 else e0[0] <= 1e-5 * e0[1] + 1e-3, "raw scores must be float32-accurate"
    return _cmp(o, ref, dtype)


def softmax_rows_case(M, L, Lp, dtype, seed=255):
    s = _rand(M, Lp + 4, dtype=torch.float32, seed=seed, scale=6.0)
    out = torch.full((M, Lp), float("nan"), dtype=dtype, device=DEV)
    native.softmax_rows(s[:, : Lp], out, L, 0.37)
    ref = torch.zeros(M, Lp, device=DEV)
    ref[:, :L] = torch.softmax(s[:, :L] * 0.37, dim=-1)
    assert float(out[:, L:].float().abs().max()) == 0.0 if Lp > L else True
    return _cmp(out, ref, dtype)


def _lora_pack(K, N, dtype, ranks, seed, perm=None):
    """(A K-blocked [R, K], sB [N, R], dense reference delta [N, K] float32) of stacked LoRAs with scales 1.0, 0.8, ...; R = the stacked
    rank rounded up to 32."""
    rt = sum(ranks)
    R = native.lora_rank(rt)
    a = torch.zeros(R, K, dtype=dtype, device=DEV)
    bs = torch.zeros(N, R, dtype=dtype, device=DEV)
    delta = torch.zeros(N, K, dtype=torch.float32, device=DEV)
    o = 0
    for i, r in enumerate(ranks):
        d = _rand(r, K, dtype=dtype, seed=seed + 10 * i, scale=K ** -0.5)
        u = _rand(N, r, dtype=dtype, seed=seed + 10 * i + 1, scale=0.5)
        sc = 1.0 - 0.2 * (i % 4)
        a[o : o + r] = d
        bs[:, o : o + r] = (u.float() * sc).to(dtype)
        delta += bs[:, o : o + r].float() @ d.float()
        o += r
    assert R
    if perm is not None:
        bs = bs[perm].contiguous()
    return native.KBlocked(a), bs, delta


def gemm_lora_inlaunch_case(M, K, N, dtype, *, ranks=(16, 16), tile=0, geglu=False, transposed=False, seed=260):
    """LoraAdapter as ONE launch: producer workgroups compute x A_cat^T once per row block, the pre-scaled up-projections are the tiles' last K steps."""
    x = _rand(M, K, dtype=dtype, seed=seed)
    w = _rand(N, K, dtype=dtype, seed=seed + 1, scale=K ** -0.5)
    b = _rand(N, dtype=dtype, seed=seed + 2)
    perm = native.geglu_pack_index(N // 2, device=DEV) if geglu else None
    a32, bs, delta = _lora_pack(K, N, dtype, ranks, seed + 3, perm)
    full = x.float() @ (w.float() + delta).t() + b.float()
    if transposed:
        vt = torch.full((N, M), float("nan"), dtype=dtype, device=DEV)
        native.gemm([(x, w)], None, bias=b, out_t=vt, nt_begin=0, lora=([(0, a32)], bs), tile=tile)
        return _cmp(vt, full.t(), dtype)
    if geglu:
        out = torch.full((M, N // 2), float("nan"), dtype=dtype, device=DEV)
        native.gemm([(x, native.KBlocked(w[perm].contiguous()))], out, bias=b[perm].contiguous(), geglu=True, lora=([(0, a32)], bs), tile=tile)
        return _cmp(out, full[:, : N // 2] * torch.nn.functional.gelu(full[:, N // 2 :]), dtype)
    r = _rand(M, N, dtype=dtype, seed=seed + 4)
    out = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    native.gemm([(x, native.KBlocked(w))], out, bias=b, res=r, lora=([(0, a32)], bs), tile=tile)
    out2 = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    native.gemm([(x, native.KBlocked(w))], out2, bias=b, res=r, lora=([(0, a32)], bs), tile=tile)
    assert torch.equal(out, out2)
    return _cmp(out, full + r.float(), dtype)


def with_lora_source(bits, fn):
    """Run a LoRA case with t forced to come from producer workgroups (64) or from t-tiles wherever the tile is wide enough (128): mi355x_set_option
    lora_dbg -- the two sources of the same hand-off (csrc/gemm_kernel.cuh, GemmP::lora_tt); the library's own choice is what the plain cases run."""
    lib = native.load()
    lib.mi355x_set_option(b"lora_dbg", bits)
    try:
        return fn()
    finally:
        lib.mi355x_set_option(b"lora_dbg", 0)


def _launch_stat(name: bytes) -> int:
    """mi355x_get_stat: launches since the library was loaded ("g8" = on the 8-wave loop, any tile id; "g8lora" = those with its in-launch LoRA; "g9" = those on 192-row tiles)."""
    lib = native.load()
    lib.mi355x_get_stat.argtypes = [ctypes.c_char_p]
    return int(lib.mi355x_get_stat(name))


def _on_path(stat: bytes, what: str, fn, launches: int):
    n0 = _launch_stat(stat)
    e = fn()
    n1 = _launch_stat(stat)
    assert n1 - n0 >= launches, f"expected {launches} launches {what}, saw {n1 - n0} (the launch fell back to another kernel)"
    return e


def on_g8_lora(fn, launches=1):
    """Run a case and require that at least `launches` of its launches took the in-launch LoRA on the 8-wave loop (csrc/gemm8_kernel.cuh: t-tiles at the head of
    the grid, the up-projection as every tile's tail) rather than falling back to the 4-wave kernel."""
    return _on_path(b"g8lora", "with the in-launch LoRA of the 8-wave loop", fn, launches)


def on_g8(fn, launches=1):
    """... ran on the 8-wave loop (any of its tile ids) instead of falling back."""
    return _on_path(b"g8", "on the 8-wave loop", fn, launches)


def on_tile11(fn, launches=1):
    """... ran the 8-wave loop as a two-height launch (192-row tiles + 128-row tiles, tile id 11)."""
    return _on_path(b"g11", "as a two-height launch of the 8-wave loop", fn, launches)


def on_tile9(fn, launches=1):
    """... ran the 8-wave loop on 192-row tiles."""
    return _on_path(b"g9", "on 192-row tiles of the 8-wave loop", fn, launches)


def gemm_qkv_lora_case(M, K, Cc, dtype, tile=0, seed=270):
    """Q | K | V^T from one launch with a different LoRA set per column group."""
    x = _rand(M, K, dtype=dtype, seed=seed)
    w = _rand(3 * Cc, K, dtype=dtype, seed=seed + 1, scale=K ** -0.5)
    packs = [_lora_pack(K, Cc, dtype, (16, 16) if g != 1 else (8,), seed + 20 * (g + 1)) for g in range(3)]
    bs = torch.cat([p_[1] for p_ in packs], 0).contiguous()
    delta = torch.cat([p_[2] for p_ in packs], 0)
    qk = torch.full((M, 2 * Cc), float("nan"), dtype=dtype, device=DEV)
    vt = torch.full((Cc, M), float("nan"), dtype=dtype, device=DEV)
    native.gemm([(x, native.KBlocked(w))], qk, out_t=vt, nt_begin=2 * Cc, lora=([(0, packs[0][0]), (Cc, packs[1][0]), (2 * Cc, packs[2][0])], bs), tile=tile)
    ref = x.float() @ (w.float() + delta).t()
    e1 = _cmp(qk, ref[:, : 2 * Cc], dtype)
    e2 = _cmp(vt, ref[:, 2 * Cc :].t(), dtype)
    return max(e1[0], e2[0]), max(e1[1], e2[1]), e1[2]


def gemm_ln_lora_case(M, Cc, N2, dtype, *, tile=0, transposed=False, seed=280):
    """LayerNorm AND the LoRAs folded into ONE launch: y = LN(x) W^T + b + sum s (LN(x) A^T) B^T from the un-normalised x."""
    x = _rand(M, Cc, dtype=dtype, seed=seed) + 1.5
    gamma = (1 + 0.2 * _rand(Cc, dtype=torch.float32, seed=seed + 4)).float()
    beta = (0.3 * _rand(Cc, dtype=torch.float32, seed=seed + 5)).float()
    w2 = _rand(N2, Cc, dtype=dtype, seed=seed + 6, scale=Cc ** -0.5)
    b2 = _rand(N2, dtype=dtype, seed=seed + 7)
    a32, bs, delta = _lora_pack(Cc, N2, dtype, (16, 16), seed + 8)
    stats = _stats_ref(x.float()).to(DEV)
    eps = 1e-5
    h = torch.nn.functional.layer_norm(x.float(), (Cc,), gamma, beta, eps)
    full = h @ (w2.float() + delta).t() + b2.float()
    wl = (w2.float() * gamma.unsqueeze(0)).to(dtype).contiguous()
    ls, lc = wl.float().sum(1).contiguous(), (w2.float() @ beta + b2.float()).contiguous()
    a = a32.dense().float()
    al = (a * gamma.unsqueeze(0)).to(dtype).contiguous()
    als, alc = al.float().sum(1).contiguous(), (a @ beta).contiguous()
    lo = ([(0, native.KBlocked(al))], bs, als, alc)
    if transposed:
        out_t = torch.full((N2, M), float("nan"), dtype=dtype, device=DEV)
        native.gemm([(x, wl)], None, out_t=out_t, nt_begin=0, ln=(stats, ls, lc, eps), lora=lo, tile=tile)
        e = _cmp(out_t, full.t(), dtype)
    else:
        out = torch.full((M, N2), float("nan"), dtype=dtype, device=DEV)
        native.gemm([(x, native.KBlocked(wl))], out, ln=(stats, ls, lc, eps), lora=lo, tile=tile)
        e = _cmp(out, full, dtype)
    return e[0], e[1], e[2] * (2.0 if dtype == torch.bfloat16 else 1.0)


def conv_lora_inlaunch_case(B, Cin, Cout, H, W, dtype, *, ranks=(16, 16), stride=1, shortcut=0, ksplit=1, tile=0, seed=290):
    """Conv2dLora inside the parent conv's launch: 3x3 down convs (the parent's stride / padding), 1x1 up convs; optionally a fused 1x1
    shortcut segment (not adapted) and split-K (the first split carries the LoRA term)."""
    x = _rand(B, H, W, Cin, dtype=dtype, seed=seed)
    w = _rand(Cout, Cin, 3, 3, dtype=dtype, seed=seed + 1, scale=(9 * Cin) ** -0.5)
    b = _rand(Cout, dtype=dtype, seed=seed + 2)
    rt = sum(ranks)
    R = native.lora_rank(rt)
    a = torch.zeros(R, 9 * Cin, dtype=dtype, device=DEV)
    bs = torch.zeros(Cout, R, dtype=dtype, device=DEV)
    xn = x.float().permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xn, w.float(), b.float(), stride=stride, padding=1)
    o = 0
    for i, r in enumerate(ranks):
        d = _rand(r, Cin, 3, 3, dtype=dtype, seed=seed + 10 * i + 3, scale=(9 * Cin) ** -0.5)
        u = _rand(Cout, r, dtype=dtype, seed=seed + 10 * i + 4, scale=0.5)
        sc = 1.0 - 0.2 * i
        a[o : o + r] = native.pack_conv_weight(d)
        bs[:, o : o + r] = (u.float() * sc).to(dtype)
        t = torch.nn.functional.conv2d(xn, d.float(), None, stride=stride, padding=1)
        ref = ref + torch.nn.functional.conv2d(t, bs[:, o : o + r].float()[:, :, None, None])
        o += r
    OH, OW = (H + stride - 1) // stride, (W + stride - 1) // stride
    segs = [(x, native.KBlocked(native.pack_conv_weight(w)), 3, stride, 1)]
    if shortcut:
        xs = _rand(B, OH, OW, shortcut, dtype=dtype, seed=seed + 7)
        ws = _rand(Cout, shortcut, dtype=dtype, seed=seed + 8, scale=shortcut ** -0.5)
        segs.append((xs, ws, 1, 1, 1))
        ref = ref + torch.nn.functional.conv2d(xs.float().permute(0, 3, 1, 2), ws.float()[:, :, None, None])
    out = torch.full((B * OH * OW, Cout), float("nan"), dtype=dtype, device=DEV)
    wsb = torch.empty(ksplit * out.numel(), dtype=torch.float32, device=DEV) if ksplit > 1 else None
    native.conv_gemm(segs, out, B, OH, OW, bias=b, lora=([(0, native.KBlocked(a))], bs), ksplit=ksplit, ws=wsb, tile=tile)
    return _cmp(out, ref.permute(0, 2, 3, 1).reshape(B * OH * OW, Cout), dtype)


def gemm_lora_repeat_case(M, K, N, dtype, seed=295, rounds=12, tiles=(1, 4, 2, 3)):
    """The hand-off under reuse: the SAME scratch / flags / epoch word driven through several launches with different inputs (what a
    replayed program does), every word of every result checked -- a stale t (flag seen early, L1-resident line) would show here."""
    w = _rand(N, K, dtype=dtype, seed=seed + 1, scale=K ** -0.5)
    a_kb, bs, delta = _lora_pack(K, N, dtype, (16, 16), seed + 3)
    sync = native.LoraSync(torch.device(DEV))
    t, flags = sync.scratch(1, M, 32, dtype), sync.flags(1, M)
    wk = native.KBlocked(w)
    worst = (0.0, 0.0, 1.0)
    for rnd in range(rounds):
        x = _rand(M, K, dtype=dtype, seed=seed + 50 + rnd)
        out = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
        sync.bump()
        native.gemm([(x, wk)], out, lora=([(0, a_kb)], bs), lora_sync=(t, flags, sync), tile=tiles[rnd % len(tiles)])
        e = _cmp(out, x.float() @ (w.float() + delta).t(), dtype)
        worst = (max(worst[0], e[0]), max(worst[1], e[1]), e[2])
    return worst


def all_cases():
    """(name, thunk) list; sizes chosen so the whole list runs in well under a minute on one MI355X."""
    cases = []
    for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        cases += [
            (f"gemm_{tag}_256x320x384", lambda dt=dt: gemm_case(256, 320, 384, dt)),
            (f"gemm_{tag}_2048x1280x1280_bias_res", lambda dt=dt: gemm_case(2048, 1280, 1280, dt, bias=True, res=True)),
            (f"gemm_{tag}_154x2048x640_Medge", lambda dt=dt: gemm_case(154, 2048, 640, dt, bias=True)),
            (f"gemm_{tag}_8x2048x1280_tinyM", lambda dt=dt: gemm_case(8, 2048, 1280, dt)),
            (f"gemm_{tag}_300x640x200_Nedge", lambda dt=dt: gemm_case(300, 640, 200, dt, bias=True, res=True)),
            (f"gemm_{tag}_2x1280x320_rowbiasless", lambda dt=dt: gemm_case(2, 1280, 320, dt, bias=True)),
            (f"gemm_{tag}_512x640x640_rowbias", lambda dt=dt: gemm_case(512, 640, 640, dt, bias=True, rowbias=256)),
            (f"gemm_{tag}_splitk2_512x1280x384", lambda dt=dt: gemm_splitk_case(512, 1280, 384, dt, 2)),
            (f"gemm_{tag}_splitk3_300x1920x200_edges", lambda dt=dt: gemm_splitk_case(300, 1920, 200, dt, 3, tile=1)),
            (f"gemm_{tag}_splitk4_tile2", lambda dt=dt: gemm_splitk_case(1024, 2560, 640, dt, 4, tile=2)),
            (f"conv_{tag}_3x3_splitk2", lambda dt=dt: conv_splitk_case(2, 640, 384, 16, 16, dt, 2)),
            (f"conv_{tag}_3x3_splitk3", lambda dt=dt: conv_splitk_case(1, 320, 320, 32, 32, dt, 3)),
            (f"gemm_{tag}_lora_2048x640x640", lambda dt=dt: gemm_lora_case(2048, 640, 640, dt)),
            (f"gemm_{tag}_lora_154x2048x1280", lambda dt=dt: gemm_lora_case(154, 2048, 1280, dt)),
            (f"gemm_{tag}_geglu_1024x640x2560", lambda dt=dt: gemm_geglu_case(1024, 640, 2560, dt)),
            (f"gemm_{tag}_vt_1024x1280", lambda dt=dt: gemm_vt_case(1024, 1280, 1280, dt)),
            (f"conv_{tag}_3x3_320_32", lambda dt=dt: conv_case(2, 320, 320, 32, 32, dt, rowbias=True)),
            (f"conv_{tag}_3x3_s2", lambda dt=dt: conv_case(2, 320, 320, 32, 32, dt, stride=2)),
            (f"conv_{tag}_3x3_s2_asym_pad", lambda dt=dt: conv_asym_case(2, 128, 128, 32, 48, dt)),
            (f"conv_{tag}_3x3_ups2", lambda dt=dt: conv_case(1, 640, 640, 16, 16, dt, ups=2)),
            (f"conv_{tag}_1x1_res", lambda dt=dt: conv_case(2, 320, 640, 16, 16, dt, ksize=1, res=True)),
            (f"conv_{tag}_3x3_split_960", lambda dt=dt: conv_case(1, 960, 320, 16, 16, dt, split=640, rowbias=True, res=True)),
            (f"conv_{tag}_3x3_to4", lambda dt=dt: conv_case(2, 320, 4, 32, 32, dt)),
            (f"conv_{tag}_first_4to320", lambda dt=dt: conv_first_case(2, 32, 32, dt)),
            (f"attn_{tag}_self_1024", lambda dt=dt: attention_case(2, 4, 1024, 1024, dt)),
            (f"attn_{tag}_self_spike", lambda dt=dt: attention_case(1, 2, 256, 512, dt, spike=True)),
            (f"attn_{tag}_cross_77", lambda dt=dt: attention_case(2, 10, 1024, 77, dt)),
            (f"attn_{tag}_cross_77_ip4", lambda dt=dt: attention_case(2, 10, 512, 77, dt, ip_tokens=4)),
            (f"attn_{tag}_Lq_edge_200", lambda dt=dt: attention_case(1, 2, 200, 128, dt)),
            # tile counts around the two-tiles-in-flight loop's peel points (1, 3, 4, 5 tiles, ragged last tile), and a second stream of 3 tiles
            (f"attn_{tag}_1tile", lambda dt=dt: attention_case(1, 3, 160, 64, dt, seed=71)),
            (f"attn_{tag}_3tiles", lambda dt=dt: attention_case(2, 2, 192, 192, dt, seed=72)),
            (f"attn_{tag}_4tiles", lambda dt=dt: attention_case(1, 2, 256, 256, dt, seed=73)),
            (f"attn_{tag}_5tiles_ragged", lambda dt=dt: attention_case(1, 2, 130, 257, dt, seed=74)),
            (f"attn_{tag}_7tiles_ip130", lambda dt=dt: attention_case(1, 2, 128, 448, dt, ip_tokens=130, seed=75)),
        ]
        # the non-default variants of the tile loop: two K/V tiles in flight (2), permlane reductions (0x10), hoisted fragment reads (0x20)
        # key-split workgroups (0x20000 = forced): tile counts 1..16, ragged tails that leave key group 1 (or both halves of the last tile) masked
        for nm, args, kw in (("self_1024", (2, 4, 1024, 1024), {}), ("spike", (1, 2, 256, 512), {"spike": True}), ("Lk20", (1, 2, 96, 20), {"seed": 76}),
                             ("Lk33", (1, 2, 70, 33), {"seed": 77}), ("Lk257", (1, 2, 130, 257), {"seed": 74}), ("Lk300_Lq200", (2, 3, 200, 300), {"seed": 78}),
                             ("3tiles", (2, 2, 192, 192), {"seed": 72})):
            for code in (0x20001, 0x20011):
                cases.append((f"attn_{tag}_kvsplit{code & 0xff:02x}_{nm}", lambda dt=dt, args=args, kw=kw, code=code: attention_case(*args, dt, pipe=code, **kw)))
        # 64-query workgroups of four 16-query waves (0x30000 = forced; round 6: what short grids take by default): the same tile counts and ragged tails
        for nm, args, kw in (("self_1024", (2, 4, 1024, 1024), {}), ("spike", (1, 2, 256, 512), {"spike": True}), ("Lk20", (1, 2, 96, 20), {"seed": 76}),
                             ("Lk257_Lq130", (1, 2, 130, 257), {"seed": 74}), ("Lk300_Lq200", (2, 3, 200, 300), {"seed": 78}), ("Lq_edge_40", (1, 2, 40, 320), {"seed": 82})):
            cases.append((f"attn_{tag}_q16waves_{nm}", lambda dt=dt, args=args, kw=kw: attention_case(*args, dt, pipe=0x30011, **kw)))
        # round 6: lazy running maximum / row sums from the matrix pipe (OPT bits 2 / 3 of attn_kernel: 0x51, 0x91, 0xd1; bf16 one-stream launches) and the
        # software-pipelined loop (attn_pipe_kernel: bits 19-20 = 1 register-staged K/V, 2 the same with a free instruction order, 3 LDS-DMA = the default),
        # each with 32-query waves (0x10000: the short-grid switch off) and 16-query waves (0x30000), 0x40000 = few-tile shapes stay off the short kernel:
        # one tile, two tiles (the loop body never runs with a successor), odd / even tile counts, ragged last tiles, Lq edges, late score spikes
        if dt == torch.bfloat16:
            shapes = (("self_1024", (2, 4, 1024, 1024), {}), ("spike", (1, 2, 256, 512), {"spike": True}), ("Lk20", (1, 2, 96, 20), {"seed": 76}), ("Lk64", (1, 3, 160, 64), {"seed": 71}),
                      ("Lk77", (2, 10, 256, 77), {"seed": 83}), ("Lk128", (1, 2, 200, 128), {}), ("Lk257_Lq130", (1, 2, 130, 257), {"seed": 74}), ("Lk300_Lq200", (2, 3, 200, 300), {"seed": 78}),
                      ("3tiles", (2, 2, 192, 192), {"seed": 72}), ("Lq_edge_40", (1, 2, 40, 320), {"seed": 82}), ("spike_1000", (1, 2, 100, 1000), {"spike": True, "seed": 84}))
            for nm, args, kw in shapes:
                for code in (0x51, 0x91, 0xD1, 0x800D1, 0x1000D1, 0x1800D1, 0x3800D1):  # (0x200000: the exponent's subtraction folded into Q K^T)
                    for q in (0x10000, 0x30000):
                        cases.append((f"attn_{tag}_r6_{code | q | 0x40000:06x}_{nm}", lambda dt=dt, args=args, kw=kw, c=code | q | 0x40000: attention_case(*args, dt, pipe=c, **kw)))
        # round 6: the short-K/V kernel's LDS-DMA form (bf16): half tiles (<= 32 valid keys) against full ones on both sides of the boundary, zero-filled lanes beyond Lk,
        # 128- / 64-query workgroups (0x800000 / 0x1000000), and the register-staged form it replaced (0x2000000)
        if dt == torch.bfloat16:
            for nm, args, kw in (("Lk32", (1, 2, 100, 32), {"seed": 85}), ("Lk33", (1, 2, 100, 33), {"seed": 86}), ("Lk96_ip32", (2, 3, 130, 96), {"ip_tokens": 32, "seed": 87}),
                                 ("Lk97_ip33", (2, 3, 130, 97), {"ip_tokens": 33, "seed": 88}), ("Lk77_ip4", (2, 10, 512, 77), {"ip_tokens": 4}), ("Lk128_ip64", (1, 2, 70, 128), {"ip_tokens": 64, "seed": 89}),
                                 ("Lk1_ip1", (1, 2, 64, 1), {"ip_tokens": 1, "seed": 90}), ("Lk190", (1, 2, 200, 190), {"seed": 91}), ("Lk17_ip1_spike", (2, 3, 300, 17), {"ip_tokens": 1, "spike": True, "seed": 80})):
                for code in (0x8000D1, 0x10000D1, 0x28000D1, 0x30000D1):
                    cases.append((f"attn_{tag}_short_r6_{code:07x}_{nm}", lambda dt=dt, args=args, kw=kw, c=code: attention_case(*args, dt, pipe=c, **kw)))
        if dt == torch.bfloat16:  # the general kernel's round-5 instance (fast = 0) stays covered
            cases += [(f"attng_{tag}_r5_d40_self", lambda dt=dt: attention_general_case(2, 8, 1024, 1024, 40, 40, dt, fast=0)),
                      (f"attng_{tag}_r5_d80_spike", lambda dt=dt: attention_general_case(2, 8, 256, 256, 80, 80, dt, spike=True, fast=0)),
                      (f"attng_{tag}_r5_d64_causal", lambda dt=dt: attention_general_case(2, 12, 77, 77, 64, 64, dt, causal=True, fast=0)),
                      (f"attng_{tag}_r5_qk208_v80", lambda dt=dt: attention_general_case(1, 2, 512, 512, 208, 80, dt, fast=0))]
        # launches of at most three K/V tiles take the all-tiles-up-front kernel by default (the cases above: cross_77, cross_77_ip4, 1tile, 3tiles,
        # Lq_edge_200); 0x40000 switches it off, so the same shapes also run through the general tile loop; and its remaining slot layouts
        cases += [
            (f"attn_{tag}_general_cross_77", lambda dt=dt: attention_case(2, 10, 1024, 77, dt, pipe=0x40011)),
            (f"attn_{tag}_general_cross_77_ip4", lambda dt=dt: attention_case(2, 10, 512, 77, dt, ip_tokens=4, pipe=0x40011)),
            (f"attn_{tag}_general_1tile", lambda dt=dt: attention_case(1, 3, 160, 64, dt, seed=71, pipe=0x40011)),
            (f"attn_{tag}_short_1plus2tiles", lambda dt=dt: attention_case(1, 2, 100, 40, dt, ip_tokens=100, seed=79)),
            (f"attn_{tag}_short_1plus1_spike", lambda dt=dt: attention_case(2, 3, 300, 17, dt, ip_tokens=1, spike=True, seed=80)),
            (f"attn_{tag}_short_Lk1", lambda dt=dt: attention_case(1, 2, 64, 1, dt, seed=81)),
        ]
        for code in (0x02, 0x11, 0x21, 0x31, 0x32):
            cases += [
                (f"attn_{tag}_pipe{code:02x}_self_1024", lambda dt=dt, code=code: attention_case(2, 4, 1024, 1024, dt, pipe=code)),
                (f"attn_{tag}_pipe{code:02x}_5tiles_ragged", lambda dt=dt, code=code: attention_case(1, 2, 130, 257, dt, seed=74, pipe=code)),
                (f"attn_{tag}_pipe{code:02x}_cross_77_ip4", lambda dt=dt, code=code: attention_case(2, 10, 512, 77, dt, ip_tokens=4, spike=True, pipe=code | 0x40000)),
            ]
        cases += [
            (f"attng_{tag}_d40_self", lambda dt=dt: attention_general_case(2, 8, 1024, 1024, 40, 40, dt)),
            (f"attng_{tag}_d80_self", lambda dt=dt: attention_general_case(2, 8, 256, 256, 80, 80, dt, spike=True)),
            (f"attng_{tag}_d160_self", lambda dt=dt: attention_general_case(2, 8, 64, 64, 160, 160, dt)),
            (f"attng_{tag}_d160_cross77", lambda dt=dt: attention_general_case(1, 8, 200, 77, 160, 160, dt)),
            (f"attng_{tag}_d128", lambda dt=dt: attention_general_case(1, 4, 130, 190, 128, 128, dt)),
            (f"attng_{tag}_d64_causal", lambda dt=dt: attention_general_case(2, 12, 77, 77, 64, 64, dt, causal=True)),
            (f"attng_{tag}_d80_causal_long", lambda dt=dt: attention_general_case(1, 4, 300, 300, 80, 80, dt, causal=True)),
            (f"attng_{tag}_qk112_v80", lambda dt=dt: attention_general_case(2, 4, 196, 196, 112, 80, dt)),
            (f"attng_{tag}_qk208_v80", lambda dt=dt: attention_general_case(1, 2, 512, 512, 208, 80, dt)),
            (f"attng_{tag}_d40_long_spike", lambda dt=dt: attention_general_case(1, 4, 100, 1000, 40, 40, dt, spike=True, seed=171)),
            (f"attng_{tag}_d80_causal_1000", lambda dt=dt: attention_general_case(1, 2, 1000, 1000, 80, 80, dt, causal=True, seed=172)),
            (f"attng_{tag}_relpos_14x14", lambda dt=dt: attention_relpos_case(2, 4, 14, 14, 80, dt)),
            (f"attng_{tag}_relpos_32x32", lambda dt=dt: attention_relpos_case(1, 2, 32, 32, 80, dt)),
            (f"layernorm_{tag}_640", lambda dt=dt: layernorm_case(1000, 640, dt)),
            (f"layernorm_{tag}_1280", lambda dt=dt: layernorm_case(2048, 1280, dt)),
            (f"groupnorm_{tag}_320_4096", lambda dt=dt: groupnorm_case(2, 320, 4096, dt)),
            (f"groupnorm_{tag}_1280_1024", lambda dt=dt: groupnorm_case(2, 1280, 1024, dt, eps=1e-6, silu=False)),
            (f"groupnorm_{tag}_2560_1024", lambda dt=dt: groupnorm_case(2, 2560, 1024, dt)),
            (f"groupnorm_{tag}_960_4096", lambda dt=dt: groupnorm_case(1, 960, 4096, dt)),
            (f"groupnorm_{tag}_640_333_ragged", lambda dt=dt: groupnorm_case(3, 640, 333, dt, seed=91)),  # pixel counts that are no multiple of the 4-deep unroll
            (f"groupnorm_{tag}_320_16384", lambda dt=dt: groupnorm_case(2, 320, 16384, dt, seed=92)),
            (f"layout_{tag}", lambda dt=dt: layout_case(2, 4, 32, 32, dt)),
            (f"layout_{tag}_320", lambda dt=dt: layout_case(1, 320, 16, 24, dt)),
            (f"concat_axpby_{tag}", lambda dt=dt: concat_axpby_case(1000, 640, 320, dt)),
            (f"cfg_ddim_{tag}", lambda dt=dt: cfg_ddim_case(4 * 128 * 128, dt)),
            (f"gemm_{tag}_gelu_res_1000x640x384", lambda dt=dt: gemm_gelu_case(1000, 640, 384, dt)),
            (f"gemm_{tag}_gelu_Nedge", lambda dt=dt: gemm_gelu_case(300, 640, 200, dt)),
            (f"gemm_{tag}_quick_gelu_154x768x3072", lambda dt=dt: gemm_quick_gelu_case(154, 768, 3072, dt)),
            (f"gemm_{tag}_geglu_kblocked_chain", lambda dt=dt: gemm_geglu_kblocked_chain_case(300, 640, 2560, 640, dt)),
            (f"gemm_{tag}_kblocked_w_300x5120x200", lambda dt=dt: gemm_kblocked_case(300, 5120, 200, dt)),
            (f"gemm_{tag}_kblocked_x_1280x640x2048", lambda dt=dt: gemm_kblocked_case(1280, 640, 2048, dt, which="x")),
            (f"gemm_{tag}_kblocked_both_154x768x320", lambda dt=dt: gemm_kblocked_case(154, 768, 320, dt, which="both")),
            (f"conv_{tag}_kblocked_3x3_640", lambda dt=dt: conv_kblocked_case(2, 640, 320, 16, 16, dt)),
            (f"conv_{tag}_kblocked_3x3_splitk3", lambda dt=dt: conv_kblocked_case(1, 640, 640, 16, 16, dt, ksplit=3)),
            (f"gemm_{tag}_prefetch_300x640x200", lambda dt=dt: gemm_prefetch_case(300, 640, 200, dt)),
            (f"gemm_{tag}_prefetch_2048x1280x1280", lambda dt=dt: gemm_prefetch_case(2048, 1280, 1280, dt)),
            (f"gemm_{tag}_prefetch_splitk3", lambda dt=dt: gemm_prefetch_case(512, 1920, 384, dt, ksplit=3)),
            (f"patchify_gather_{tag}", lambda dt=dt: patchify_gather_case(dt)),
            (f"sinusoidal_{tag}_timestep", lambda dt=dt: sinusoidal_case(2, 320, 1, dt)),
            (f"sinusoidal_{tag}_time_ids", lambda dt=dt: sinusoidal_case(12, 256, 6, dt)),
        ]
        for tile in (1, 2, 3, 4, 6):
            for st in ((2,) if tile == 6 else (2, 3, 4)):
                cases.append((f"gemm_{tag}_tile{tile}_s{st}_300x1472x328", lambda dt=dt, tile=tile, st=st: gemm_tile_case(300, 1472, 328, dt, tile, st)))
        cases += [
            (f"gemm_{tag}_tile6_2048x1280x1280_prefetch", lambda dt=dt: gemm_tile_case(2048, 1280, 1280, dt, 6, 2, prefetch=True)),
            (f"gemm_{tag}_tile6_oneblock", lambda dt=dt: gemm_tile_case(200, 128 // (4 if dt == torch.float32 else 2), 136, dt, 6, 2)),
            (f"gemm_{tag}_tile1_s4_short_k", lambda dt=dt: gemm_tile_case(256, 2 * 128 // (4 if dt == torch.float32 else 2), 256, dt, 1, 4)),
            (f"gemm_{tag}_tile6_two_segments_odd", lambda dt=dt: gemm_kgroups_multiseg_case(300, 640, 320 + 64, 200, dt)),
            (f"conv_{tag}_tile6", lambda dt=dt: conv_tile_case(2, 320, 320, 16, 16, dt, 6, 2)),
            (f"conv_{tag}_tile6_splitk2", lambda dt=dt: conv_tile_case(1, 640, 256, 16, 16, dt, 6, 2, ksplit=2)),
            (f"conv_{tag}_tile1_s3", lambda dt=dt: conv_tile_case(2, 320, 384, 16, 24, dt, 1, 3)),
            (f"gemm_{tag}_qkv_2048x1280", lambda dt=dt: gemm_qkv_case(2048, 1280, 1280, dt)),
            (f"gemm_{tag}_qkv_2048x1280_padded_vt", lambda dt=dt: gemm_qkv_case(2048, 1280, 1280, dt, pad=64)),
            (f"gemm_{tag}_qkv_1000x640_padded_vt_tile4", lambda dt=dt: gemm_qkv_case(1000, 640, 640, dt, tile=4, bias=True, pad=32, seed=231)),
            (f"gemm_{tag}_qkv_1000x640_bias_tile4", lambda dt=dt: gemm_qkv_case(1000, 640, 640, dt, tile=4, bias=True)),
            (f"gemm_{tag}_qkv_tile6", lambda dt=dt: gemm_qkv_case(512, 640, 384, dt, tile=6)),
            (f"gemm_{tag}_t_only_ragged", lambda dt=dt: gemm_t_only_case(77 * 2, 2048, 640, dt)),
            (f"gemm_{tag}_t_only_tile2_small_n", lambda dt=dt: gemm_t_only_case(300, 640, 72, dt, tile=2)),
            (f"gemm_{tag}_ln_chain_2048x1280", lambda dt=dt: gemm_ln_chain_case(2048, 1280, 1280, dt)),
            (f"gemm_{tag}_ln_chain_geglu", lambda dt=dt: gemm_ln_chain_case(512, 640, 5120, dt, geglu=True)),
            (f"gemm_{tag}_ln_chain_tiles_4_6", lambda dt=dt: gemm_ln_chain_case(1000, 640, 640, dt, tile1=4, tile2=6)),
            (f"gemm_{tag}_ln_chain_tiles_6_2", lambda dt=dt: gemm_ln_chain_case(300, 320, 384, dt, tile1=6, tile2=2)),
            (f"gemm_{tag}_ln_chain_tiles_1_3", lambda dt=dt: gemm_ln_chain_case(520, 640, 256, dt, tile1=1, tile2=3)),
            (f"gemm_{tag}_ln_chain_tiles_2_1", lambda dt=dt: gemm_ln_chain_case(520, 640, 256, dt, tile1=2, tile2=1)),
            # round 4: GroupNorm statistics from the producer's epilogue
            (f"colstats_{tag}_auto_2048x1280", lambda dt=dt: colstats_case(2048, 640, 1280, dt)),
            (f"colstats_{tag}_tile1_edge_rows", lambda dt=dt: colstats_case(96, 256, 320, dt, tile=1)),
            (f"colstats_{tag}_tile2", lambda dt=dt: colstats_case(224, 320, 192, dt, tile=2, res=False)),
            (f"colstats_{tag}_tile3", lambda dt=dt: colstats_case(320, 384, 640, dt, tile=3)),
            (f"colstats_{tag}_tile4_oddM", lambda dt=dt: colstats_case(77, 128, 64, dt, tile=4)),
            (f"colstats_{tag}_splitk3", lambda dt=dt: colstats_case(256, 3072, 320, dt, tile=1, ksplit=3)),
            (f"groupnorm2_{tag}_1280+1280", lambda dt=dt: groupnorm_two_source_case(2, 1280, 1280, 1024, dt)),
            (f"groupnorm2_{tag}_1280+640_straddling", lambda dt=dt: groupnorm_two_source_case(2, 1280, 640, 1024, dt)),
            (f"groupnorm2_{tag}_640+320_stats", lambda dt=dt: groupnorm_two_source_case(2, 640, 320, 4096, dt, stats=True)),
            (f"groupnorm2_{tag}_320+320_stats_nosilu", lambda dt=dt: groupnorm_two_source_case(1, 320, 320, 384, dt, stats=True, silu=False)),
            (f"groupnorm2_{tag}_1280+640_stats", lambda dt=dt: groupnorm_two_source_case(2, 1280, 640, 256, dt, stats=True)),
            (f"conv_gn_{tag}_32x32x320", lambda dt=dt: conv_groupnorm_chain_case(2, 320, 320, 32, 32, dt)),
            (f"conv_gn_{tag}_splitk_16x16x640", lambda dt=dt: conv_groupnorm_chain_case(2, 640, 640, 16, 16, dt, ksplit=3, tile=1)),
            (f"conv_gn_{tag}_24x16x960_nosilu", lambda dt=dt: conv_groupnorm_chain_case(1, 320, 960, 24, 16, dt, silu=False)),
            # a large common offset (mean 10, deviation 0.1): the cross-block sums are double (round-4 advisor); what remains is the float32 of the per-block moments
            (f"conv_gn_{tag}_dc_offset", lambda dt=dt: conv_groupnorm_chain_case(2, 320, 320, 32, 32, dt, dc=10.0, wscale=0.1, chain_tol=3e-2 if dt == torch.bfloat16 else 1e-3, seed=321)),
            (f"gemm_{tag}_lora1_2048x1280x1280", lambda dt=dt: gemm_lora_inlaunch_case(2048, 1280, 1280, dt)),
            (f"gemm_{tag}_lora1_rank8_tile4_edges", lambda dt=dt: gemm_lora_inlaunch_case(300, 640, 200, dt, ranks=(8,), tile=4)),
            (f"gemm_{tag}_lora1_tile2", lambda dt=dt: gemm_lora_inlaunch_case(520, 320, 384, dt, ranks=(16, 4), tile=2)),
            (f"gemm_{tag}_lora1_tile3", lambda dt=dt: gemm_lora_inlaunch_case(154, 2048, 640, dt, tile=3)),
            (f"gemm_{tag}_lora1_geglu", lambda dt=dt: gemm_lora_inlaunch_case(512, 640, 2560, dt, geglu=True)),
            (f"gemm_{tag}_lora1_transposed", lambda dt=dt: gemm_lora_inlaunch_case(154, 2048, 640, dt, transposed=True)),
            (f"gemm_{tag}_qkv_lora_1024x1280", lambda dt=dt: gemm_qkv_lora_case(1024, 1280, 1280, dt)),
            (f"gemm_{tag}_qkv_lora_tile4", lambda dt=dt: gemm_qkv_lora_case(512, 640, 384, dt, tile=4)),
            # round 4: one producer per row block serves all three column groups (64-row producers under the 128 x 128 tile, 32-row ones elsewhere);
            # odd M: the groups' t blocks start on 128-byte lines (padded group stride), rows beyond M are clamped / suppressed
            (f"gemm_{tag}_qkv_lora_tile1_shared64", lambda dt=dt: gemm_qkv_lora_case(1024, 1280, 1280, dt, tile=1)),
            (f"gemm_{tag}_qkv_lora_tile1_oddM", lambda dt=dt: gemm_qkv_lora_case(301, 640, 384, dt, tile=1)),
            (f"gemm_{tag}_qkv_lora_tile2_oddM", lambda dt=dt: gemm_qkv_lora_case(77, 640, 256, dt, tile=2)),
            (f"gemm_{tag}_lora1_oddM_tile4", lambda dt=dt: gemm_lora_inlaunch_case(333, 1280, 320, dt, tile=4)),
            (f"gemm_{tag}_ln_lora_1024x1280", lambda dt=dt: gemm_ln_lora_case(1024, 1280, 1280, dt)),
            (f"gemm_{tag}_ln_lora_tile4", lambda dt=dt: gemm_ln_lora_case(300, 640, 384, dt, tile=4)),
            (f"gemm_{tag}_ln_lora_transposed_tile3", lambda dt=dt: gemm_ln_lora_case(512, 640, 256, dt, tile=3, transposed=True)),
            (f"gemm_{tag}_lora1_rank64", lambda dt=dt: gemm_lora_inlaunch_case(520, 640, 384, dt, ranks=(32, 16, 8))),
            (f"gemm_{tag}_lora1_rank128", lambda dt=dt: gemm_lora_inlaunch_case(2048, 1280, 1280, dt, ranks=(128,))),
            (f"gemm_{tag}_lora1_rank128_tile4_to_3", lambda dt=dt: gemm_lora_inlaunch_case(300, 640, 200, dt, ranks=(64, 64), tile=4)),
            (f"gemm_{tag}_lora1_rank64_tile2", lambda dt=dt: gemm_lora_inlaunch_case(520, 320, 384, dt, ranks=(64,), tile=2)),
            (f"gemm_{tag}_lora1_rank96_transposed", lambda dt=dt: gemm_lora_inlaunch_case(154, 2048, 640, dt, ranks=(64, 32), transposed=True)),
            (f"gemm_{tag}_lora1_ff2_2048x1280x5120", lambda dt=dt: gemm_lora_inlaunch_case(2048, 5120, 1280, dt)),
            (f"gemm_{tag}_lora1_repeat_shared_scratch", lambda dt=dt: gemm_lora_repeat_case(1024, 640, 1280, dt)),
            (f"gemm_{tag}_lora1_repeat_shared_scratch_2048x1280x3840", lambda dt=dt: gemm_lora_repeat_case(2048, 1280, 3840, dt, rounds=8)),
            # the same hand-off with t from t-tiles (forced for one-group launches, whose default is producers) / from producers (forced for Q|K|V^T)
            (f"gemm_{tag}_lora1_2048x1280x1280_tt", lambda dt=dt: with_lora_source(128, lambda: gemm_lora_inlaunch_case(2048, 1280, 1280, dt))),
            (f"gemm_{tag}_lora1_rank8_tile4_edges_tt", lambda dt=dt: with_lora_source(128, lambda: gemm_lora_inlaunch_case(300, 640, 200, dt, ranks=(8,), tile=4))),
            (f"gemm_{tag}_lora1_tile2_tt", lambda dt=dt: with_lora_source(128, lambda: gemm_lora_inlaunch_case(520, 320, 384, dt, ranks=(16, 4), tile=2))),
            (f"gemm_{tag}_lora1_tile3_tt", lambda dt=dt: with_lora_source(128, lambda: gemm_lora_inlaunch_case(154, 2048, 640, dt, tile=3))),
            (f"gemm_{tag}_lora1_geglu_tt", lambda dt=dt: with_lora_source(128, lambda: gemm_lora_inlaunch_case(512, 640, 2560, dt, geglu=True))),
            (f"gemm_{tag}_lora1_transposed_tt", lambda dt=dt: with_lora_source(128, lambda: gemm_lora_inlaunch_case(154, 2048, 640, dt, transposed=True))),
            (f"gemm_{tag}_ln_lora_1024x1280_tt", lambda dt=dt: with_lora_source(128, lambda: gemm_ln_lora_case(1024, 1280, 1280, dt))),
            (f"gemm_{tag}_ln_lora_tile4_tt", lambda dt=dt: with_lora_source(128, lambda: gemm_ln_lora_case(300, 640, 384, dt, tile=4))),
            (f"gemm_{tag}_ln_lora_transposed_tile3_tt", lambda dt=dt: with_lora_source(128, lambda: gemm_ln_lora_case(512, 640, 256, dt, tile=3, transposed=True))),
            (f"gemm_{tag}_lora1_rank64_tile2_tt", lambda dt=dt: with_lora_source(128, lambda: gemm_lora_inlaunch_case(520, 320, 384, dt, ranks=(64,), tile=2))),
            (f"gemm_{tag}_lora1_rank128_tt", lambda dt=dt: with_lora_source(128, lambda: gemm_lora_inlaunch_case(2048, 1280, 1280, dt, ranks=(128,)))),
            (f"gemm_{tag}_lora1_ff2_2048x1280x5120_tt", lambda dt=dt: with_lora_source(128, lambda: gemm_lora_inlaunch_case(2048, 5120, 1280, dt))),
            (f"gemm_{tag}_lora1_repeat_shared_scratch_tt", lambda dt=dt: with_lora_source(128, lambda: gemm_lora_repeat_case(1024, 640, 1280, dt))),
            (f"gemm_{tag}_qkv_lora_tile1_shared64_prod", lambda dt=dt: with_lora_source(64, lambda: gemm_qkv_lora_case(1024, 1280, 1280, dt, tile=1))),
            (f"gemm_{tag}_qkv_lora_tile1_oddM_prod", lambda dt=dt: with_lora_source(64, lambda: gemm_qkv_lora_case(301, 640, 384, dt, tile=1))),
            (f"gemm_{tag}_qkv_lora_2048x1280_tile1", lambda dt=dt: gemm_qkv_lora_case(2048, 1280, 1280, dt, tile=1)),
            (f"gemm_{tag}_qkv_lora_8192x640_tile1", lambda dt=dt: gemm_qkv_lora_case(8192, 640, 640, dt, tile=1)),  # the 4096-token level's launch: 64 t-tiles
            (f"conv_{tag}_lora1_2x64x128_32x32", lambda dt=dt: conv_lora_inlaunch_case(2, 64, 128, 32, 32, dt)),
            (f"conv_{tag}_lora1_rank128_stride2", lambda dt=dt: conv_lora_inlaunch_case(2, 128, 256, 32, 32, dt, ranks=(128,), stride=2)),
            (f"conv_{tag}_lora1_shortcut_tile1", lambda dt=dt: conv_lora_inlaunch_case(1, 64, 128, 24, 40, dt, ranks=(8,), shortcut=64, tile=1)),
            (f"conv_{tag}_lora1_splitk3", lambda dt=dt: conv_lora_inlaunch_case(2, 320, 128, 16, 16, dt, ksplit=3, tile=1)),
            (f"wide_head_{tag}_384x512", lambda dt=dt: wide_head_attention_case(384, 512, dt)),
            (f"wide_head_{tag}_1024x512", lambda dt=dt: wide_head_attention_case(1024, 512, dt)),
            (f"softmax_rows_{tag}_vec", lambda dt=dt: softmax_rows_case(33, 1000, 1024, dt)),
            (f"softmax_rows_{tag}_unaligned", lambda dt=dt: softmax_rows_case(5, 77, 77, dt)),
            (f"softmax_rows_{tag}_long", lambda dt=dt: softmax_rows_case(3, 20000, 20032, dt)),
            (f"gemm_{tag}_ln_chain_transposed", lambda dt=dt: gemm_ln_chain_case(1024, 1280, 1280, dt, transposed=True)),
        ]
        # round 5: the 8-wave / eight-phase loop (csrc/gemm8_kernel.cuh) -- tile 7 = whole 256 x 256 tiles per workgroup (persistent when there are more
        # tiles than CUs), tile 8 = stream-K (partial tiles summed in a fixed order through the scratch of native.StreamK).  K tile counts around the
        # loop's peel points (1 .. 5, odd / even), ragged M / N, every epilogue, transposed column groups, K-blocked operands, several K segments, conv.
        kt = 128 // (4 if dt == torch.float32 else 2)  # elements per K tile
        for tile in (7, 8):
            for nkt in (1, 2, 3, 4, 5, 8):
                cases.append((f"gemm_{tag}_tile{tile}_{nkt}ktiles", lambda dt=dt, tile=tile, nkt=nkt, kt=kt: on_g8(lambda: gemm_tile_case(300, nkt * kt, 528, dt, tile, 0, seed=400 + nkt), 2)))  # (N % 16 == 0: the loop's entry condition)
            cases += [
                (f"gemm_{tag}_tile{tile}_2048x1280x1280_prefetch", lambda dt=dt, tile=tile: on_g8(lambda: gemm_tile_case(2048, 1280, 1280, dt, tile, 0, prefetch=True))),
                (f"gemm_{tag}_tile{tile}_many_tiles", lambda dt=dt, tile=tile: on_g8(lambda: gemm_tile_case(4352, 256, 4608, dt, tile, 0, seed=410))),  # 17 x 18 = 306 tiles: more than CUs
                (f"gemm_{tag}_tile{tile}_long_k_few_tiles", lambda dt=dt, tile=tile: on_g8(lambda: gemm_tile_case(512, 5120, 768, dt, tile, 0, seed=411))),
                (f"gemm_{tag}_tile{tile}_two_segments", lambda dt=dt, tile=tile: on_g8(lambda: gemm_multiseg_case(600, 640, 320 + 64, 528, dt, tile))),
                (f"gemm_{tag}_tile{tile}_geglu", lambda dt=dt, tile=tile: on_g8(lambda: gemm_geglu_case(1024, 640, 2560, dt, tile=tile))),
                (f"gemm_{tag}_tile{tile}_qkv", lambda dt=dt, tile=tile: on_g8(lambda: gemm_qkv_case(1000, 640, 640, dt, tile=tile, bias=True, pad=32, seed=431))),
                (f"gemm_{tag}_tile{tile}_qkv_2048x1280", lambda dt=dt, tile=tile: on_g8(lambda: gemm_qkv_case(2048, 1280, 1280, dt, tile=tile))),
                (f"gemm_{tag}_tile{tile}_t_only_ragged", lambda dt=dt, tile=tile: on_g8(lambda: gemm_t_only_case(77 * 2, 2048, 640, dt, tile=tile))),
                (f"gemm_{tag}_tile{tile}_ln_chain", lambda dt=dt, tile=tile: on_g8(lambda: gemm_ln_chain_case(1000, 640, 640, dt, tile1=tile, tile2=tile))),
                (f"gemm_{tag}_tile{tile}_ln_chain_geglu", lambda dt=dt, tile=tile: on_g8(lambda: gemm_ln_chain_case(512, 640, 5120, dt, geglu=True, tile1=1, tile2=tile))),
                (f"gemm_{tag}_tile{tile}_ln_chain_transposed", lambda dt=dt, tile=tile: on_g8(lambda: gemm_ln_chain_case(1024, 1280, 1280, dt, transposed=True, tile1=tile, tile2=tile))),
                (f"colstats_{tag}_tile{tile}", lambda dt=dt, tile=tile: on_g8(lambda: colstats_case(600, 384, 640, dt, tile=tile))),
                (f"colstats_{tag}_tile{tile}_edge_rows", lambda dt=dt, tile=tile: on_g8(lambda: colstats_case(96, 256, 320, dt, tile=tile))),
                (f"conv_{tag}_tile{tile}", lambda dt=dt, tile=tile: on_g8(lambda: conv_tile_case(2, 320, 384, 16, 24, dt, tile, 0))),
                (f"conv_{tag}_tile{tile}_odd_taps", lambda dt=dt, tile=tile: on_g8(lambda: conv_tile_case(1, 64, 320, 40, 24, dt, tile, 0, seed=221))),
                (f"conv_gn_{tag}_tile{tile}", lambda dt=dt, tile=tile: on_g8(lambda: conv_groupnorm_chain_case(2, 320, 320, 32, 32, dt, tile=tile))),
            ]
        # tile 9: the 8-wave loop on 192 x 256 tiles (wave tile 96 x 64; two of the eight waves stage no activation rows): K tile counts around the peel points, ragged
        # M (not a multiple of 192, of 96, of 32) and N, more tiles than CUs (persistent), every epilogue kind, K-blocked operands, two K segments, convolutions, LoRA
        for nkt in (1, 2, 3, 4, 5, 8):
            cases.append((f"gemm_{tag}_tile9_{nkt}ktiles", lambda dt=dt, nkt=nkt, kt=kt: on_tile9(lambda: gemm_tile_case(300, nkt * kt, 528, dt, 9, 0, seed=500 + nkt), 2)))
        cases += [
            (f"gemm_{tag}_tile9_2048x1280x1280_prefetch", lambda dt=dt: gemm_tile_case(2048, 1280, 1280, dt, 9, 0, prefetch=True)),
            (f"gemm_{tag}_tile9_8192x256x1280_one_round", lambda dt=dt: on_tile9(lambda: gemm_tile_case(8192, 256, 1280, dt, 9, 0, seed=509))),  # 43 x 5 = 215 tiles
            (f"gemm_{tag}_tile9_many_tiles", lambda dt=dt: gemm_tile_case(4352, 256, 4608, dt, 9, 0, seed=510)),  # 23 x 18 = 414 tiles: more than CUs
            (f"gemm_{tag}_tile9_ragged_rows", lambda dt=dt: gemm_tile_case(193 + 96 + 17, 320, 272, dt, 9, 0, seed=512)),
            (f"gemm_{tag}_tile9_long_k_few_tiles", lambda dt=dt: gemm_tile_case(512, 5120, 768, dt, 9, 0, seed=511)),
            (f"gemm_{tag}_tile9_two_segments", lambda dt=dt: on_tile9(lambda: gemm_multiseg_case(600, 640, 320 + 64, 528, dt, 9))),
            (f"gemm_{tag}_tile9_geglu", lambda dt=dt: on_tile9(lambda: gemm_geglu_case(1024, 640, 2560, dt, tile=9))),
            (f"gemm_{tag}_tile9_ln_chain", lambda dt=dt: gemm_ln_chain_case(1000, 640, 640, dt, tile1=9, tile2=9)),
            (f"gemm_{tag}_tile9_ln_chain_geglu", lambda dt=dt: gemm_ln_chain_case(512, 640, 5120, dt, geglu=True, tile1=1, tile2=9)),
            (f"gemm_{tag}_tile9_qkv_falls_back", lambda dt=dt: gemm_qkv_case(1000, 640, 640, dt, tile=9, bias=True, pad=32, seed=531)),  # (a transposed part: not on this tile)
            (f"colstats_{tag}_tile9", lambda dt=dt: on_tile9(lambda: colstats_case(600, 384, 640, dt, tile=9))),
            (f"colstats_{tag}_tile9_edge_rows", lambda dt=dt: colstats_case(96, 256, 320, dt, tile=9)),
            (f"conv_{tag}_tile9", lambda dt=dt: on_tile9(lambda: conv_tile_case(2, 320, 384, 16, 24, dt, 9, 0))),
            (f"conv_{tag}_tile9_odd_taps", lambda dt=dt: conv_tile_case(1, 64, 320, 40, 24, dt, 9, 0, seed=521)),
            (f"conv_gn_{tag}_tile9", lambda dt=dt: conv_groupnorm_chain_case(2, 320, 320, 32, 32, dt, tile=9)),
            (f"conv_{tag}_tile9_s2_ups_shortcut", lambda dt=dt: conv_forced_tile_case(dt, 9)),
            (f"gemm_{tag}_tile9_lora1_2048x1280x1280", lambda dt=dt: on_g8_lora(lambda: gemm_lora_inlaunch_case(2048, 1280, 1280, dt, tile=9), 2)),
        ]
        if dt == torch.bfloat16:  # tile id 10: the 8-wave loop on 128 x 256 tiles (bf16 GEMMs; float32 / convolutions asking for it run on the library's choice)
            cases += [(f"gemm_{tag}_tile10_{nkt}ktiles", lambda dt=dt, nkt=nkt: on_g8(lambda: gemm_tile_case(300, nkt * (128 // 2), 528, dt, 10, 0, seed=540 + nkt), 2)) for nkt in (1, 2, 3, 4, 5, 8)]
            cases += [
                (f"gemm_{tag}_tile10_one_round", lambda dt=dt: on_g8(lambda: gemm_tile_case(1024, 256, 8192, dt, 10, 0, seed=549))),  # 8 x 32 = 256 tiles
                (f"gemm_{tag}_tile10_many_tiles", lambda dt=dt: on_g8(lambda: gemm_tile_case(2048, 256, 10240, dt, 10, 0, seed=550))),  # 640 tiles: persistent, 2.5 rounds
                (f"gemm_{tag}_tile10_ragged_rows", lambda dt=dt: gemm_tile_case(129 + 64 + 17, 320, 272, dt, 10, 0, seed=552)),
                (f"gemm_{tag}_tile10_two_segments", lambda dt=dt: on_g8(lambda: gemm_multiseg_case(600, 640, 320 + 64, 528, dt, 10))),
                (f"gemm_{tag}_tile10_geglu", lambda dt=dt: on_g8(lambda: gemm_geglu_case(1024, 640, 2560, dt, tile=10))),
                (f"gemm_{tag}_tile10_ln_chain_geglu", lambda dt=dt: gemm_ln_chain_case(512, 640, 5120, dt, geglu=True, tile1=1, tile2=10)),
                (f"colstats_{tag}_tile10", lambda dt=dt: on_g8(lambda: colstats_case(600, 384, 640, dt, tile=10))),
                (f"gemm_{tag}_tile10_lora1_2048x1280x1280", lambda dt=dt: on_g8_lora(lambda: gemm_lora_inlaunch_case(2048, 1280, 1280, dt, tile=10), 2)),
                (f"gemm_{tag}_tile10_lora1_rank8_edges", lambda dt=dt: on_g8_lora(lambda: gemm_lora_inlaunch_case(300, 640, 208, dt, ranks=(8,), tile=10), 2)),
                # tile id 11: 192-row tiles for a whole number of rounds + 128-row tiles for a whole number of rounds in ONE launch (csrc/gemm8_kernel.cuh plan_mix / mix_coords):
                # FF1 of a CFG pair = 256 + 256 tiles; 4096 x 10240 = 512 + 512 (another row / column split); every epilogue FF1 uses; shapes without such a split run as tile 9
                (f"gemm_{tag}_tile11_ff1_plain", lambda dt=dt: on_tile11(lambda: gemm_tile_case(2048, 256, 10240, dt, 11, 0, seed=560))),
                (f"gemm_{tag}_tile11_ff1_geglu", lambda dt=dt: on_tile11(lambda: gemm_geglu_case(2048, 320, 5120, dt, seed=561, tile=11))),
                (f"gemm_{tag}_tile11_ff1_ln_chain_geglu", lambda dt=dt: on_tile11(lambda: gemm_ln_chain_case(2048, 640, 10240, dt, geglu=True, tile1=1, tile2=11, seed=562))),
                (f"gemm_{tag}_tile11_ff1_lora1_geglu", lambda dt=dt: on_tile11(lambda: gemm_lora_inlaunch_case(2048, 640, 10240, dt, tile=11, geglu=True, seed=563))),
                (f"gemm_{tag}_tile11_4096x10240", lambda dt=dt: on_tile11(lambda: gemm_tile_case(4096, 192, 10240, dt, 11, 0, seed=564))),
                (f"gemm_{tag}_tile11_no_split_runs_as_tile9", lambda dt=dt: on_tile9(lambda: gemm_tile_case(1000, 256, 1280, dt, 11, 0, seed=565))),
            ]
        cases += [
            (f"gemm_{tag}_tile9_lora1_rank8_edges", lambda dt=dt: on_g8_lora(lambda: gemm_lora_inlaunch_case(300, 640, 208, dt, ranks=(8,), tile=9), 2)),
            (f"gemm_{tag}_tile9_lora1_rank128", lambda dt=dt: on_g8_lora(lambda: gemm_lora_inlaunch_case(2048, 1280, 1280, dt, ranks=(128,), tile=9), 2)),
            (f"gemm_{tag}_tile9_lora1_geglu", lambda dt=dt: on_tile9(lambda: on_g8_lora(lambda: gemm_lora_inlaunch_case(512, 640, 2560, dt, geglu=True, tile=9)))),
            (f"gemm_{tag}_tile9_ln_lora_edges", lambda dt=dt: on_g8_lora(lambda: gemm_ln_lora_case(300, 640, 384, dt, tile=9))),
            (f"gemm_{tag}_tile9_lora1_repeat_shared_scratch", lambda dt=dt: on_g8_lora(lambda: gemm_lora_repeat_case(1024, 640, 1280, dt, tiles=(9, 1, 7, 9)), 9)),
        ]
        # the in-launch LoRA on the 8-wave loop (one column group of a plain GEMM; everything else keeps the 4-wave kernel's producers)
        cases += [
            (f"gemm_{tag}_tile7_lora1_2048x1280x1280", lambda dt=dt: on_g8_lora(lambda: gemm_lora_inlaunch_case(2048, 1280, 1280, dt, tile=7), 2)),
            (f"gemm_{tag}_tile7_lora1_rank8_edges", lambda dt=dt: on_g8_lora(lambda: gemm_lora_inlaunch_case(300, 640, 208, dt, ranks=(8,), tile=7), 2)),
            (f"gemm_{tag}_tile7_lora1_rank64", lambda dt=dt: on_g8_lora(lambda: gemm_lora_inlaunch_case(520, 640, 384, dt, ranks=(32, 16, 8), tile=7), 2)),
            (f"gemm_{tag}_tile7_lora1_rank128", lambda dt=dt: on_g8_lora(lambda: gemm_lora_inlaunch_case(2048, 1280, 1280, dt, ranks=(128,), tile=7), 2)),
            (f"gemm_{tag}_tile7_lora1_geglu", lambda dt=dt: on_g8_lora(lambda: gemm_lora_inlaunch_case(512, 640, 2560, dt, geglu=True, tile=7))),
            (f"gemm_{tag}_tile7_lora1_ff1_geglu_more_tiles_than_cus", lambda dt=dt: on_g8_lora(lambda: gemm_lora_inlaunch_case(2048, 1280, 10240, dt, geglu=True, tile=7))),
            (f"gemm_{tag}_tile7_lora1_ff2_2048x1280x5120", lambda dt=dt: on_g8_lora(lambda: gemm_lora_inlaunch_case(2048, 5120, 1280, dt, tile=7), 2)),
            (f"gemm_{tag}_tile7_lora1_short_k", lambda dt=dt: on_g8_lora(lambda: gemm_lora_inlaunch_case(1000, 64, 528, dt, tile=7), 2)),
            (f"gemm_{tag}_tile7_ln_lora_1024x1280", lambda dt=dt: on_g8_lora(lambda: gemm_ln_lora_case(1024, 1280, 1280, dt, tile=7))),
            (f"gemm_{tag}_tile7_ln_lora_edges", lambda dt=dt: on_g8_lora(lambda: gemm_ln_lora_case(300, 640, 384, dt, tile=7))),
            (f"gemm_{tag}_tile7_lora1_repeat_shared_scratch", lambda dt=dt: on_g8_lora(lambda: gemm_lora_repeat_case(1024, 640, 1280, dt, tiles=(7, 1, 7, 3)), 6)),
            (f"gemm_{tag}_tile7_lora1_transposed_keeps_4wave", lambda dt=dt: gemm_lora_inlaunch_case(154, 2048, 640, dt, transposed=True, tile=7)),
            (f"gemm_{tag}_tile7_qkv_lora_keeps_4wave", lambda dt=dt: gemm_qkv_lora_case(512, 640, 640, dt, tile=7)),
        ]
        cases += [
            (f"conv_{tag}_tile7_s2_ups_shortcut", lambda dt=dt: conv_forced_tile_case(dt, 7)),
            (f"conv_{tag}_tile8_s2_ups_shortcut", lambda dt=dt: conv_forced_tile_case(dt, 8)),
        ]
    return cases
