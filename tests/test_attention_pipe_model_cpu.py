"""CPU model of the software-pipelined self-attention loop (csrc/attention.hip, attn_pipe_kernel) -- what can be checked without a GPU:

  schedule    K runs one tile ahead of V through two K buffers and two V buffers; iteration i reads K(i + 1) and V(i) and, at its START, has the
              LDS-DMA of K(min(i + 2, last)) and V(i + 1) written into the buffers the previous iteration left.  No buffer may be written in the
              iteration that reads it, and every read must see the tile it expects -- for every tile count;
  arithmetic  the lazy running maximum (reference moves only when some score of the tile exceeds it by more than thr = 8 / c, P = 2^(c (s - ref))),
              the row sums taken over the ROUNDED P (what the ones-fragment MFMA accumulates), the folded form (accumulators start at -ref in
              scaled units) -- against softmax(Q K^T scale) V in float64 (ScaledDotProductAttention, fluxion/layers/attentions.py:60-202);
  index maps  the loader's per-lane offsets reproduce the LDS image the fragment reads expect: S^T block t, lane (g, c16), element r is the score of
              key k_row_key(16 t + 4 g + r) (the tail mask's formula) and P^T fragment s2 of a lane covers eight consecutive keys = one 16-byte
              chunk of a V^T row.
The kernel itself is checked on the GPU (tests/kernel_cases.py, the attn_*_r6_* cases)."""
import numpy as np
import pytest

BKV = 64


# ------------------------------------------------------------------------------------------------------------------------------ schedule
@pytest.mark.parametrize("ntile", list(range(1, 12)))
def test_buffers_are_never_written_in_the_iteration_that_reads_them(ntile):
    kbuf, vbuf = [None, None], [None, None]  # which tile a buffer holds
    # prologue: K(0) -> K0, V(0) -> V0, K(min(1, last)) -> K1; wait; barrier; Q K^T of tile 0 from K0
    kbuf[0], vbuf[0], kbuf[1] = 0, 0, min(1, ntile - 1)
    assert kbuf[0] == 0
    scores_of = 0  # the tile whose scores the wave holds
    pv_done = []
    for i in range(ntile):
        nxt = i + 1 < ntile
        writes_k = writes_v = None
        if nxt:  # DMA issued at the start of the iteration, waited for at its end
            writes_k, writes_v = i & 1, (i + 1) & 1
            tile_k, tile_v = min(i + 2, ntile - 1), i + 1
        # phase 1: Q K^T of tile i + 1
        if nxt:
            rb = (i + 1) & 1
            assert rb != writes_k, "K buffer read while its DMA is in flight"
            assert kbuf[rb] == i + 1, f"iteration {i}: K buffer {rb} holds tile {kbuf[rb]}"
        # phases 2..: P V of tile i
        assert scores_of == i
        rv = i & 1
        if nxt:
            assert rv != writes_v, "V buffer read while its DMA is in flight"
        assert vbuf[rv] == i, f"iteration {i}: V buffer {rv} holds tile {vbuf[rv]}"
        pv_done.append(i)
        if nxt:  # vmcnt(0) + barrier: the DMA has landed for every wave
            kbuf[writes_k], vbuf[writes_v] = tile_k, tile_v
            scores_of = i + 1
    assert pv_done == list(range(ntile))


@pytest.mark.parametrize("ntile", [1, 2, 3, 6])
def test_no_interval_between_two_barriers_reads_and_writes_one_buffer(ntile):
    """The kernel's LDS traffic as intervals between barriers: waves of a workgroup drift inside an interval, so a buffer that one wave still reads
    must not be (DMA-)written by another in the same interval.  The prologue's Q K^T of tile 0 reads K buffer 0, iteration 0 re-fills it: they are
    separate intervals (the barrier after the prologue's fragment reads)."""
    intervals = [{"w": {("K", 0), ("V", 0), ("K", 1)}, "r": set()},            # prologue DMA; vmcnt(0); barrier
                 {"w": set(), "r": {("K", 0)}}]                                 # Q K^T of tile 0; barrier
    for i in range(ntile):
        nxt = i + 1 < ntile
        it = {"w": set(), "r": {("V", i & 1)}}
        if nxt:
            it["w"] |= {("K", i & 1), ("V", (i + 1) & 1)}
            it["r"] |= {("K", (i + 1) & 1)}
        intervals.append(it)                                                    # (NEXT: vmcnt(0); barrier)
    for n, it in enumerate(intervals):
        assert not (it["w"] & it["r"]), f"interval {n} reads and writes {it['w'] & it['r']}"


# ---------------------------------------------------------------------------------------------------------------------------- arithmetic
def bf16_round(x):
    """Round-to-nearest-even to bfloat16, returned as float32 values."""
    x = np.asarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def lazy_attention(q, k, v, scale, Lk, fold=False, stats=None):
    """One head: q [Lq, 64], k / v [Lkp, 64] (bf16 values held in float32).  Mirrors attn_pipe_kernel per query: tiles of 64 keys, keys >= Lk masked,
    P rounded to bf16, l = sum of the rounded P, O += P V in float32."""
    c = np.float32(scale * 1.4426950408889634)
    thr = np.float32(8.0) / c
    Lq = q.shape[0]
    ntile = (Lk + BKV - 1) // BKV
    o = np.zeros((Lq, 64), np.float32)
    l = np.zeros(Lq, np.float32)
    ref = np.zeros(Lq, np.float32) if fold else np.full(Lq, -np.inf, np.float32)  # fold: scaled units
    qq = bf16_round(q * c) if fold else q
    moves = 0
    for t in range(ntile):
        ks = k[t * BKV:(t + 1) * BKV]
        s = (qq.astype(np.float32) @ ks.T.astype(np.float32)).astype(np.float32)  # raw scores (fold: already scaled)
        if fold:
            s = s - ref[:, None]  # the accumulators start at -ref
        valid = (t * BKV + np.arange(BKV)) < Lk
        s[:, ~valid] = -np.inf
        mx = s.max(axis=1)
        # the kernel decides per WAVE (16 / 32 queries); per query is the same arithmetic: a query whose maximum did not grow gets alpha = 1 exactly
        if fold:
            need = (mx > 8.0) | (t == 0)
            d = np.where(t == 0, mx, np.maximum(mx, 0.0)).astype(np.float32)
            d = np.where(need, d, 0.0).astype(np.float32)
            alpha = np.exp2(-d).astype(np.float32)
            ref = ref + d
            s = s - d[:, None]
            p = np.exp2(s).astype(np.float32)
        else:
            need = mx > ref + thr
            mnew = np.where(need, np.maximum(ref, mx), ref).astype(np.float32)
            with np.errstate(invalid="ignore"):
                alpha = np.where(need, np.exp2((ref - mnew) * c), 1.0).astype(np.float32)
            alpha = np.nan_to_num(alpha, nan=0.0)
            ref = mnew
            p = np.exp2(s * c - (ref * c)[:, None]).astype(np.float32)
        moves += int(need.sum())
        assert p.max() <= 2.0 ** 8 * 1.01, "P exceeded 2^8"
        pb = bf16_round(p)
        o = o * alpha[:, None] + pb @ v[t * BKV:(t + 1) * BKV].astype(np.float32)
        l = l * alpha + pb.sum(axis=1)
    if stats is not None:
        stats["moves"] = moves
    assert (l > 0).all()
    return o / l[:, None]


def reference_attention(q, k, v, scale, Lk):
    s = (q.astype(np.float64) @ k[:Lk].T.astype(np.float64)) * scale
    s -= s.max(axis=1, keepdims=True)
    p = np.exp(s)
    return (p / p.sum(axis=1, keepdims=True)) @ v[:Lk].astype(np.float64)


def operands(Lq, Lk, seed, kind):
    rng = np.random.default_rng(seed)
    Lkp = (Lk + 63) // 64 * 64
    q = bf16_round(rng.standard_normal((Lq, 64)))
    k = bf16_round(rng.standard_normal((Lkp, 64)))
    v = bf16_round(rng.standard_normal((Lkp, 64)))
    if kind == "growing":  # every tile moves the reference of most queries
        k = bf16_round(k * (1.0 + 3.0 * np.arange(Lkp)[:, None] / Lkp) + 0.5)
        q = bf16_round(q + 0.5)
    elif kind == "spike":  # late large scores: the rescale path after many quiet tiles
        k[Lk - 3] *= 6.0
        k[Lk // 2] *= 4.0
        k = bf16_round(k)
    elif kind == "negative":  # all scores far below zero: the first tile must still set the reference
        k = bf16_round(-np.abs(k) * 4.0)
        q = bf16_round(np.abs(q) * 4.0)
    return q, k, v


@pytest.mark.parametrize("kind", ["normal", "growing", "spike", "negative"])
@pytest.mark.parametrize("Lq,Lk", [(48, 1024), (32, 257), (16, 20), (16, 64), (40, 130)])
@pytest.mark.parametrize("fold", [False, True])
def test_lazy_reference_matches_softmax(kind, Lq, Lk, fold):
    q, k, v = operands(Lq, Lk, 7 + Lq + Lk, kind)
    st = {}
    out = lazy_attention(q, k, v, 0.125, Lk, fold=fold, stats=st)
    ref = reference_attention(q, k, v, 0.125, Lk)
    err = np.abs(out - ref).max() / np.abs(ref).max()
    # bf16 P: 2^-9 relative per term; the folded form rounds c Q to bf16 once more (a score error that grows with the scores' size)
    bound = 6e-3 if not fold else (4e-2 if kind in ("growing", "negative", "spike") else 8e-3)
    assert err < bound, (err, st)
    if kind == "normal" and Lk >= 1024 and not fold:
        assert st["moves"] < 3 * Lq, f"the reference of normal data should settle after the first tiles, moved {st['moves']} times"


def test_lazy_is_exact_when_the_reference_never_moves():
    """With every later score below reference + thr the lazy loop IS the plain loop with a fixed shift: same P, same l, same O as an online softmax
    that takes tile 0's maximum as the shift."""
    q, k, v = operands(16, 512, 3, "normal")
    k[64:] *= 0.05  # later tiles far below tile 0's maxima
    k = bf16_round(k)
    st = {}
    out = lazy_attention(q, k, v, 0.125, 512, stats=st)
    assert st["moves"] == 16  # tile 0 only
    c = np.float32(0.125 * 1.4426950408889634)
    s = (q @ k.T).astype(np.float32)
    m0 = s[:, :64].max(axis=1)
    pb = bf16_round(np.exp2(s * c - (m0 * c)[:, None]).astype(np.float32))
    # tile-wise float32 accumulation in the kernel's order
    o = np.zeros((16, 64), np.float32)
    l = np.zeros(16, np.float32)
    for t in range(8):
        o = o + pb[:, t * 64:(t + 1) * 64] @ v[t * 64:(t + 1) * 64]
        l = l + pb[:, t * 64:(t + 1) * 64].sum(axis=1)
    np.testing.assert_array_equal(out, o / l[:, None])


# ---------------------------------------------------------------------------------------------------------------------------- index maps
def k_row_key(row):
    return (row & 32) + 8 * ((row >> 2) & 3) + 4 * ((row >> 4) & 1) + (row & 3)


def swz128(row):
    return (row >> 1) & 7


def test_loader_image_and_fragment_reads_agree_with_the_mask_and_the_p_fragment():
    rng = np.random.default_rng(11)
    ROWB, CPR, NTHR, LI = 128, 8, 256, 2
    K = rng.integers(0, 1 << 15, size=(64, 64))          # [key][d], distinct values
    VT = rng.integers(0, 1 << 15, size=(64, 64))         # [d][key]
    # loader: thread tid, iteration it writes LDS chunk (it * NTHR + tid) (lane-linear DMA) from global (row, chunk pch ^ swz(row))
    kl = np.empty((64, 64), dtype=K.dtype)                # LDS image as [row][physical element]
    vl = np.empty((64, 64), dtype=K.dtype)
    for it in range(LI):
        for tid in range(NTHR):
            qi = it * NTHR + tid
            row, pch = qi // CPR, qi % CPR
            src = pch ^ swz128(row)
            kl[row, pch * 8:(pch + 1) * 8] = K[k_row_key(row), src * 8:(src + 1) * 8]
            j, a, bb = row >> 4, (row >> 2) & 3, row & 3
            vl[row, pch * 8:(pch + 1) * 8] = VT[16 * a + 4 * j + bb, src * 8:(src + 1) * 8]

    def frag(lds, row, chunk):  # tile_off<128>(row, chunk): logical chunk -> physical chunk ^ swz(row)
        pc = chunk ^ swz128(row)
        return lds[row, pc * 8:(pc + 1) * 8]

    # S^T block t: A operand = LDS rows 16 t + c16 over chunks 4 s + g; lane (g, c16) receives rows 4 g + r of the block
    for t in range(4):
        for c16 in range(16):
            for s in range(2):
                for g in range(4):
                    row = 16 * t + c16
                    np.testing.assert_array_equal(frag(kl, row, 4 * s + g), K[k_row_key(row), (4 * s + g) * 8:(4 * s + g + 1) * 8])
    # the P^T fragment of a lane in group g for the 32-key half s2: elements [st[2 s2][r] (r = 0..3), st[2 s2 + 1][r]] = keys of LDS rows
    # 16 (2 s2) + 4 g + r and 16 (2 s2 + 1) + 4 g + r -- eight CONSECUTIVE keys 32 s2 + 8 g .. + 7, i.e. chunk 4 s2 + g of a V^T row
    for s2 in range(2):
        for g in range(4):
            keys = [k_row_key(16 * (2 * s2) + 4 * g + r) for r in range(4)] + [k_row_key(16 * (2 * s2 + 1) + 4 * g + r) for r in range(4)]
            assert keys == list(range(32 * s2 + 8 * g, 32 * s2 + 8 * g + 8))
    # V^T fragments: block i, lane c16 reads LDS row 16 i + c16, chunk 4 s2 + g; the row holds head-dim 16 a + 4 j + b of row = 16 j + 4 a + b,
    # so block i gives the lane (g', c16') = MFMA rows 4 g' + r' -> d = 16 g' + 4 i + r': 16 consecutive outputs per lane over i, r'
    for i in range(4):
        for c16 in range(16):
            row = 16 * i + c16
            j, a, bb = row >> 4, (row >> 2) & 3, row & 3
            d = 16 * a + 4 * j + bb
            for ch in range(8):
                np.testing.assert_array_equal(frag(vl, row, ch), VT[d, ch * 8:(ch + 1) * 8])
            gq, rq = c16 >> 2, c16 & 3  # MFMA output row c16 = 4 g' + r' lands in lane group g', element r'
            assert d == 16 * gq + 4 * i + rq


@pytest.mark.parametrize("Lk", [1, 20, 63, 64, 65, 77, 130, 1000])
def test_last_tile_offsets_stay_inside_the_tensor(Lk):
    """koff_last clamps the key of every loader row of the LAST tile to Lk - 1; the descriptor's range ends with the last valid row."""
    ntile = (Lk + 63) // 64
    ldkb = 2560
    last = Lk - 1 - (ntile - 1) * 64
    hi = 0
    for row in range(64):
        key = k_row_key(row)
        key = key if key < last else last
        off = (ntile - 1) * 64 * ldkb + key * ldkb + 7 * 16
        hi = max(hi, off + 16)
    assert hi <= (Lk - 1) * ldkb + 128
