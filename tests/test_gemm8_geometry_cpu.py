"""Index arithmetic of the 8-wave GEMM loop (refiners_amd/csrc/gemm8_kernel.cuh), restated in Python and checked exhaustively on the CPU for both tile heights
(MT = 8: 256 x 256, MT = 6: 192 x 256).  The kernel is checked against references on the GPU (tests/test_kernels_gpu.py); this guards the constants that tie its
three views of LDS together -- where the loader's lanes put a row, where the fragment reads expect it, which output rows a wave's accumulators are -- so that an edit
to one of them fails here, without a GPU, and names the broken relation."""
import pytest

NTHR, BN = 512, 256


def geometry(MT):
    return dict(MT=MT, BM=32 * MT, WR=16 * MT, QR=8 * MT, XW=MT)  # XW = QR / 8: the waves that stage rows of an X half tile


def loader_lanes():
    for wid in range(8):
        for lane in range(64):
            yield wid, lane, lane >> 3, lane & 7  # lr8 = row within the wave's 8 rows, physical 16-byte chunk (the destination is lane-linear)


@pytest.mark.parametrize("MT", [8, 6])
def test_x_slot_every_row_is_staged_once_and_the_swizzle_matches_the_reads(MT):
    g = geometry(MT)
    written = {}
    for h in range(2):
        for s in range(2):
            for wid, lane, lr8, pc in loader_lanes():
                real = MT == 8 or wid < g["XW"]
                if not real:
                    continue  # (these waves' loads carry the out-of-range marker and land in the spare area: nothing of the slot is touched)
                R = g["WR"] * s + g["QR"] * h + 8 * wid + lr8          # LDS row: stage_x
                src_row = g["WR"] * s + g["QR"] * h + (8 * wid + lr8)  # tile row: xrow (xs_0 + ... + xlane)
                assert R == src_row < g["BM"]
                coff_chunk = pc ^ (4 * (wid & 1) + (lane >> 4))        # source chunk this lane fetches (coff)
                assert ((R >> 1) & 7) == 4 * (wid & 1) + (lane >> 4), "the loader's one-offset-per-thread swizzle needs (R >> 1) & 7 == 4 (wid & 1) + (lane >> 4)"
                key = (R, pc)
                assert key not in written, key
                written[key] = coff_chunk  # physical chunk pc of row R holds logical chunk pc ^ swizzle(R)
    assert len(written) == g["BM"] * 8
    # fragment reads (read_x): wave row wm, half h, 16-row block i, lane (c16, gq), K half kk -> logical chunk 4 kk + gq of row WR wm + QR h + 16 i + c16
    seen_rows = set()
    for wm in range(2):
        for h in range(2):
            for i in range(MT // 2):
                for c16 in range(16):
                    row = g["WR"] * wm + g["QR"] * h + 16 * i + c16
                    seen_rows.add(row)
                    for kk in range(2):
                        for gq in range(4):
                            phys = (4 * kk + gq) ^ ((c16 >> 1) & 7)  # fo[kk]
                            assert written[(row, phys)] == 4 * kk + gq, (row, kk, gq)
    assert seen_rows == set(range(g["BM"]))


def test_w_slot_rows_are_permuted_so_that_a_lane_owns_sixteen_consecutive_columns():
    holds = {}
    for h in range(2):
        for s in range(2):
            for wid, lane, lr8, pc in loader_lanes():
                R = 64 * (2 * s + (wid >> 2)) + 32 * h + 8 * (wid & 3) + lr8  # stage_w
                wq = 8 * (wid & 3) + lr8
                wlane = 64 * (wid >> 2) + 16 * ((wq >> 2) & 3) + 4 * (wq >> 4) + (wq & 3)
                src = 128 * s + 8 * h + wlane                                   # wrow
                assert ((R >> 1) & 7) == 4 * (wid & 1) + (lane >> 4)
                assert holds.setdefault(R, src) == src
    assert sorted(holds) == list(range(BN)) and sorted(holds.values()) == list(range(BN))
    for R, src in holds.items():
        rl = R % 64
        j, a, b = rl >> 4, (rl >> 2) & 3, rl & 3
        assert src == (R - rl) + 16 * a + 4 * j + b  # gemm_kernel.cuh header: MMA row 4 a + b of block j is column 16 a + 4 j + b of the wave's 64
    # the accumulator view: acc[i][j][r] of lane (ge, ce) in wave column wn is column 64 wn + 16 ge + 4 j + r: sixteen consecutive columns per lane
    for wn in range(4):
        for ge in range(4):
            cols = []
            for j in range(4):
                for r in range(4):
                    R = 64 * wn + 16 * j + 4 * ge + r  # W-slot LDS row feeding MMA block j, MMA row 4 ge + r
                    cols.append(holds[R])
            assert sorted(cols) == list(range(64 * wn + 16 * ge, 64 * wn + 16 * ge + 16))


@pytest.mark.parametrize("MT", [8, 6])
def test_lora_flag_blocks_of_a_tile_and_of_a_wave(MT):
    g = geometry(MT)
    assert g["BM"] % 32 == 0 and g["WR"] % 32 == 0
    for m0 in (0, g["BM"], 7 * g["BM"]):
        blocks = {m0 // 32 + t for t in range(g["BM"] // 32)}  # the 32-row blocks of the tile: one producer (one flag) each
        polled = set()
        for wm in range(2):
            for lane in range(64):
                polled.add((m0 + g["WR"] * wm) // 32 + min(lane & 3, g["WR"] // 32 - 1))  # the consumer's poll
        assert polled == blocks
        rows = {m0 + g["WR"] * wm + 16 * i + ce for wm in range(2) for i in range(MT) for ce in range(16)}
        assert {r // 32 for r in rows} == blocks


@pytest.mark.parametrize("MT", [8, 6])
@pytest.mark.parametrize("R", [32, 64, 128])
@pytest.mark.parametrize("M", [2048, 8192, 300, 33])
def test_lora_producers_of_the_8_wave_loop_cover_every_row_block_once(MT, R, M):
    """Round 6: t comes from producer workgroups (gemm_lora_producer.cuh) instead of t-tiles.  Workgroup `bid` of the head of the grid runs, for ranks 32 / 64,
    TWO producers (waves 0-3 / 4-7: row blocks 2 bid, 2 bid + 1, half of the stage buffers each), for rank 128 one; the launcher pads the workgroup count
    to a multiple of 8.  Every 32-row block must be produced exactly once, every ring must fit its share of the LDS with at least two stages."""
    npb = (M + 31) // 32
    halves = 2 if R <= 64 else 1
    lp_blocks = ((npb + (1 if halves == 2 else 0)) // halves + 7) // 8 * 8
    made = []
    for bid in range(lp_blocks):
        for half in range(2):
            q = halves * bid + half
            if half >= halves or q >= npb:
                continue
            made.append(q)
    assert sorted(made) == list(range(npb))
    ring = 2 * (32 * MT + 256) * 128
    share = ring // halves
    stage = (32 + R) * 128  # 32 rows of x + R stacked down rows, 128 bytes of K each
    pst = min(8, share // stage)
    assert pst >= 2, (MT, R, pst)
    assert pst * stage * halves <= ring


@pytest.mark.parametrize("MT", [8, 6])
def test_lds_budget(MT):
    lds = 2 * (32 * MT + 256) * 128 + 2 * (256 * 8 + 2 * 256 * 4) + 2048
    assert lds <= 160 * 1024
    assert lds == {8: 141312, 6: 124928}[MT]
