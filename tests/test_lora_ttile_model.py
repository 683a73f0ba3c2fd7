"""Index algebra of the t-tiles (csrc/gemm_kernel.cuh, GemmP::lora_tt), CPU only.  A t-tile is an ordinary GEMM tile whose "weight rows" are the
stacked LoRA down rows of all column groups, addressed as a virtual column tile; three mappings have to agree for its epilogue to publish
t[group][m][rank] = x[m] . A_group[rank] (LoraAdapter's down projection, fluxion/adapters/lora.py:14-60, 383-397):
  loader    LDS weight row `row` of the tile holds virtual column v = perm(row); v -> (group v // R, rank v % R), columns past groups x R re-read rank 0
  MFMA      block j of wave column wn multiplies LDS rows wn WNE + 16 j + (0..15); lane (g, c16) ends up with rows 4 g + r of that block in acc[.][j][r]
  epilogue  lane (g, c16) of wave column wn owns the RUN = 4 NT consecutive virtual columns wn WNE + RUN g + (4 j + r)
This restates the three in numpy and checks the composition for every tile width / stacked rank / group count the library routes to t-tiles.  The
kernel itself is checked on the GPU (tests/kernel_cases.py, the *_tt cases and the Q|K|V^T ones)."""
import numpy as np
import pytest


def perm(row: int, NT: int) -> int:
    """Virtual column held by LDS weight row `row` (the loader's `(row - rl) + 4 NT a + 4 j + b`): inside each wave's span of WNE = 16 NT rows the
    16-row MFMA block j, row 4 a + b holds column 4 NT a + 4 j + b, so that after the MFMA a lane owns 4 NT CONSECUTIVE columns."""
    WNE = 16 * NT
    rl = row % WNE
    j, a, b = rl >> 4, (rl >> 2) & 3, rl & 3
    return (row - rl) + 4 * NT * a + 4 * j + b


@pytest.mark.parametrize("BN,R,groups", [(128, 32, 3), (128, 32, 1), (64, 32, 1), (128, 64, 2), (128, 128, 1), (64, 64, 1), (128, 32, 2)])
def test_t_tile_publishes_every_rank_of_every_group_exactly_once(BN, R, groups):
    rng = np.random.default_rng(BN + R + groups)
    WN, K, rows = 2, 64, 16                      # one 16-row MMA block of x is enough: rows are independent
    NT = BN // WN // 16
    WNE, RUN = 16 * NT, 4 * NT
    A = rng.standard_normal((groups, R, K))      # the groups' stacked down rows
    x = rng.standard_normal((rows, K))
    # loader: the LDS image of the virtual weight tile
    lds = np.empty((BN, K))
    for row in range(BN):
        v = perm(row, NT)
        v = v if v < groups * R else 0           # (the kernel's clamp: a valid row, its product is dropped)
        lds[row] = A[v // R, v % R]
    assert sorted(perm(r, NT) for r in range(BN)) == list(range(BN))  # a permutation of the tile's columns
    # MFMA + epilogue
    t = np.full((groups, rows, R), np.nan)
    writes = np.zeros((groups, rows, R), dtype=int)
    for wn in range(WN):
        for g in range(4):
            nv = wn * WNE + RUN * g              # the lane group's first virtual column
            gi, r0 = nv // R, nv % R
            assert (nv + RUN - 1) // R == gi     # RUN consecutive ranks of ONE group (R >= 32 >= RUN)
            for j in range(NT):
                for r in range(4):
                    acc = x @ lds[wn * WNE + 16 * j + 4 * g + r]  # D[weight row 4 g + r of block j][x row c16], all c16 at once
                    if gi < groups:
                        t[gi, :, r0 + 4 * j + r] = acc
                        writes[gi, :, r0 + 4 * j + r] += 1
    assert (writes == 1).all()
    np.testing.assert_allclose(t, np.einsum("mk,grk->gmr", x, A), rtol=1e-12, atol=1e-12)


def test_flags_of_a_t_tile_cover_its_row_blocks_for_every_group():
    """One flag per (group, 32 rows); a t-tile of BM rows at row tile tm publishes BM / 32 of them per group, threads 0 .. groups BM / 32 - 1."""
    for BM, M, groups in ((128, 2048, 3), (128, 301, 3), (64, 154, 1), (128, 8192, 3)):
        nfl, FB = (M + 31) // 32, BM // 32
        seen = np.zeros((groups, nfl), dtype=int)
        for tm in range((M + BM - 1) // BM):
            for tid in range(groups * FB):
                fg, fb = tid // FB, tm * BM // 32 + tid % FB
                if fb < nfl:
                    seen[fg, fb] += 1
        assert (seen == 1).all()
