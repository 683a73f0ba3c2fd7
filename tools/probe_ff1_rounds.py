"""Pricing of "FF1 as two full rounds of two tile heights"  with what exists: GEGLU launches that are EXACTLY one dispatch round of the 8-wave loop
-- 256 tiles of 192 x 256 (1536 x 8192 x 1280) and 256 tiles of 256 x 256 (2048 x 8192) -- and exactly two (512 tiles: 3072 x 8192 / 4096 x 8192), hot, against the CFG pair's FF1
(2048 x 10240: 440 tiles of 192 rows = 1.72 rounds).  t(128-row round) is read off as t(256-row round) / 2 + the fixed part (fill + epilogue) the one- / two-round pairs give."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402

dev, dt = "cuda", torch.bfloat16


def run(M, N, K, tile, n=5):
    sets = []
    for _ in range(6):
        x = torch.randn(M, K, device=dev).to(dt)
        w = native.KBlocked((torch.randn(N, K, device=dev) * K ** -0.5).to(dt))
        o = torch.empty(M, N // 2, device=dev, dtype=dt)
        sets.append((x, w, o))
    best = 1e9
    for _ in range(3):
        for x, w, o in sets:
            native.gemm([(x, w)], o, geglu=True, tile=tile)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            for x, w, o in sets:
                native.gemm([(x, w)], o, geglu=True, tile=tile)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / (n * len(sets)) * 1e3)
    return best


def main():
    native.load()
    K = 1280
    rows = [("FF1 as it runs: 2048 x 10240, tile 9 (440 tiles of 192 rows)", 2048, 10240, 9), ("FF1 on tile 7 (320 tiles of 256 rows)", 2048, 10240, 7),
            ("one round of 192-row tiles: 1536 x 8192", 1536, 8192, 9), ("two rounds of 192-row tiles: 3072 x 8192", 3072, 8192, 9),
            ("one round of 256-row tiles: 2048 x 8192", 2048, 8192, 7), ("two rounds of 256-row tiles: 4096 x 8192", 4096, 8192, 7),
            ("one round of 128-row tiles (tile 10): 1024 x 8192", 1024, 8192, 10), ("two rounds of 128-row tiles: 2048 x 8192", 2048, 8192, 10),
            ("FF1 on tile 10 (640 tiles of 128 rows)", 2048, 10240, 10), ("FF1 on tile 11: 256 x (192 x 256) + 256 x (128 x 256) in ONE launch", 2048, 10240, 11)]
    t = {}
    for name, M, N, tile in rows:
        t[name] = run(M, N, K, tile)
        print(f"{name:64s} {t[name]:7.1f} us  {2.0 * M * N * K / t[name] / 1e6:6.0f} TF", flush=True)
    r192, r192x2 = t[rows[2][0]], t[rows[3][0]]
    r256, r256x2 = t[rows[4][0]], t[rows[5][0]]
    loop192, fix192 = r192x2 - r192, 2 * r192 - r192x2
    loop256, fix256 = r256x2 - r256, 2 * r256 - r256x2
    print(f"per round: 192-row tiles {loop192:.1f} us + {fix192:.1f} fixed; 256-row tiles {loop256:.1f} us + {fix256:.1f} fixed")
    est = fix192 + loop192 + loop256 / 2 * 1.08  # a 128-row round: half a 256-row round's loop, +8 % for the smaller tile's worse reads-per-MFMA ratio
    print(f"estimate for one launch of 256 x (192 x 256) + 256 x (128 x 256): {est:.1f} us against {t[rows[0][0]]:.1f} today")
    r128, r128x2 = t[rows[6][0]], t[rows[7][0]]
    loop128, fix128 = r128x2 - r128, 2 * r128 - r128x2
    print(f"MEASURED 128-row tiles per round: {loop128:.1f} us + {fix128:.1f} fixed  ->  192-row round + 128-row round in one launch: {fix192 + loop192 + loop128:.1f} us")
    # correctness of the 128-row instance on this very shape (GEGLU epilogue, 2.5 rounds, persistent)
    x = torch.randn(2048, K, device=dev).to(dt)
    wd = (torch.randn(10240, K, device=dev) * K ** -0.5).to(dt)
    w = native.KBlocked(wd)
    o10 = torch.empty(2048, 5120, device=dev, dtype=dt)
    o9 = torch.empty_like(o10)
    o11 = torch.full_like(o10, float("nan"))
    native.gemm([(x, w)], o10, geglu=True, tile=10)
    native.gemm([(x, w)], o9, geglu=True, tile=9)
    lib = native.load()
    import ctypes
    lib.mi355x_get_stat.argtypes = [ctypes.c_char_p]
    n0 = lib.mi355x_get_stat(b"g11")
    native.gemm([(x, w)], o11, geglu=True, tile=11)
    torch.cuda.synchronize()
    print(f"tile 10 against tile 9 on FF1: max |d| {(o10.float() - o9.float()).abs().max().item():.3e} (bit-equal expected: same K order per output)")
    print(f"tile 11 against tile 9 on FF1: max |d| {(o11.float() - o9.float()).abs().max().item():.3e}, nan {int(torch.isnan(o11.float()).sum())}, launches on the two-height path {lib.mi355x_get_stat(b'g11') - n0}")


if __name__ == "__main__":
    main()
