"""Per-tile fixed cost of the 8-wave loop: time of M = 8192, N = 10240 (1280 tiles of 256 x 256 = 5 per CU) against K, plain and GEGLU epilogue; a straight
line t = a + b K per tile.  Run once with the product library and once with REFINERS_AMD_LIB=<abl1 variant> (no epilogue) to split `a`.
`python tools/probe_g8_overhead.py`"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402

dev, dt = "cuda", torch.bfloat16


def run(M, K, N, geglu, tile):
    sets = []
    for _ in range(4):
        x = torch.randn(M, K, device=dev).to(dt)
        w = native.KBlocked((torch.randn(N, K, device=dev) * K ** -0.5).to(dt))
        o = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=dt)
        sets.append((x, w, o))
    best = 1e9
    for _ in range(3):
        for x, w, o in sets:
            native.gemm([(x, w)], o, geglu=geglu, tile=tile)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            for x, w, o in sets:
                native.gemm([(x, w)], o, geglu=geglu, tile=tile)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 20 * 1e3)
    return best


def main():
    print("library:", native.loaded_library_path() or native.load() and native.loaded_library_path(), flush=True)
    M, N = 8192, 10240
    for geglu in (False, True):
        pts = []
        for K in (640, 1280, 2560, 5120):
            t = run(M, K, N, geglu, 7)
            tiles_per_cu = (M // 256) * (N // 256) / 256
            pts.append((K // 64, t / tiles_per_cu))
            print(f"geglu={int(geglu)} K={K}: {t:7.1f} us  {2.0 * M * K * N / t / 1e6:5.0f} TF  per tile {t / tiles_per_cu:6.2f} us ({K // 64} K tiles)", flush=True)
        (k0, t0), (k1, t1) = pts[0], pts[-1]
        b = (t1 - t0) / (k1 - k0)
        print(f"   per K tile {b:.3f} us, fixed per tile {t0 - b * k0:.2f} us", flush=True)


if __name__ == "__main__":
    main()
