"""Hot A/B of GEMM tile configurations on the step's shapes (one process, interleaved rounds, operands rotated over 6 sets so that nothing but
the weights of the current set is L2-resident), with hipBLASLt (torch.matmul) on the same operands as the yardstick.  Round 4 also ran the
one-workgroup-per-CU tiles 5 (256x128), 8 (128x256) and 7 (256x256, 8 waves) through it (profiles/r04_a_probe_tiles.log; they lost and are gone:
note that their GEGLU numbers in that log are void -- the epilogue is only instantiated for 64-column wave tiles).  `python tools/probe_tiles.py`"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402

dev, dt = "cuda", torch.bfloat16


def set_tile(v, st=0):
    native.load().mi355x_set_option(b"tile", v)
    native.load().mi355x_set_option(b"stages", st)


def main():
    shapes = [("FF1", 2048, 1280, 10240, True), ("QKV", 2048, 1280, 3840, False), ("FF2", 2048, 5120, 1280, False), ("proj", 2048, 1280, 1280, False),
              ("FF1x4", 8192, 1280, 10240, True), ("QKVx4", 8192, 1280, 3840, False), ("640", 8192, 640, 640, False), ("4096^3", 4096, 4096, 4096, False),
              ("projx4", 8192, 1280, 1280, False), ("FF2x4", 8192, 5120, 1280, False), ("640x4", 32768, 640, 640, False)]  # (N = 1280 at 4 images: 160 tiles of 256 rows, 215 of 192)
    if "--n1280" in sys.argv:
        shapes = shapes[-3:] + [s_ for s_ in shapes if s_[0] in ("FF1x4", "4096^3")]
    tiles = [(0, 0), (1, 2), (4, 2), (7, 0), (8, 0), (9, 0)]
    for name, M, K, N, geglu in shapes:
        sets = []
        for _ in range(6):
            x = torch.randn(M, K, device=dev).to(dt)
            w = native.KBlocked((torch.randn(N, K, device=dev) * K ** -0.5).to(dt))
            o = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=dt)
            sets.append((x, w, o))
        res = {t: [] for t in tiles}
        for rnd in range(3):
            for t in tiles:
                if geglu and t[0] in (2, 4):
                    continue
                try:
                    for x, w, o in sets:
                        native.gemm([(x, w)], o, geglu=geglu, tile=t[0], stages=t[1])
                    torch.cuda.synchronize()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(5):
                        for x, w, o in sets:
                            native.gemm([(x, w)], o, geglu=geglu, tile=t[0], stages=t[1])
                    b.record()
                    torch.cuda.synchronize()
                    res[t].append(a.elapsed_time(b) / 30 * 1e3)
                except Exception as exc:  # noqa: BLE001
                    res[t].append(float("nan"))
                    print(f"   tile {t}: {exc}")
        set_tile(0)
        x, w, o = sets[0]
        wd = w.dense()
        tt = []
        for _ in range(3):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                torch.matmul(x, wd.t())
            b.record()
            torch.cuda.synchronize()
            tt.append(a.elapsed_time(b) / 10 * 1e3)
        fl = 2.0 * M * K * N
        line = f"{name:7s} M={M} K={K} N={N}: " + "  ".join(f"t{t[0]}/{t[1]}: {min(v):6.1f}us {fl / min(v) / 1e6:5.0f}TF" for t, v in res.items() if v)
        print(line + f"  | hipBLASLt {min(tt):6.1f}us {fl / min(tt) / 1e6:5.0f}TF", flush=True)


if __name__ == "__main__":
    main()
