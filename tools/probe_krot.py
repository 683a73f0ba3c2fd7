"""A/B: K-block rotation per workgroup (L2 channel spreading) on the UNet's GEMM / conv shapes, plus hipBLASLt (torch.matmul) as a yardstick."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402

dev, dt = "cuda", torch.bfloat16


def timeit(fn, iters=30):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def opt(name, v):
    native.load().mi355x_set_option(name, v)


def main():
    GEMMS = [(2048, 1280, 1280, False), (2048, 1280, 2560, False), (1280, 1280, 2048, False), (2048, 1280, 10240, True), (2048, 5120, 1280, False),
             (8192, 640, 640, False), (8192, 640, 1280, False), (8192, 640, 5120, True), (8192, 2560, 640, False), (8192, 1280, 10240, True), (8192, 5120, 1280, False)]
    for (M, K, N, geglu) in GEMMS:
        x = torch.randn(M, K, device=dev).to(dt)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
        o = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=dt)
        line = f"gemm M={M:5d} K={K:5d} N={N:5d} geglu={int(geglu)}:"
        for tile in (0, 1):
            for kr in (0, 1, 3, 7):
                opt(b"tile", tile)
                opt(b"krot", kr)
                t = min(timeit(lambda: native.gemm([(x, w)], o, geglu=geglu)) for _ in range(3))
                line += f"  t{tile}r{kr}: {t*1e6:6.1f}us {2*M*K*N/t/1e12:5.0f}TF"
        opt(b"tile", 0)
        opt(b"krot", 0)
        tt = min(timeit(lambda: torch.matmul(x, w.t())) for _ in range(3))
        line += f"  hipBLASLt: {tt*1e6:6.1f}us {2*M*K*N/tt/1e12:5.0f}TF"
        print(line, flush=True)
    for (B, C, Co, H) in ((2, 1280, 1280, 32), (2, 2560, 1280, 32), (2, 640, 640, 64), (2, 1920, 640, 64), (2, 320, 320, 128), (2, 960, 320, 128)):
        x = torch.randn(B, H, H, C, device=dev).to(dt)
        w = (torch.randn(Co, 9 * C, device=dev) * (9 * C) ** -0.5).to(dt)
        o = torch.empty(B * H * H, Co, device=dev, dtype=dt)
        line = f"conv B={B} C={C:5d} Co={Co:5d} H={H:4d}:"
        for kr in (0, 1, 3, 7):
            opt(b"krot", kr)
            t = min(timeit(lambda: native.conv_gemm([(x, w, 3, 1, 1)], o, B, H, H)) for _ in range(3))
            line += f"  r{kr}: {t*1e6:7.1f}us {2*B*H*H*9*C*Co/t/1e12:5.0f}TF"
        opt(b"krot", 0)
        print(line, flush=True)


if __name__ == "__main__":
    main()
