"""Time the UNet's hot GEMM / conv shapes under every tile configuration (A/B inside one process, interleaved rounds)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402

dev = "cuda"
dt = torch.bfloat16


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


def set_tile(v, st=0):
    native.load().mi355x_set_option(b"tile", v)
    native.load().mi355x_set_option(b"stages", st)


def main():
    rows = []
    GEMMS = [  # (M, K, N, geglu, count per step)
        (2048, 1280, 1280, False, 250), (2048, 1280, 2560, False, 60), (1280, 1280, 2048, False, 60), (2048, 1280, 10240, True, 60),
        (2048, 5120, 1280, False, 60), (8192, 640, 640, False, 50), (8192, 640, 1280, False, 10), (640, 640, 8192, False, 10),
        (8192, 640, 5120, True, 10), (8192, 2560, 640, False, 10), (256, 2048, 1280, False, 120), (2048, 1280, 64, False, 0),
    ]
    for (M, K, N, geglu, cnt) in GEMMS:
        x = torch.randn(M, K, device=dev).to(dt)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
        o = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=dt)
        best = None
        for tile in (1, 2, 3, 4, 6):
            if geglu and tile in (2, 4):
                continue
            line = f"gemm M={M:5d} K={K:5d} N={N:5d} geglu={int(geglu)} tile={tile}:"
            for st in (2, 3):
                set_tile(tile, st)
                t = min(timeit(lambda: native.gemm([(x, w)], o, geglu=geglu)) for _ in range(3))
                tf = 2 * M * K * N / t / 1e12
                rows.append(dict(kind="gemm", M=M, K=K, N=N, geglu=geglu, tile=tile, stages=st, us=t * 1e6, tflops=tf))
                line += f"  s{st}: {t*1e6:7.1f} us {tf:6.1f} TF"
            print(line, flush=True)
        set_tile(0)
        t = min(timeit(lambda: native.gemm([(x, w)], o, geglu=geglu)) for _ in range(3))
        print(f"   auto: {t*1e6:8.1f} us {2*M*K*N/t/1e12:7.1f} TF", flush=True)
        tt = min(timeit(lambda: torch.matmul(x, w.t())) for _ in range(3))
        print(f"   torch(hipBLASLt): {tt*1e6:8.1f} us {2*M*K*N/tt/1e12:7.1f} TF", flush=True)
        rows.append(dict(kind="torch", M=M, K=K, N=N, us=tt * 1e6, tflops=2 * M * K * N / tt / 1e12))
    # split-K candidates: long K, few tiles
    for (M, K, N) in ((2048, 5120, 1280), (2048, 1280, 1280), (8192, 2560, 640)):
        x = torch.randn(M, K, device=dev).to(dt)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
        o = torch.empty(M, N, device=dev, dtype=dt)
        ws = torch.empty(4 * M * N, device=dev, dtype=torch.float32)
        for tile in (1, 2, 4):
            line = f"splitk gemm M={M} K={K} N={N} tile={tile}:"
            for ks in (1, 2, 3, 4):
                set_tile(0, 2)
                t = min(timeit(lambda: native.gemm([(x, w)], o, tile=tile, ksplit=ks, ws=ws)) for _ in range(3))
                line += f"  k{ks}: {t*1e6:7.1f} us {2*M*K*N/t/1e12:6.1f} TF"
            print(line, flush=True)
    for (B, C, Co, H) in ((2, 1280, 1280, 32), (2, 2560, 1280, 32), (2, 640, 640, 64)):
        x = torch.randn(B, H, H, C, device=dev).to(dt)
        w = (torch.randn(Co, 9 * C, device=dev) * (9 * C) ** -0.5).to(dt)
        o = torch.empty(B * H * H, Co, device=dev, dtype=dt)
        ws = torch.empty(4 * B * H * H * Co, device=dev, dtype=torch.float32)
        for tile in (1, 3):
            line = f"splitk conv C={C} Co={Co} H={H} tile={tile}:"
            for ks in (1, 2, 3, 4):
                set_tile(0, 2)
                t = min(timeit(lambda: native.conv_gemm([(x, w, 3, 1, 1)], o, B, H, H, tile=tile, ksplit=ks, ws=ws), iters=10) for _ in range(3))
                line += f"  k{ks}: {t*1e6:7.1f} us {2*B*H*H*9*C*Co/t/1e12:6.1f} TF"
            print(line, flush=True)
    CONVS = [(2, 1280, 1280, 32), (2, 2560, 1280, 32), (2, 320, 320, 128), (2, 640, 640, 64), (2, 1920, 640, 64), (2, 960, 320, 128), (2, 640, 320, 128)]
    for (B, C, Co, H) in CONVS:
        x = torch.randn(B, H, H, C, device=dev).to(dt)
        w = (torch.randn(Co, 9 * C, device=dev) * (9 * C) ** -0.5).to(dt)
        o = torch.empty(B * H * H, Co, device=dev, dtype=dt)
        for tile in (1, 3, 5, 0):
            line = f"conv B={B} C={C:5d} Co={Co:5d} H={H:4d} tile={tile}:"
            for st in ((2,) if tile else (0,)):
                set_tile(tile, st)
                t = min(timeit(lambda: native.conv_gemm([(x, w, 3, 1, 1)], o, B, H, H), iters=10) for _ in range(3))
                tf = 2 * B * H * H * 9 * C * Co / t / 1e12
                rows.append(dict(kind="conv", B=B, C=C, Co=Co, H=H, tile=tile, stages=st, us=t * 1e6, tflops=tf))
                line += f"  s{st}: {t*1e6:7.1f} us {tf:6.1f} TF"
            print(line, flush=True)
        set_tile(0)
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "probe_gemm.json").write_text(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
