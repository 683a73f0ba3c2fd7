"""Staggered 8-wave tile (7) vs the 4-wave 128x128 tile (1) and the lockstep 256x128 tile (5) on the step's large GEMM shapes, hot."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402


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


native.load()
for (M, K, N, geglu) in ((2048, 1280, 10240, True), (2048, 1280, 3840, False), (2048, 5120, 1280, False), (8192, 640, 5120, True), (8192, 2560, 640, False), (8192, 1280, 10240, True), (4096, 4096, 4096, False)):
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = native.KBlocked((torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16())
    o = torch.empty(M, N // 2 if geglu else N, device="cuda", dtype=torch.bfloat16)
    line = f"M={M:5d} K={K:5d} N={N:5d} geglu={int(geglu)}:"
    for tile, st in ((1, 2), (5, 3), (7, 3), (8, 3)):
        t = min(timeit(lambda: native.gemm([(x, w)], o, geglu=geglu, tile=tile, stages=st)) for _ in range(3))
        line += f"  tile{tile}/s{st} {t * 1e6:7.1f} us {2 * M * K * N / t / 1e12:6.0f} TF"
    print(line, flush=True)
