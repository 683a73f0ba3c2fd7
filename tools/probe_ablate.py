"""Which resource bounds the GEMM main loop?  Time the real kernel against builds with one ingredient removed."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402
from tools.probe_gemm import timeit  # noqa: E402

lib = native.load()
dt = torch.bfloat16
for (M, K, N) in ((2048, 1280, 10240), (8192, 2560, 640), (2048, 1280, 1280), (2048, 5120, 1280)):
    x = torch.randn(M, K, device="cuda").to(dt)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
    o = torch.empty(M, N, device="cuda", dtype=dt)
    for tile in (1, 4):
        lib.mi355x_set_option(b"tile", tile)
        lib.mi355x_set_option(b"stages", 2)
        line = f"M={M} K={K} N={N} tile={tile}:"
        for abl, name in ((0, "real"), (1, "noMFMA"), (2, "noLDSread"), (3, "noGLDS")):
            lib.mi355x_set_option(b"ablate", abl)
            t = min(timeit(lambda: native.gemm([(x, w)], o)) for _ in range(3))
            line += f"  {name} {t*1e6:7.1f} us"
        lib.mi355x_set_option(b"ablate", 0)
        print(line, flush=True)
