"""GroupNorm (+ SiLU) with the statistics from the producer (finalize_cs + apply: the step's form) at the step's shapes -- CFG pair and 4 images per GPU --
for several grid sizes / loads in flight of the apply pass (mi355x_set_option gnwgs / gnunroll), inside a HIP graph.  `python tools/probe_gn_apply.py`"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402

dev, dt = "cuda", torch.bfloat16
N = 8


def graph_time(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters / N * 1e3)
    return best


def main():
    lib = native.load()
    shapes = [(2, 1024, 1280), (2, 4096, 640), (2, 16384, 320), (2, 1024, 2560), (8, 1024, 1280), (8, 4096, 640), (8, 16384, 320), (8, 16384, 640)]
    variants = [(512, 4), (1024, 4), (2048, 4), (4096, 4), (1024, 8), (2048, 8), (4096, 8)]
    if "--vae" in sys.argv:  # the VAE decoder's large GroupNorms (one 1024 x 1024 image)
        shapes = [(2, 16384, 320), (2, 4096, 640), (1, 262144, 256), (1, 1048576, 128)]
        variants = [(2048, 4), (2048, 8), (8192, 4), (8192, 8)]
    for B, HW, C in shapes:
        xs = [torch.randn(B, HW, C, device=dev).to(dt) for _ in range(4)]
        o = torch.empty(B, HW, C, device=dev, dtype=dt)
        g, be = torch.ones(C, device=dev, dtype=dt), torch.zeros(C, device=dev, dtype=dt)
        cs = torch.rand(native.colstats_shape(B * HW, C), device=dev, dtype=torch.float32) + 1.0
        row = []
        for wgs, un in variants:
            lib.mi355x_set_option(b"gnwgs", wgs)
            lib.mi355x_set_option(b"gnunroll", un)

            def fn():
                for i in range(N):
                    native.groupnorm_nhwc(xs[i % 4], g, be, 32, 1e-5, True, o, colstats=cs)
            t = graph_time(fn)
            row.append(f"{wgs}/{un}: {t:6.2f}us")
        gb = 2 * B * HW * C * 2 / 1e9
        print(f"B={B} HW={HW} C={C} ({gb * 1e3:.0f} MB): " + "  ".join(row), flush=True)
    lib.mi355x_set_option(b"gnwgs", 512)
    lib.mi355x_set_option(b"gnunroll", 4)


if __name__ == "__main__":
    main()
