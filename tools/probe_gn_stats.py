"""Conv2d -> GroupNorm(+SiLU) pairs of the SDXL step with and without the producer's column statistics (mi355x_gemm_args.colstats_out):
time per pair inside a HIP graph (N pairs over rotating buffers), and of the convolution / the GroupNorm alone.  `python tools/probe_gn_stats.py`"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402

dev, dt = "cuda", torch.bfloat16
N_PAIR = 12


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
        best = min(best, s.elapsed_time(e) / iters / N_PAIR * 1e3)
    return best


def case(B, Cin, Cout, H, tile=0, ksplit=1):
    M = B * H * H
    sets = []
    for _ in range(4):
        x = torch.randn(B, H, H, Cin, device=dev).to(dt)
        w = native.KBlocked(native.pack_conv_weight((torch.randn(Cout, Cin, 3, 3, device=dev) * (9 * Cin) ** -0.5).to(dt)))
        sets.append((x, w))
    b = torch.randn(Cout, device=dev).to(dt)
    g, be = torch.ones(Cout, device=dev, dtype=dt), torch.zeros(Cout, device=dev, dtype=dt)
    y, o = torch.empty(M, Cout, device=dev, dtype=dt), torch.empty(M, Cout, device=dev, dtype=dt)
    cs = torch.empty(native.colstats_shape(M, Cout), device=dev, dtype=torch.float32)
    ws = torch.empty(max(ksplit, 1) * M * Cout, device=dev, dtype=torch.float32) if ksplit > 1 else None

    def run(conv, gn, stats):
        def fn():
            for i in range(N_PAIR):
                x, w = sets[i % 4]
                if conv:
                    native.conv_gemm([(x, w, 3, 1, 1)], y, B, H, H, bias=b, tile=tile, ksplit=ksplit, ws=ws, colstats_out=cs if stats else None)
                if gn:
                    native.groupnorm_nhwc(y.view(B, H * H, Cout), g, be, 32, 1e-5, True, o.view(B, H * H, Cout), colstats=cs if stats else None)
        return graph_time(fn)

    r = {k: run(*v) for k, v in {"conv": (1, 0, 0), "conv+cs": (1, 0, 1), "gn3": (0, 1, 0), "gn2": (0, 1, 1), "pair": (1, 1, 0), "pair+cs": (1, 1, 1)}.items()}
    print(f"B={B} {Cin}->{Cout} {H}x{H} tile={tile} ksplit={ksplit}: " + "  ".join(f"{k} {v:7.2f}" for k, v in r.items()) + f"   | pair saves {r['pair'] - r['pair+cs']:+.2f} us", flush=True)


def main():
    native.load()
    case(2, 1280, 1280, 32, tile=1, ksplit=3)
    case(2, 1280, 1280, 32)
    case(2, 640, 640, 64)
    case(2, 1280, 640, 64)
    case(2, 320, 320, 128)
    case(2, 640, 320, 128)


if __name__ == "__main__":
    main()
