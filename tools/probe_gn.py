"""GroupNorm on every GroupNorm shape of the SDXL step (CFG pair, bf16): single-launch kernel vs the three-kernel path, timed as the step
runs them -- captured in a HIP graph (no host launch cost), rotating over 4 independent tensors so that the input is not L2-hot."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402

REPS = 8


def graph_us(fns, iters=10):
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS):
            for f in fns:
                f()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (iters * REPS * len(fns)) * 1e3


def main():
    lib = native.load()
    for (B, HW, C) in ((2, 1024, 1280), (2, 1024, 640), (2, 1024, 1920), (2, 1024, 2560), (2, 4096, 320), (2, 4096, 640), (2, 4096, 960), (2, 4096, 1280), (2, 4096, 1920),
                       (2, 16384, 320), (2, 16384, 640), (2, 16384, 960), (8, 1024, 1280), (8, 4096, 640), (8, 16384, 320)):
        xs = [torch.randn(B, HW, C, device="cuda").bfloat16() for _ in range(4)]
        g, b = torch.randn(C, device="cuda").bfloat16(), torch.randn(C, device="cuda").bfloat16()
        os_ = [torch.empty_like(x) for x in xs]
        fns = [(lambda x=x, o=o: native.groupnorm_nhwc(x, g, b, 32, 1e-5, True, o)) for x, o in zip(xs, os_)]
        mb = 2 * B * HW * C * 2 / 1e6
        line = f"B={B} HW={HW:5d} C={C:4d} ({mb:5.1f} MB in+out):"
        lib.mi355x_groupnorm_set_fused(0, 0)
        t3 = graph_us(fns)
        line += f"  3-kernel {t3:6.1f} us ({mb / t3:4.2f} TB/s)"
        lib.mi355x_groupnorm_set_fused(1, 1 << 30)
        t1 = graph_us(fns)
        line += f"  fused {t1:6.1f} us"
        print(line, flush=True)
    lib.mi355x_groupnorm_set_fused(0, 160 << 10)


if __name__ == "__main__":
    main()
