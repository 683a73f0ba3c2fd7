"""GroupNorm: single-launch kernel vs the three-kernel path on every GroupNorm shape of the SDXL step (CFG pair, bf16)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402


def timeit(fn, iters=50):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    lib = native.load()
    for (B, HW, C) in ((2, 1024, 1280), (2, 1024, 640), (2, 1024, 1920), (2, 1024, 2560), (2, 4096, 320), (2, 4096, 640), (2, 4096, 960), (2, 4096, 1280), (2, 4096, 1920),
                       (2, 16384, 320), (2, 16384, 640), (2, 16384, 960), (8, 1024, 1280), (8, 4096, 640)):
        x = torch.randn(B, HW, C, device="cuda").bfloat16()
        g, b = torch.randn(C, device="cuda").bfloat16(), torch.randn(C, device="cuda").bfloat16()
        o = torch.empty_like(x)
        line = f"B={B} HW={HW:5d} C={C:4d}:"
        lib.mi355x_groupnorm_set_fused(0, 0)
        line += f"  3-kernel {timeit(lambda: native.groupnorm_nhwc(x, g, b, 32, 1e-5, True, o)):7.1f} us"
        lib.mi355x_groupnorm_set_fused(1, 1 << 30)
        line += f"  fused {timeit(lambda: native.groupnorm_nhwc(x, g, b, 32, 1e-5, True, o)):7.1f} us"
        print(line, flush=True)
    lib.mi355x_groupnorm_set_fused(1, 160 << 10)


if __name__ == "__main__":
    main()
