"""Attention at the SDXL step's shapes: 4-wave (128-query) vs 2-wave (64-query) workgroups."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402


def time_us(fn, n=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    lib = native.load()
    dt = torch.bfloat16
    for (B, H, Lq, Lk) in ((2, 20, 1024, 1024), (2, 10, 4096, 4096), (2, 20, 1024, 77), (2, 10, 4096, 77), (8, 20, 1024, 1024)):
        C = H * 64
        q = torch.randn(B, Lq, C, device="cuda", dtype=dt)
        k = torch.randn(B, Lk, C, device="cuda", dtype=dt)
        lkp = (Lk + 63) // 64 * 64
        vt = torch.randn(C, B, lkp, device="cuda", dtype=dt)
        out = torch.empty(B, Lq, C, device="cuda", dtype=dt)
        line = f"B={B} H={H} Lq={Lq} Lk={Lk}:"
        for glds in (0,):
            for nw in (4, 14, 18):
                lib.mi355x_attention_set_nw(nw)
                lib.mi355x_attention_set_glds(glds)
                us = time_us(lambda: native.attention(q, out, H, [(k, vt, Lk, 1.0)]))
                line += f"  glds{glds}nw{nw}: {us:7.1f} us {4.0 * B * H * Lq * Lk * 64 / us / 1e6:6.0f} TF"
        lib.mi355x_attention_set_nw(0)
        lib.mi355x_attention_set_glds(0)
        print(line, flush=True)


if __name__ == "__main__":
    main()
