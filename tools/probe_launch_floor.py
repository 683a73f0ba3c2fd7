"""What one dependent launch costs inside a replayed HIP graph on this GPU: chains of N launches of (a) a tiny kernel (silu over 1 K elements), (b) silu over 2.6 M elements
(the step's 2 x 1024 x 1280 activation: ~640 workgroups of traffic-bound work), (c) pairs that alternate two buffers (a true dependency chain), timed per launch."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402


def chain_us(fn, n_launch=200, n_replay=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n_launch):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n_replay):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (n_replay * n_launch) * 1e3


def main():
    native.load()
    dt = torch.bfloat16
    for n in (1 << 10, 1 << 16, 1 << 20, 2 * 1024 * 1280, 2 * 4096 * 640 * 2, 2 * 16384 * 320 * 4):
        x = torch.randn(n, device="cuda", dtype=dt)
        y = torch.empty_like(x)
        state = {"i": 0}

        def step():
            if state["i"] & 1:
                native.silu(y, x)
            else:
                native.silu(x, y)
            state["i"] += 1

        us = chain_us(step)
        print(f"silu over {n:>9d} bf16 elements ({2 * n * 2 / 1e6:7.2f} MB moved): {us:6.2f} us per dependent launch  ({2 * n * 2 / us / 1e6:6.2f} TB/s)", flush=True)


if __name__ == "__main__":
    main()
