"""Does pulling the NEXT GEMM's weights into the Infinity Cache ahead of time pay?  Cold weights (a ring of distinct buffers
larger than the 256 MB Infinity Cache) vs the same launches preceded by a touch of the weights, vs hot (one buffer)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402

dev, dt = "cuda", torch.bfloat16


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e-3)
    return best


def main():
    native.load()
    for (M, K, N, geglu, ring) in ((2048, 1280, 10240, True, 48), (2048, 5120, 1280, False, 96), (2048, 1280, 1280, False, 200)):
        x = torch.randn(M, K, device=dev).to(dt)
        ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(dt) for _ in range(ring)]
        o = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=dt)
        sink = torch.zeros(1, device=dev, dtype=torch.int32)

        def touch(w):  # one 4-byte read per 64 bytes of the weight
            sink.add_(w.view(torch.int32).view(-1)[::16].sum().to(torch.int32))

        def cold():
            for w in ws:
                native.gemm([(x, w)], o, geglu=geglu)

        def hot():
            for _ in ws:
                native.gemm([(x, ws[0])], o, geglu=geglu)

        def touched():
            for w in ws:
                touch(w)
                native.gemm([(x, w)], o, geglu=geglu)

        def touches():
            for w in ws:
                touch(w)

        def linked():  # launch i carries launch i+1's weights as its prefetch span (extra workgroups of the same grid)
            for i, w in enumerate(ws):
                native.gemm([(x, w)], o, geglu=geglu, prefetch=ws[(i + 1) % len(ws)])

        tc, th, tt, to = (timed(f) / ring * 1e6 for f in (cold, hot, touched, touches))
        line = f"M={M} K={K} N={N}: cold {tc:6.1f} us  hot {th:6.1f} us  gemm after a torch touch {tt - to:6.1f} us | in-kernel prefetch of the next weights:"
        for mode in (1,):
            for blocks in (8, 16, 32, 64):
                native.load().mi355x_set_option(b"pfmode", mode)
                native.load().mi355x_set_option(b"pfblocks", blocks)
                line += f"  m{mode}b{blocks}: {timed(linked) / ring * 1e6:6.1f}"
        native.load().mi355x_set_option(b"pfmode", 1)
        native.load().mi355x_set_option(b"pfblocks", 8)
        print(line, flush=True)


if __name__ == "__main__":
    main()
