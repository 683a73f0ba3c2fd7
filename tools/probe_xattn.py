"""q-projection + cross-attention: two launches (mi355x_gemm with the LayerNorm folded in, then mi355x_attention) vs ONE launch (xattn epilogue),
at the SDXL step's two shapes, inside a HIP graph over a rotation of 4 independent activation sets."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402

REPS = 6


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
    native.load()
    dt = torch.bfloat16
    for (B, Lq, K, H) in ((2, 1024, 1280, 20), (2, 4096, 640, 10), (8, 1024, 1280, 20)):
        C, M = 64 * H, B * Lq
        w = (torch.randn(C, K, device="cuda") * K ** -0.5).to(dt)
        wk = native.KBlocked(w)
        ls, lc = w.float().sum(1).contiguous(), torch.randn(C, device="cuda")
        k = torch.randn(B, 128, C, device="cuda", dtype=dt)
        vt = torch.randn(C, B, 128, device="cuda", dtype=dt)
        k2 = torch.randn(B, 64, C, device="cuda", dtype=dt)
        vt2 = torch.randn(C, B, 64, device="cuda", dtype=dt)
        streams = [(k, vt, 77, 1.0), (k2, vt2, 4, 0.6)]
        sets = []
        for _ in range(4):
            x = torch.randn(M, K, device="cuda", dtype=dt)
            xc = x.float().reshape(M, K // 32, 32)
            cm = xc.mean(2)
            stats = torch.stack([cm, ((xc - cm[:, :, None]) ** 2).sum(2)], 2).permute(1, 0, 2).contiguous()
            sets.append((x, stats, torch.empty(M, C, device="cuda", dtype=dt), torch.empty(M, C, device="cuda", dtype=dt)))

        def two(x, stats, q, o):
            native.gemm([(x, wk)], q, ln=(stats, ls, lc, 1e-5))
            native.attention(q.view(B, Lq, C), o.view(B, Lq, C), H, streams)

        def one(x, stats, q, o):
            native.gemm([(x, wk)], o, ln=(stats, ls, lc, 1e-5), xattn=(streams, Lq, None))

        def gemm_only(x, stats, q, o):
            native.gemm([(x, wk)], q, ln=(stats, ls, lc, 1e-5), tile=1)

        t2 = graph_us([(lambda s=s: two(*s)) for s in sets])
        t1 = graph_us([(lambda s=s: one(*s)) for s in sets])
        tg = graph_us([(lambda s=s: gemm_only(*s)) for s in sets])
        two(*sets[0])
        ref = sets[0][3].float().clone()
        one(*sets[0])
        err = (sets[0][3].float() - ref).abs().max().item() / ref.abs().max().item()
        print(f"B={B} Lq={Lq} K={K} heads={H}:  gemm + attention {t2:6.1f} us   one launch {t1:6.1f} us   (the 128x128-tile projection alone {tg:6.1f} us)   rel diff {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
