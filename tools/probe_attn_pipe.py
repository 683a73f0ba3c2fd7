"""Attention at the SDXL step's shapes: K/V tiles in flight (1 / 2 register sets) x XCD-aware block order (mi355x_attention_set_pipeline).

Each configuration is timed over a rotation of 6 independent (q, k, v^T) sets (63-94 MB apiece at the large shapes), so that K / V come
from the Infinity Cache or HBM as they do in the step, not from a hot L2.
"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402


def time_us(fns, n=10, reps=5):
    """Per-launch time inside a HIP graph (no host launch cost: the short kernels are otherwise timed at the host's ~10 us per call)."""
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            for f in fns:
                f()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (n * reps * len(fns)) * 1e3


def main():
    lib = native.load()
    lib.mi355x_attention_set_pipeline.argtypes = [C.c_int, C.c_int]
    dt = torch.bfloat16
    shapes = ((2, 20, 1024, 1024, 0), (2, 10, 4096, 4096, 0), (2, 20, 1024, 77, 4), (2, 10, 4096, 77, 4), (8, 20, 1024, 1024, 0), (8, 10, 4096, 4096, 0))
    for (B, H, Lq, Lk, Lk2) in shapes:
        Cc = H * 64
        sets = []
        for _ in range(6):
            q = torch.randn(B, Lq, Cc, device="cuda", dtype=dt)
            out = torch.empty(B, Lq, Cc, device="cuda", dtype=dt)
            streams = []
            for L in filter(None, (Lk, Lk2)):
                k = torch.randn(B, L, Cc, device="cuda", dtype=dt)
                vt = torch.randn(Cc, B, (L + 63) // 64 * 64, device="cuda", dtype=dt)
                streams.append((k, vt, L, 1.0))
            sets.append((q, out, streams))
        fns = [(lambda s=s: native.attention(s[0], s[1], H, s[2])) for s in sets]
        line = f"B={B} H={H} Lq={Lq} Lk={Lk}{'+%d' % Lk2 if Lk2 else ''}:"
        ref = None
        variants = [("general", 0x11 | 0x40000, 1), ("default", 0x11, 1)]
        if Lk2 == 0 and B == 2:  # where a tile's time goes: pieces removed (results are wrong by construction)
            variants += ([] if "--ablate" not in sys.argv else [("key-split never", 0x11 | (1 << 16), 1)]) and [("-kvload", 1 | (1 << 8), 1), ("-softmax", 1 | (2 << 8), 1), ("-pv", 1 | (4 << 8), 1), ("-qk", 1 | (8 << 8), 1), ("-softmax-pv", 1 | (6 << 8), 1),
                         ("barriers+loads only", 1 | (14 << 8), 1), ("barriers only", 1 | (15 << 8), 1), ("  and no store", 1 | (31 << 8), 1),
                         ("  and no Q load", 1 | (63 << 8), 1), ("  and no K/V tile 0", 1 | (127 << 8), 1), ("full, no store", 1 | (16 << 8), 1)]
        if Lk2 == 0:  # round 6: 16-query waves (64-query workgroups of 4 waves / 128-query workgroups of 8 waves), key-split forced / forbidden
            variants += [("4 waves x 16 queries", 0x11 | (14 << 24), 1), ("8 waves x 16 queries", 0x11 | (18 << 24), 1), ("key-split always", 0x11 | (2 << 16), 1), ("key-split never", 0x11 | (1 << 16), 1)]
        for name, code, xcd in variants:
            lib.mi355x_attention_set_nw((code >> 24) & 31)
            code &= (1 << 24) - 1
            lib.mi355x_attention_set_pipeline(code, xcd)
            us = time_us(fns)
            o = sets[0][1].float().clone()
            if ref is None:
                ref = o
            same = bool(torch.allclose(ref, o, atol=2e-2, rtol=2e-2)) or ((code >> 8) & 255) != 0
            line += f"\n    {name:22s} {us:7.1f} us {4.0 * B * H * Lq * (Lk + Lk2) * 64 / us / 1e6:6.0f} TF{'' if same else ' DIFF'}"
        lib.mi355x_attention_set_nw(0)
        native.attention_pipeline_from_env()
        print(line, flush=True)


if __name__ == "__main__":
    main()
