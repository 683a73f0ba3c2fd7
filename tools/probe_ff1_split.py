"""Does splitting the GEGLU Linear's columns between the 8-wave loop and the 4-wave kernel pay?  At the CFG pair FF1 is M=2048 x N=10240: 320 tiles of 256 x 256 on 256
CUs = one full round and a quarter-full one.  Columns [0, 8192) make exactly one round of the 8-wave loop; the other 2048 are one round of 128 x 128 tiles.
Hot operands, HIP graph of 24 launches (pairs), bf16, LayerNorm folded, output K-blocked like the engine's."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402
from tools.probe_lora import graph_time  # noqa: E402

dev, dt = "cuda", torch.bfloat16
N_LAUNCH = 24


def case(M, K, N, n8):
    x = torch.randn(M, K, device=dev).to(dt)
    wd = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    w, w8, wr = native.KBlocked(wd), native.KBlocked(wd[:n8].contiguous()), native.KBlocked(wd[n8:].contiguous())
    stats = torch.zeros(K // 32, M, 2, device=dev)
    stats[..., 1] = 32.0
    ls, lc = torch.randn(N, device=dev), torch.randn(N, device=dev)
    out = torch.empty(M, N // 2, device=dev, dtype=dt)
    o8 = out.view(-1)[: M * n8 // 2].view(M, n8 // 2)
    orr = out.view(-1)[M * n8 // 2 :].view(M, (N - n8) // 2)
    res = {}
    for tile in (0, 1, 7):
        def one(tile=tile):
            for _ in range(N_LAUNCH):
                native.gemm([(x, w)], out, geglu=True, out_kblocked=True, ln=(stats, ls, lc, 1e-5), tile=tile)
        res[f"one launch tile {tile}"] = graph_time(one)
    for tr in (0, 1, 4):
        def two(tr=tr):
            for _ in range(N_LAUNCH):
                native.gemm([(x, wr)], orr, geglu=True, out_kblocked=True, ln=(stats, ls[n8:], lc[n8:], 1e-5), tile=tr)
                native.gemm([(x, w8)], o8, geglu=True, out_kblocked=True, ln=(stats, ls[:n8], lc[:n8], 1e-5), tile=7)
        res[f"split {n8}@7 + {N - n8}@{tr}"] = graph_time(two)
    def only8():
        for _ in range(N_LAUNCH):
            native.gemm([(x, w8)], o8, geglu=True, out_kblocked=True, ln=(stats, ls[:n8], lc[:n8], 1e-5), tile=7)
    res[f"only {n8}@7"] = graph_time(only8)
    fl = 2.0 * M * K * N
    best1 = min(v for k, v in res.items() if k.startswith("one"))
    print(f"M={M} K={K} N={N}: " + "  ".join(f"{k} {v:6.1f} us" for k, v in res.items()) + f"   | best single {fl / best1 / 1e6:.0f} TF", flush=True)
    # same bytes either way
    native.gemm([(x, w)], out, geglu=True, out_kblocked=True, ln=(stats, ls, lc, 1e-5), tile=1)
    ref = out.clone()
    out.zero_()
    native.gemm([(x, wr)], orr, geglu=True, out_kblocked=True, ln=(stats, ls[n8:], lc[n8:], 1e-5), tile=1)
    native.gemm([(x, w8)], o8, geglu=True, out_kblocked=True, ln=(stats, ls[:n8], lc[:n8], 1e-5), tile=7)
    torch.cuda.synchronize()
    print("   split == single launch:", bool(torch.equal(ref, out)), flush=True)


native.load()
case(2048, 1280, 10240, 8192)
case(4096, 1280, 10240, 8192)
case(8192, 640, 5120, 4096)
case(6144, 1280, 10240, 7680)   # 3 images: 24 x 40 = 960 tiles = 3.75 rounds; 24 x 32 = 768 = 3 rounds
