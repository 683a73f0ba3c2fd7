"""First contact + race screen + timing of the 8-wave / eight-phase GEMM loop (tile 7, csrc/gemm8_kernel.cuh) beside tile 1 and hipBLASLt.
`python tools/probe_g8.py [--quick]`"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402

dev = "cuda"


def check(M, K, N, dt, *, bias=True, res=True, kblocked=False, tile=7, reps=1):
    g = torch.Generator(device="cpu").manual_seed(M * 7 + K * 3 + N)
    x = torch.randn(M, K, generator=g).to(dev, dt)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev, dt)
    b = torch.randn(N, generator=g).to(dev, dt) if bias else None
    r = torch.randn(M, N, generator=g).to(dev, dt) if res else None
    ref = x.float() @ w.float().t()
    if bias:
        ref = ref + b.float()
    if res:
        ref = ref + r.float()
    wk = native.KBlocked(w) if kblocked else w
    first = None
    worst = 0.0
    for _ in range(reps):
        out = torch.full((M, N), float("nan"), dtype=dt, device=dev)
        native.gemm([(x, wk)], out, bias=b, res=r, tile=tile)
        torch.cuda.synchronize()
        if first is None:
            first = out.clone()
        elif not torch.equal(first, out):
            return float("inf"), "NOT REPRODUCIBLE"
        o = out.float()
        if not torch.isfinite(o).all():
            bad = (~torch.isfinite(o)).nonzero()
            return float("nan"), f"non-finite at {bad[:4].tolist()} ({bad.shape[0]} elements)"
        worst = max(worst, ((o - ref).abs().max() / ref.abs().max()).item())
    return worst, ""


def main():
    quick = "--quick" in sys.argv
    print(native.device_info(), flush=True)
    bad = 0
    cases = [(256, 64, 256), (256, 128, 256), (256, 192, 256), (256, 256, 256), (256, 320, 256), (256, 384, 256), (512, 1280, 768), (300, 448, 520), (2048, 1280, 1280), (1000, 640, 330)]
    for tile in (7, 8):
        for dt in (torch.bfloat16, torch.float32):
            tol = 1.6e-2 if dt == torch.bfloat16 else 1e-4
            for (M, K, N) in cases:
                if dt == torch.float32 and K * 4 % 128:
                    continue
                for kbl in (False, True):
                    e, msg = check(M, K, N, dt, kblocked=kbl, tile=tile)
                    ok = e <= tol
                    bad += not ok
                    print(f"{'ok  ' if ok else 'FAIL'} tile {tile} {str(dt)[6:]:9s} {M}x{K}x{N} kblocked={int(kbl)} rel err {e:.3e} {msg}", flush=True)
        # race screen: long-K launches with many tiles, repeated; every result must equal the first bit for bit and match the reference
        for (M, K, N) in [(2048, 5120, 1280), (2048, 1280, 10240), (2048, 1280, 3840), (4096, 4096, 4096), (1000, 2560, 1500)] + ([] if quick else [(8192, 1280, 10240)]):
            e, msg = check(M, K, N, torch.bfloat16, kblocked=True, reps=6 if quick else 12, tile=tile)
            ok = e <= 1.6e-2
            bad += not ok
            print(f"{'ok  ' if ok else 'FAIL'} tile {tile} race screen {M}x{K}x{N}: rel err {e:.3e} {msg}", flush=True)
    for sk in native._streamk_eager.values():
        sk.check()
    # stream-K with fewer workgroups than CUs, and with a number that does not divide anything
    for skg in (97, 200):
        native.load().mi355x_set_option(b"skg", skg)
        for (M, K, N) in [(2048, 1280, 3840), (1000, 2560, 1500), (2048, 5120, 1280)]:
            e, msg = check(M, K, N, torch.bfloat16, kblocked=True, reps=3, tile=8)
            ok = e <= 1.6e-2
            bad += not ok
            print(f"{'ok  ' if ok else 'FAIL'} tile 8 with {skg} workgroups {M}x{K}x{N}: rel err {e:.3e} {msg}", flush=True)
    native.load().mi355x_set_option(b"skg", 0)
    for sk in native._streamk_eager.values():
        sk.check()
    print("FAILURES:", bad, flush=True)


if __name__ == "__main__":
    main()
