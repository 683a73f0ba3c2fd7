"""Where the in-launch LoRA's time goes, per shape class of the SDXL step (hot weights, HIP graph of N launches, bf16):
  plain      the un-adapted launch (what lora_mode="merged" runs)
  full       producers (or t-tiles) + tiles, epoch bumped before every launch (what lora_mode="fused" runs); full,prod / full,tt: the same with t
             forced to come from producer workgroups / from t-tiles (mi355x_set_option lora_dbg 64 / 128)
  nowait     the same launches WITHOUT the bump: the flags still hold the epoch, no tile ever waits (producers still run)
  tail       nowait + producers exit at once (mi355x_set_option lora_dbg 1): only the tiles' hand-off + up-projection remain
  as-plain / hooks-only / mma-only   tail minus: everything (the LoRA kernel variant doing an un-adapted launch's work) / the post-loop product /
             the hand-off (timing only; the results are then wrong by construction)
  bump+plain the un-adapted launch behind a bump kernel (the extra launch boundary `full` pays in this probe, not in the engine)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402

dev, dt = "cuda", torch.bfloat16
N_LAUNCH = 24


def graph_time(fn, iters=20):
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
        best = min(best, s.elapsed_time(e) / iters / N_LAUNCH * 1e3)
    return best  # us per launch


def case(M, K, N, *, geglu=False, ln=False, qkv=False, tile=0, ranks=(16, 16), only=None):
    x = torch.randn(M, K, device=dev).to(dt)
    w = native.KBlocked((torch.randn(N, K, device=dev) * K ** -0.5).to(dt))
    R = native.lora_rank(sum(ranks))
    groups = 3 if qkv else 1
    a = [native.KBlocked((torch.randn(R, K, device=dev) * K ** -0.5).to(dt)) for _ in range(groups)]
    bs = (torch.randn(N, R, device=dev) * 0.1).to(dt)
    out = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=dt)
    kw = {}
    lo_extra = ()
    if ln:
        stats = torch.zeros(K // 32, M, 2, device=dev)
        stats[..., 1] = 32.0
        kw["ln"] = (stats, torch.randn(N, device=dev), torch.randn(N, device=dev), 1e-5)
        lo_extra = (torch.randn(groups * R, device=dev), torch.randn(groups * R, device=dev))
    vt = None
    if qkv:
        C = N // 3
        out = torch.empty(M, 2 * C, device=dev, dtype=dt)
        vt = torch.empty(C, M, device=dev, dtype=dt)
        kw.update(out_t=vt, nt_begin=2 * C)
        lgroups = [(0, a[0]), (C, a[1]), (2 * C, a[2])]
    else:
        lgroups = [(0, a[0])]
    sync = native.LoraSync(torch.device(dev))
    t, flags = sync.scratch(groups, M, R, dt), sync.flags(groups, M)
    lora = (lgroups, bs) + lo_extra
    bias = None if ln else torch.randn(N, device=dev).to(dt)

    def plain():
        for _ in range(N_LAUNCH):
            native.gemm([(x, w)], out, bias=bias, geglu=geglu, tile=tile, **kw)

    def bump_plain():
        for _ in range(N_LAUNCH):
            sync.bump()
            native.gemm([(x, w)], out, bias=bias, geglu=geglu, tile=tile, **kw)

    def full():
        for _ in range(N_LAUNCH):
            sync.bump()
            native.gemm([(x, w)], out, bias=bias, geglu=geglu, tile=tile, lora=lora, lora_sync=(t, flags, sync), **kw)

    def nowait():
        for _ in range(N_LAUNCH):
            native.gemm([(x, w)], out, bias=bias, geglu=geglu, tile=tile, lora=lora, lora_sync=(t, flags, sync), **kw)

    lib = native.load()
    res = {}
    res["plain"] = graph_time(plain)
    if only == ("g8",):  # the 8-wave loop against the 4-wave kernel: un-adapted / adapted / adapted with nobody waiting
        res["bump+plain"] = graph_time(bump_plain)
        res["full"] = graph_time(full)
        res["nowait"] = graph_time(nowait)
        fl = 2.0 * M * K * N
        print(f"M={M} K={K} N={N} geglu={int(geglu)} ln={int(ln)} tile={tile} R={R}: " + "  ".join(f"{k} {v:7.2f} us" for k, v in res.items())
              + f"   | plain {fl / res['plain'] / 1e6:.0f} TF, (full - bump) / plain = {(res['full'] - (res['bump+plain'] - res['plain'])) / res['plain']:.3f}", flush=True)
        return res
    if only is None:
        res["bump+plain"] = graph_time(bump_plain)
        res["full"] = graph_time(full)
        lib.mi355x_set_option(b"lora_dbg", 64)  # t from producer workgroups everywhere
        res["full,prod"] = graph_time(full)
        lib.mi355x_set_option(b"lora_dbg", 128)  # t from t-tiles wherever the tile is wide enough for groups x rank columns
        res["full,tt"] = graph_time(full)
        lib.mi355x_set_option(b"lora_dbg", 0)
    sync.bump()
    native.gemm([(x, w)], out, bias=bias, geglu=geglu, tile=tile, lora=lora, lora_sync=(t, flags, sync), **kw)  # flags now hold the epoch
    torch.cuda.synchronize()
    if only is None:
        res["nowait"] = graph_time(nowait)
    for name, bits in (("tail", 1), ("as-plain", 1 | 4), ("hooks-only", 1 | 8), ("mma-only", 1 | 16)):
        if only is not None and name not in only:
            continue
        lib.mi355x_set_option(b"lora_dbg", bits)
        res[name] = graph_time(nowait)
    lib.mi355x_set_option(b"lora_dbg", 0)
    fl = 2.0 * M * K * N
    print(f"M={M} K={K} N={N} geglu={int(geglu)} ln={int(ln)} qkv={int(qkv)} tile={tile} R={R}: " + "  ".join(f"{k} {v:6.2f} us" for k, v in res.items())
          + (f"   | plain {fl / res['plain'] / 1e6:.0f} TF, full-(bump+plain) = {res['full'] - res['bump+plain']:+.2f} us" if only is None else ""), flush=True)


def main():
    if "--lib" in sys.argv:  # a bisect build of the library (scratch/bisect/lib_b<n>.so): only the as-plain variants are meaningful with it
        native.load(sys.argv[sys.argv.index("--lib") + 1])
    native.load()
    if "--bisect" in sys.argv:
        case(2048, 1280, 1280, tile=1, only=("plain", "as-plain"))
        return
    if "--g8" in sys.argv:  # round 5: the in-launch LoRA on the 8-wave loop (tile 7) beside the 4-wave kernel's (tile 1 / the heuristic), CFG-pair and 4-image sizes
        for M in (2048, 4096):
            for tile in (1, 7):
                case(M, 1280, 1280, tile=tile, only=("g8",))
                case(M, 1280, 10240, geglu=True, ln=True, tile=tile, only=("g8",))
                case(M, 5120, 1280, tile=tile, only=("g8",))
                case(M, 1280, 1280, ln=True, tile=tile, only=("g8",))
        for M in (8192, 16384):
            for tile in (1, 7):
                case(M, 640, 640, tile=tile, only=("g8",))
                case(M, 640, 5120, geglu=True, ln=True, tile=tile, only=("g8",))
                case(M, 2560, 640, tile=tile, only=("g8",))
        return
    if "--ff1" in sys.argv:  # round 6: the classes the tuning table sends to the 8-wave loop with live LoRAs (producers instead of t-tiles, persistent workgroups)
        for tile in (1, 9, 7):
            case(2048, 1280, 10240, geglu=True, ln=True, tile=tile, only=("g8",))
        for tile in (1, 7):
            case(8192, 640, 5120, geglu=True, ln=True, tile=tile, only=("g8",))
        for tile in (0, 9):
            case(2048, 1280, 1280, tile=tile, only=("g8",))
            case(2048, 5120, 1280, tile=tile, only=("g8",))
        return
    if "--ablate" in sys.argv:
        case(2048, 1280, 1280, tile=1)
        return
    for tile in (0, 1):
        case(2048, 1280, 1280, tile=tile)
    case(2048, 1280, 1280, ln=True)
    case(2048, 1280, 3840, ln=True, qkv=True)           # (the heuristic's tile: 128 x 64 -- too narrow for three groups' t columns)
    case(2048, 1280, 3840, ln=True, qkv=True, tile=1)   # what the engine's tuning table launches
    case(8192, 640, 1920, ln=True, qkv=True, tile=1)
    case(2048, 1280, 10240, geglu=True, ln=True)
    case(2048, 5120, 1280)
    case(8192, 640, 640)
    case(2048, 1280, 1280, ranks=(128,))


if __name__ == "__main__":
    main()
