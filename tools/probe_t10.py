"""Round 6 probe (variant build -DMI355X_PROBE_T10): 128 x 96 tiles with the four waves stacked along M (wave tile 32 x 96, NT = 6) against the product tiles
on shapes whose N is a multiple of 96 and whose tile count is 256 -- a stand-in for the 128 x 80 tile that would give N = 1280 exactly 256 workgroups (NT = 5: a
run of 20 columns per lane, which the shared epilogue cannot vectorise today).  Hot operands rotated over 6 sets.

    python -m refiners_amd.build_native --variant t10 MI355X_PROBE_T10=1 && REFINERS_AMD_LIB=refiners_amd/csrc/variants/libmi355x_refiners_t10.so python tools/probe_t10.py"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402

dev, dt = "cuda", torch.bfloat16


def main():
    native.load(os.environ.get("REFINERS_AMD_LIB"))
    # (M, K, N): N = 1536 = 16 x 96 -> 16 x 16 = 256 tiles of 128 x 96; the same FLOPs per tile as 128 x 80 x 1.2
    shapes = [("proj-like", 2048, 1280, 1536), ("FF2-like", 2048, 5120, 1536), ("proj", 2048, 1280, 1280), ("FF2", 2048, 5120, 1280)]
    tiles = [(0, 0), (1, 2), (2, 2), (3, 2), (4, 2), (6, 2), (10, 0), (11, 0)]
    if "--two-waves" in sys.argv:  # 128-thread workgroups: 64 x 64 (ids 12 / 13: 2 / 3 stages) and 128 x 64 (14 / 15) against the 4-wave product tiles
        shapes = shapes[2:] + [("QKV", 2048, 1280, 3840), ("640", 8192, 640, 640)]
        tiles = [(4, 2), (2, 2), (1, 2), (12, 0), (13, 0), (14, 0), (15, 0)]
    if "--stages" in sys.argv:  # the product tiles with 3 / 4 LDS stages (the in-launch LoRA instances exist with two only)
        shapes = shapes[2:] + [("QKV", 2048, 1280, 3840), ("640", 8192, 640, 640)]
        tiles = [(4, 2), (4, 3), (4, 4), (3, 2), (3, 3), (3, 4), (2, 2), (2, 3), (1, 2), (1, 3)]
    for name, M, K, N in shapes:
        sets = []
        for _ in range(6):
            x = torch.randn(M, K, device=dev).to(dt)
            w = native.KBlocked((torch.randn(N, K, device=dev) * K ** -0.5).to(dt))
            o = torch.empty(M, N, device=dev, dtype=dt)
            sets.append((x, w, o))
        ref = (sets[0][0].float() @ sets[0][1].dense().float().t())
        res = {t: [] for t in tiles}
        for rnd in range(3):
            for t in tiles:
                try:
                    for x, w, o in sets:
                        native.gemm([(x, w)], o, tile=t[0], stages=t[1])
                    torch.cuda.synchronize()
                    if rnd == 0:
                        err = float((sets[0][2].float() - ref).abs().max() / ref.abs().max())
                        assert err < 2e-2, (name, t, err)
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(5):
                        for x, w, o in sets:
                            native.gemm([(x, w)], o, tile=t[0], stages=t[1])
                    b.record()
                    torch.cuda.synchronize()
                    res[t].append(a.elapsed_time(b) / 30 * 1e3)
                except Exception as exc:  # noqa: BLE001
                    res[t].append(float("nan"))
                    print(f"   tile {t}: {exc}")
        fl = 2.0 * M * K * N
        print(f"{name:10s} M={M} K={K} N={N}: " + "  ".join(f"t{t[0]}/{t[1]}: {min(v):6.1f}us {fl / min(v) / 1e6:5.0f}TF" for t, v in res.items() if v), flush=True)


if __name__ == "__main__":
    main()
