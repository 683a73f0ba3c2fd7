"""Where a 256 x 256 tile's time goes: wall-clock stamps (100 MHz) from workgroup 0 of a build with -DMI355X_G8_ABL=4 (REFINERS_AMD_LIB=<variant>):
per tile: setup + first K tile landing | K loop | epilogue issue."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402

dev, dt = "cuda", torch.bfloat16
M, K, N = 8192, 1280, 10240
for geglu in (False, True):
    x = torch.randn(M, K, device=dev).to(dt)
    w = native.KBlocked((torch.randn(N, K, device=dev) * K ** -0.5).to(dt))
    o = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=dt)
    for _ in range(3):
        native.gemm([(x, w)], o, geglu=geglu, tile=8)
    torch.cuda.synchronize()
    sk = list(native._streamk_eager.values())[0]
    st = sk.ws[:512].view(torch.int64)[:32].cpu().tolist()
    t0 = st[0]
    rel = [(v - t0) / 100.0 for v in st[:20]]
    print(f"geglu={int(geglu)} stamps (us from start):", " ".join(f"{v:.2f}" for v in rel))
    rows = sk.ws[:512].view(torch.int64)[128:136].cpu().tolist()
    print("   last tile, epilogue row starts (us after the K loop's end):", " ".join(f"{(v - st[18]) / 100.0:.2f}" for v in rows), f"| end {(st[19] - st[18]) / 100.0:.2f}")
    for r in range(5):
        a, b, c, d = rel[4 * r: 4 * r + 4]
        nxt = rel[4 * r + 4] if 4 * r + 4 < len(rel) else float("nan")
        print(f"   tile {r}: setup+landing {b - a:.2f}  K loop {c - b:.2f}  epilogue {d - c:.2f}  to next {nxt - d:.2f}")
