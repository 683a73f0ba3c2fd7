"""Micro-program for the HBM-side traffic counters: the step's dominant GEMM / conv shape classes, each launched REPS times in a row on fresh
operands (weights K-blocked like the engine's), nothing else on the stream.  tools/profile_round.py runs it under `rocprofv3 --pmc FETCH_SIZE`
/ `WRITE_SIZE` when the whole-step counter passes die in the profiler (they did all through round 4) and folds the result per class.
Prints one JSON line: the class list in launch order with their algorithmic bytes."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402

dev, dt = "cuda", torch.bfloat16
REPS = 6


def main():
    out = []
    shapes = [("gemm:bf16:2048x10240x1280:geglu (FF1)", 2048, 1280, 10240, True, 60), ("gemm:bf16:2048x1280x1280 (out-projection)", 2048, 1280, 1280, False, 192),
              ("gemm:bf16:2048x1280x5120 (FF2)", 2048, 5120, 1280, False, 60), ("gemm:bf16:2048x3840x1280 (Q|K|V)", 2048, 1280, 3840, False, 60),
              ("gemm:bf16:8192x10240x1280:geglu (FF1, 4 images)", 8192, 1280, 10240, True, 0)]
    for name, M, K, N, geglu, per_step in shapes:
        sets = []
        for _ in range(REPS):
            x = torch.randn(M, K, device=dev).to(dt)
            w = native.KBlocked((torch.randn(N, K, device=dev) * K ** -0.5).to(dt))
            o = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=dt)
            sets.append((x, w, o))
        torch.cuda.synchronize()
        for x, w, o in sets:
            native.gemm([(x, w)], o, geglu=geglu)
        torch.cuda.synchronize()
        out.append({"class": name, "launches": REPS, "per_step": per_step, "algorithmic_bytes": 2 * (M * K + N * K + M * (N // 2 if geglu else N))})
    # one 3x3 convolution of the 64 x 64 level (M = 8192, 640 -> 640 channels)
    B, H, W, C = 2, 64, 64, 640
    sets = []
    for _ in range(REPS):
        x = torch.randn(B, H, W, C, device=dev).to(dt)
        w = native.KBlocked(native.pack_conv_weight((torch.randn(C, C, 3, 3, device=dev) * (9 * C) ** -0.5).to(dt)))
        o = torch.empty(B * H * W, C, device=dev, dtype=dt)
        sets.append((x, w, o))
    torch.cuda.synchronize()
    for x, w, o in sets:
        native.conv_gemm([(x, w, 3, 1, 1)], o, B, H, W)
    torch.cuda.synchronize()
    out.append({"class": "conv:bf16:8192x640x5760 (3x3, 64 x 64 level)", "launches": REPS, "per_step": 6, "algorithmic_bytes": 2 * (B * H * W * C * 2 + 9 * C * C)})
    print("TRAFFIC_PROGRAM " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
