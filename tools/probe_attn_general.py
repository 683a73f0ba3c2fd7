"""Throughput of mi355x_attention_general at the SD1.5 and SAM ViT-H shapes (bf16), against torch SDPA on the same GPU."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from refiners_amd import native  # noqa: E402


def time_ms(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    native.load()
    dt = torch.bfloat16
    rows = []
    for name, B, H, L, Dq, Dv in (("sd1_64x64_d40", 2, 8, 4096, 40, 40), ("sd1_32x32_d80", 2, 8, 1024, 80, 80), ("sd1_16x16_d160", 2, 8, 256, 160, 160),
                                  ("sam_window_14x14", 25, 16, 196, 112, 80), ("sam_global_64x64", 1, 16, 4096, 208, 80), ("d64_reference_point", 2, 10, 4096, 64, 64)):
        q = torch.randn(B, L, H * Dq, device="cuda", dtype=dt)
        k = torch.randn(B, L, H * Dq, device="cuda", dtype=dt)
        v = torch.randn(B, L, H * Dv, device="cuda", dtype=dt)
        lkp = (L + 63) // 64 * 64
        vt = torch.zeros(H * Dv, B, lkp, device="cuda", dtype=dt)
        vt[:, :, :L] = v.permute(2, 0, 1)
        out = torch.empty(B, L, H * Dv, device="cuda", dtype=dt)
        lib = native.load()
        lib.mi355x_attention_general_set_fast(0)
        ms5 = time_ms(lambda: native.attention_general(q, k, vt, out, H, L))
        o5 = out.float().clone()
        lib.mi355x_attention_general_set_fast(1)
        ms = time_ms(lambda: native.attention_general(q, k, vt, out, H, L))
        flop = 2.0 * B * H * L * L * (Dq + Dv)
        row = {"shape": name, "ms": round(ms, 4), "tflops": round(flop / ms / 1e9, 1), "round5_instance_ms": round(ms5, 4), "max_abs_vs_round5": round((out.float() - o5).abs().max().item(), 5)}
        if Dq == Dv:
            qh, kh, vh = (t.view(B, L, H, Dq).transpose(1, 2) for t in (q, k, v))
            row["torch_sdpa_ms"] = round(time_ms(lambda: F.scaled_dot_product_attention(qh, kh, vh)), 4)
            if Dq == 64:
                row["flash64_ms"] = round(time_ms(lambda: native.attention(q, out, H, [(k, vt, L, 1.0)])), 4)
        rows.append(row)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
