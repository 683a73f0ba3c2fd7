"""Self-attention at the SDXL step's shapes: OPT bits of attn_kernel (mi355x_attention_set_pipeline bits 4-7), time AND error.

    bit 0 = permlane reductions, bit 2 = lazy running maximum, bit 3 = row sums from the matrix pipe; bits 19-20 of the code = the software-pipelined loop.

Timed like tools/probe_attn_pipe.py (HIP graph over a rotation of 6 operand sets); the error of every variant is taken against a float32
softmax(Q K^T / 8) V of the same bf16 operands, on plain normal data and on a set whose scores GROW along the keys (every tile moves the
running maximum of most queries: the rescale path of the lazy variant)."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402
from tools.probe_attn_pipe import time_us  # noqa: E402


def reference(q, k, vt, H, Lk):
    B, Lq, Cc = q.shape
    qh = q.float().view(B, Lq, H, 64).transpose(1, 2)
    kh = k.float().view(B, Lk, H, 64).transpose(1, 2)
    vh = vt.float()[:, :, :Lk].reshape(H, 64, B, Lk).permute(2, 0, 3, 1)  # (B, H, Lk, 64)
    o = torch.softmax(qh @ kh.transpose(-1, -2) * 0.125, dim=-1) @ vh
    return o.transpose(1, 2).reshape(B, Lq, Cc)


def main():
    lib = native.load()
    lib.mi355x_attention_set_pipeline.argtypes = [C.c_int, C.c_int]
    dt = torch.bfloat16
    variants = [("opt 1 (round 5)", 1 << 4), ("opt 5 lazy", 5 << 4), ("opt 9 ones", 9 << 4), ("opt 13 lazy+ones", 13 << 4), ("pipelined", (13 << 4) | (1 << 19)),
                ("pipelined, order free", (13 << 4) | (2 << 19)), ("pipelined, LDS-DMA", (13 << 4) | (3 << 19)), ("  16-query waves", (13 << 4) | (3 << 19) | (3 << 16)), ("  32-query waves", (13 << 4) | (3 << 19) | (1 << 16)),
                ("pipelined, LDS-DMA, folded", (13 << 4) | (3 << 19) | (1 << 21)), ("  16-query waves", (13 << 4) | (3 << 19) | (1 << 21) | (3 << 16)), ("  32-query waves", (13 << 4) | (3 << 19) | (1 << 21) | (1 << 16))]
    if "--short" in sys.argv:
        variants = [v for v in variants if "opt" not in v[0] and "order free" not in v[0]]
    shapes = ((2, 20, 1024, 1024), (2, 10, 4096, 4096), (8, 20, 1024, 1024), (8, 10, 4096, 4096), (2, 20, 1000, 1000))
    for (B, H, Lq, Lk) in shapes:
        Cc = H * 64
        sets = []
        g = torch.Generator(device="cuda").manual_seed(7)
        for i in range(6):
            q = torch.randn(B, Lq, Cc, device="cuda", dtype=dt, generator=g)
            k = torch.randn(B, Lk, Cc, device="cuda", dtype=dt, generator=g)
            if i == 1:  # growing scores: key j is scaled by 1 + 3 j / Lk, queries get a common component
                k = (k.float() * (1.0 + 3.0 * torch.arange(Lk, device="cuda").view(1, Lk, 1) / Lk) + 0.5).to(dt)
                q = (q.float() + 0.5).to(dt)
            out = torch.empty(B, Lq, Cc, device="cuda", dtype=dt)
            vt = torch.randn(Cc, B, (Lk + 63) // 64 * 64, device="cuda", dtype=dt, generator=g)
            sets.append((q, out, [(k, vt, Lk, 1.0)]))
        fns = [(lambda s=s: native.attention(s[0], s[1], H, s[2])) for s in sets]
        refs = [reference(s[0], s[2][0][0], s[2][0][1], H, Lk) if B == 2 else None for s in sets[:2]]
        line = f"B={B} H={H} Lq={Lq} Lk={Lk}:"
        for name, code in variants:
            lib.mi355x_attention_set_pipeline(1 | code, 1)
            us = time_us(fns)
            err = ""
            if B == 2:
                for tag, s, r in zip(("normal", "growing"), sets[:2], refs):
                    d = s[1].float() - r
                    err += f"  {tag}: max {d.abs().max().item():.2e} rel-l2 {(d.norm() / r.norm()).item():.2e}"
            line += f"\n    {name:22s} {us:7.1f} us {4.0 * B * H * Lq * Lk * 64 / us / 1e6:6.0f} TF{err}"
        native.attention_pipeline_from_env()
        print(line, flush=True)


if __name__ == "__main__":
    main()
