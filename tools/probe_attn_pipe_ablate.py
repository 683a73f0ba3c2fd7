"""Anatomy of the software-pipelined attention loop (attn_pipe_kernel): pieces removed in a probing build (results wrong by construction).

Build:  hipcc ... -DMI355X_ATTN_PIPE_ABL=1 attention.hip -> csrc/variants/libmi355x_refiners_pipeabl.so   (tools/build_attn_variant.sh pipeabl MI355X_ATTN_PIPE_ABL=1)
Run:    REFINERS_AMD_LIB=refiners_amd/csrc/variants/libmi355x_refiners_pipeabl.so python tools/probe_attn_pipe_ablate.py
"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402
from tools.probe_attn_pipe import time_us  # noqa: E402


def main():
    lib = native.load()
    lib.mi355x_attention_set_pipeline.argtypes = [C.c_int, C.c_int]
    dt = torch.bfloat16
    variants = [("attn_kernel opt 13", 13 << 4), ("pipelined", (13 << 4) | (1 << 19)), ("- exponentials", (13 << 4) | (1 << 19) | (1 << 8)), ("- maxima", (13 << 4) | (1 << 19) | (32 << 8)),
                ("- exponentials - maxima", (13 << 4) | (1 << 19) | (33 << 8)), ("- Q K^T MFMAs", (13 << 4) | (1 << 19) | (2 << 8)), ("- P V MFMAs", (13 << 4) | (1 << 19) | (4 << 8)),
                ("- all MFMAs", (13 << 4) | (1 << 19) | (6 << 8)), ("- K/V loads + commits", (13 << 4) | (1 << 19) | (8 << 8)), ("- barrier", (13 << 4) | (1 << 19) | (16 << 8)),
                ("- loads - barrier", (13 << 4) | (1 << 19) | (24 << 8)), ("- all vector work - all MFMAs", (13 << 4) | (1 << 19) | (39 << 8)), ("LDS reads + loop only", (13 << 4) | (1 << 19) | (63 << 8)),
                ("LDS-DMA", (13 << 4) | (1 << 19) | (128 << 8)), ("LDS-DMA - exponentials", (13 << 4) | (1 << 19) | (129 << 8)), ("LDS-DMA - all MFMAs", (13 << 4) | (1 << 19) | (134 << 8)),
                ("LDS-DMA - barrier", (13 << 4) | (1 << 19) | (144 << 8)), ("LDS-DMA - all vector work - all MFMAs", (13 << 4) | (1 << 19) | (167 << 8)),
                ("LDS-DMA - loads", (13 << 4) | (1 << 19) | (136 << 8)), ("LDS-DMA - loads - barrier", (13 << 4) | (1 << 19) | (152 << 8)),
                ("LDS-DMA - vector - MFMAs - loads", (13 << 4) | (1 << 19) | (175 << 8)), ("LDS-DMA: LDS reads + loop only", (13 << 4) | (1 << 19) | (191 << 8))]
    if "--dma" in sys.argv:
        variants = [v for v in variants if "LDS-DMA" in v[0]]
    for (B, H, Lq, Lk) in ((8, 10, 4096, 4096), (2, 10, 4096, 4096)):
        Cc = H * 64
        sets = []
        for _ in range(6):
            q = torch.randn(B, Lq, Cc, device="cuda", dtype=dt)
            k = torch.randn(B, Lk, Cc, device="cuda", dtype=dt)
            out = torch.empty(B, Lq, Cc, device="cuda", dtype=dt)
            vt = torch.randn(Cc, B, Lk, device="cuda", dtype=dt)
            sets.append((q, out, [(k, vt, Lk, 1.0)]))
        fns = [(lambda s=s: native.attention(s[0], s[1], H, s[2])) for s in sets]
        line = f"B={B} H={H} Lq={Lq} Lk={Lk}:"
        for name, code in variants:
            lib.mi355x_attention_set_pipeline(1 | code, 1)
            us = time_us(fns)
            line += f"\n    {name:32s} {us:7.1f} us {4.0 * B * H * Lq * Lk * 64 / us / 1e6:6.0f} TF"
        native.attention_pipeline_from_env()
        print(line, flush=True)


if __name__ == "__main__":
    main()
