"""Cross-attention of the step (77 text keys + 4 image-prompt keys): the short-K/V kernel with 128- and 64-query workgroups (bits 23-24 of the pipeline code:
1 = 128 always, 2 = 64 always), timed like tools/probe_attn_pipe.py; every variant is compared with the first."""
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
    base = 1 | (13 << 4) | (3 << 19)
    variants = [("register-staged, 128-query", base | (1 << 23) | (1 << 25)), ("register-staged, 64-query", base | (2 << 23) | (1 << 25)), ("LDS-DMA + half tiles, 128-query", base | (1 << 23)),
                ("LDS-DMA + half tiles, 64-query", base | (2 << 23)), ("default", base)]
    for (B, H, Lq, Lk, Lk2) in ((2, 20, 1024, 77, 4), (2, 10, 4096, 77, 4), (2, 20, 1024, 77, 0), (8, 20, 1024, 77, 4), (8, 10, 4096, 77, 4), (2, 20, 1000, 128, 33), (2, 4, 200, 20, 100)):
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
        for name, code in variants:
            lib.mi355x_attention_set_pipeline(code, 1)
            us = time_us(fns)
            o = sets[0][1].float().clone()
            ref = o if ref is None else ref
            line += f"\n    {name:34s} {us:7.1f} us   max |d| vs first {(o - ref).abs().max().item():.1e}"
        native.attention_pipeline_from_env()
        print(line, flush=True)


if __name__ == "__main__":
    main()
