"""Time the UNet's attention shapes: waves per workgroup, loader variants, vs torch SDPA."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402
from tools.probe_gemm import timeit  # noqa: E402


def main():
    lib = native.load()
    dt = torch.bfloat16
    for (B, H, L, Lk) in ((2, 20, 1024, 1024), (2, 10, 4096, 4096), (2, 20, 1024, 77), (2, 10, 4096, 77), (8, 20, 1024, 1024)):
        C = H * 64
        q = torch.randn(B, L, C, device="cuda").to(dt)
        k = torch.randn(B, (Lk + 63) // 64 * 64, C, device="cuda").to(dt)
        vt = torch.randn(C, B, (Lk + 63) // 64 * 64, device="cuda").to(dt)
        o = torch.empty_like(q)
        line = f"attn B={B} H={H} Lq={L} Lk={Lk}:"
        for nw in (2, 4):
            for glds in (1, 0):
                lib.mi355x_attention_set_nw(nw)
                lib.mi355x_attention_set_glds(glds)
                t = min(timeit(lambda: native.attention(q, o, H, [(k, vt, Lk, 1.0)]), iters=10) for _ in range(3))
                line += f"  nw{nw}/{'glds' if glds else 'regs'} {t*1e6:7.1f} us {4*B*L*Lk*C/t/1e12:6.1f} TF"
        lib.mi355x_attention_set_nw(0)
        lib.mi355x_attention_set_glds(1)
        if Lk == L:
            qh = q.reshape(B, L, H, 64).transpose(1, 2)
            t = min(timeit(lambda: torch.nn.functional.scaled_dot_product_attention(qh, qh, qh), iters=10) for _ in range(3))
            line += f"  torch_sdpa {t*1e6:7.1f} us {4*B*L*Lk*C/t/1e12:6.1f} TF"
        print(line, flush=True)


if __name__ == "__main__":
    main()
