"""Per-entry-point replay of the SAM ViT-H encoder's recorded program (configs[4]; `--vae`: of the VAE decoder's): launches, ms and share per entry point and per GEMM shape class."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402
from refiners_amd import native  # noqa: E402
from refiners_amd.engine.sam import CompiledSAMViT  # noqa: E402
from refiners_amd.segment_anything import SAMViTAdapter, SAMViTH  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    native.load()
    if "--vae" in sys.argv:  # the SDXL VAE decoder, one 128 x 128 latent -> 1024 x 1024 image
        from refiners_amd.engine.vae import CompiledVAEDecoder
        from refiners_amd.latent_diffusion.vae import SDXLAutoencoder

        vae = SDXLAutoencoder(device="meta")
        bench.gpu_weights(vae, seed=7, dtype=torch.bfloat16, device=dev)
        fast = CompiledVAEDecoder(vae)
        with torch.no_grad():
            fast((torch.randn(1, 4, 128, 128, device=dev) * 0.13).to(torch.bfloat16))
    else:
        vit = SAMViTH(device="meta")
        bench.gpu_weights(vit, seed=11, dtype=torch.bfloat16, device=dev)
        SAMViTAdapter(vit).inject().set_context("hq_sam", {"early_vit_embedding": None})
        fast = CompiledSAMViT(vit, use_graph=False)
        with torch.no_grad():
            fast(torch.rand(1, 3, 1024, 1024, device=dev).to(torch.bfloat16))
    ops = fast.low.step
    whole = bench.time_ops(ops, iters=5) * 1e3
    groups, classes = {}, {}
    for e in ops:
        if e[0] is None:
            continue
        groups.setdefault(e[2], []).append(e)
        a = getattr(e[1][0], "_obj", None)
        if e[2].startswith("mi355x_gemm"):
            classes.setdefault(native.gemm_signature(a) + f":gelu{int(a.geglu)}", []).append(e)
        elif e[2] == "mi355x_groupnorm":
            classes.setdefault(f"groupnorm:B{a.B}:HW{a.HW}:C{a.C}" + (":cs" if a.colstats else ""), []).append(e)
        elif e[2] == "mi355x_attention_general":
            classes.setdefault(f"attn_general:B{a.B}:H{a.H}:Lq{a.Lq}:Lk{a.Lk}:Dqk{a.Dqk}:Dv{a.Dv}", []).append(e)
    print(f"whole program: {len(ops)} entries, {whole:.3f} ms per replay (no graph)")
    for name, sub in sorted(groups.items(), key=lambda kv: -len(kv[1])):
        ms = bench.time_ops(sub, iters=3) * 1e3
        print(f"  {name:28s} {len(sub):4d} launches {ms:7.3f} ms  {ms / len(sub) * 1e3:7.1f} us each")
    for name, sub in sorted(classes.items(), key=lambda kv: -len(kv[1])):
        ms = bench.time_ops(sub, iters=3) * 1e3
        fl = sum(bench.op_flops(e) for e in sub)
        print(f"    {name:60s} {len(sub):4d} launches {ms:7.3f} ms  {ms / len(sub) * 1e3:7.1f} us each" + (f"  {fl / ms / 1e9:6.0f} TF" if fl else ""))


if __name__ == "__main__":
    main()
