"""One UNet step of a workload lowered twice in one process -- with the measured tile table and with the library heuristic only -- on the same weights and inputs:
the two outputs may differ by summation order (other tiles), not by more.  A wrong tile configuration behind a table entry shows up here as an O(1) difference.

    python tools/check_tables_agree.py [--workload bare|lora_ip|control] [--images 4] [--lora-mode fused]"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402
from refiners_amd import native  # noqa: E402
from refiners_amd.engine import tuning  # noqa: E402
from refiners_amd.engine.compiled import CompiledSDXL  # noqa: E402


def other_engine(workload: str, dev: torch.device) -> None:
    if workload == "vae":
        from refiners_amd.engine.vae import CompiledVAEDecoder
        from refiners_amd.latent_diffusion.vae import SDXLAutoencoder

        model = SDXLAutoencoder(device="meta")
        bench.gpu_weights(model, seed=7, dtype=torch.bfloat16, device=dev)
        x = (torch.randn(1, 4, 128, 128, device=dev) * 0.13).to(torch.bfloat16)
        make = lambda: CompiledVAEDecoder(model)  # noqa: E731
    else:
        from refiners_amd.engine.sam import CompiledSAMViT
        from refiners_amd.segment_anything import SAMViTAdapter, SAMViTH

        model = SAMViTH(device="meta")
        bench.gpu_weights(model, seed=11, dtype=torch.bfloat16, device=dev)
        SAMViTAdapter(model).inject().set_context("hq_sam", {"early_vit_embedding": None})
        x = torch.rand(1, 3, 1024, 1024, device=dev).to(torch.bfloat16)
        make = lambda: CompiledSAMViT(model, use_graph=False)  # noqa: E731
    outs = {}
    for name, on in (("table", True), ("heuristic", False)):
        tuning.enabled = on
        tuning._table = None
        eng = make()
        with torch.no_grad():
            outs[name] = eng(x).float().clone()
        tiles = {}
        for e in eng.low.step:
            if e[0] is not None and e[2].startswith("mi355x_gemm"):
                t = int(e[1][0]._obj.tile)
                tiles[t] = tiles.get(t, 0) + 1
        print(f"{name}: launches per tile id {dict(sorted(tiles.items()))}", flush=True)
    a, b = outs["table"], outs["heuristic"]
    rel = float((a - b).norm() / b.norm())
    print(f"{workload}, bf16: rel l2 between the two lowerings {rel:.3e}, finite {bool(torch.isfinite(a).all())}", flush=True)
    assert rel < 2e-2, rel


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="bare")
    ap.add_argument("--images", type=int, default=4)
    ap.add_argument("--lora-mode", default="fused")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    native.load()
    if args.workload in ("vae", "sam"):  # the programs beside the step whose classes the table also holds (other_workloads): same check, their own engines
        other_engine(args.workload, dev)
        return
    unet, specs, bare_sd, pipe0, _ = bench.build_pipeline(args.workload, args.images, 0, dev, torch.bfloat16, args.lora_mode, use_graph=False, broadcast=False)
    x0 = pipe0.x.clone()
    outs = {}
    for name, on in (("table", True), ("heuristic", False)):
        tuning.enabled = on
        tuning._table = None
        p = CompiledSDXL(unet, num_inference_steps=50, condition_scale=5.0, use_graph=False, lora_mode=args.lora_mode)
        p.inputs, p.x = pipe0.inputs, x0.clone()  # (the way tools/ab_step.py shares one set of inputs between lowerings)
        p._tables(dev)
        p.step(0)
        torch.cuda.synchronize()
        outs[name] = p.x.float().clone()
        tiles = {}
        for e in p.engine.low.step:
            if e[0] is not None and e[2].startswith("mi355x_gemm"):
                t = int(e[1][0]._obj.tile)
                tiles[t] = tiles.get(t, 0) + 1
        print(f"{name}: launches per tile id {dict(sorted(tiles.items()))}", flush=True)
    a, b = outs["table"], outs["heuristic"]
    rel = float((a - b).norm() / b.norm())
    print(f"{args.workload} x {args.images} images, bf16: rel l2 between the two lowerings {rel:.3e}, finite {bool(torch.isfinite(a).all())}", flush=True)
    assert rel < 2e-2, rel


if __name__ == "__main__":
    main()
