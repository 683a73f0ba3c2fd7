"""In-place tile tuner for mi355x_gemm on the recorded SDXL step (writes refiners_amd/engine/tuning_gfx950.json).

For every shape class of GEMM / conv launches in the step program (native.gemm_signature), every legal (tile, LDS depth)
is tried by switching THAT class only and replaying the WHOLE step (so cold weights, the prefetch of the next launch's
weights and what neighbours leave in the caches are all in the measurement); the best one is kept before the next class
is tried (greedy, classes in order of their share of the step).  Small classes are timed on their own.

    python tools/autotune.py [--workload lora_ip|bare|control] [--images 1] [--out <json>] [--merge]
"""
from __future__ import annotations

import argparse
import json
import statistics
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402
from refiners_amd import native  # noqa: E402
from refiners_amd.engine import tuning  # noqa: E402


def step_ms(ops, iters: int = 4, reps: int = 3) -> float:
    """Median of `reps` measurements of `iters` replays of the whole program (ms per replay)."""
    vals = []
    for _ in range(reps):
        vals.append(bench.time_ops(ops, iters=iters) * 1e3)
    return statistics.median(vals)


def candidates(a, only=None) -> list[tuple[int, int]]:
    out = []
    if not a.lora_b and not (a.out_t and a.nt_begin % 256):  # the 8-wave / eight-phase loop: 7 = whole 256 x 256 tiles, 8 = stream-K, 9 = whole 192 x 256 tiles
        out += [(7, 0), (8, 0)] + ([] if a.out_t else [(9, 0)])
    elif a.lora_b and not a.conv and a.lora_groups == 1 and a.nseg == 1 and not a.out_t:  # its in-launch LoRA: one column group of a plain GEMM, whole tiles
        out += [(7, 0), (9, 0)]
    if a.ksplit > 1:  # a launch the lowering split along K: only the 8-wave loop (which takes the whole K) is an alternative
        return [c for c in out if only is None or c[0] in only]
    for tile in (1, 2, 3, 4, 6):  # 128x128, 128x64, 64x128, 64x64 (4 waves); 6 = 128x128 with two K groups (8 waves)
        if a.geglu == 1 and tile in (2, 4):
            continue
        if a.lora_b and tile == 6:  # in-launch LoRA exists for the 4-wave tiles only, with two LDS stages
            continue
        if a.lora_b and a.lora_r > 64 and tile in (2, 4):  # a stacked rank above 64 needs the 128-column tiles
            continue
        for st in ((2,) if tile == 6 or a.lora_b else (2, 3, 4)):
            out.append((tile, st))
    return [c for c in out if only is None or c[0] in only]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="lora_ip")
    ap.add_argument("--images", type=int, default=1)
    ap.add_argument("--lora-mode", default="merged")
    ap.add_argument("--out", default=str(tuning.TABLE_PATH))
    ap.add_argument("--merge", action="store_true", help="keep the entries of an existing table for shapes this run does not see")
    ap.add_argument("--min-share", type=float, default=0.004, help="classes below this share of the step are tuned on their own replay")
    ap.add_argument("--budget-s", type=float, default=420.0)
    ap.add_argument("--tiles", default="", help="comma-separated tile ids to try (default: all legal ones)")
    ap.add_argument("--only-class", default="", help="tune only the classes whose signature contains this (e.g. lora)")
    args = ap.parse_args()

    only = {int(t) for t in args.tiles.split(",")} if args.tiles else None
    tuning.enabled = False  # start from the library heuristic
    dev = torch.device("cuda", 0)
    native.load()
    t_start = time.time()
    if args.workload == "sam":  # BASELINE configs[4]: the SAM ViT-H image encoder + HQ-SAM hook on one 1024 x 1024 image (its launches carry their own shape classes)
        from refiners_amd.engine.sam import CompiledSAMViT
        from refiners_amd.segment_anything import SAMViTAdapter, SAMViTH

        vit = SAMViTH(device="meta")
        bench.gpu_weights(vit, seed=11, dtype=torch.bfloat16, device=dev)
        ad = SAMViTAdapter(vit).inject()
        ad.set_context("hq_sam", {"early_vit_embedding": None})
        fast = CompiledSAMViT(vit, use_graph=False)
        with torch.no_grad():
            fast(torch.rand(1, 3, 1024, 1024, device=dev).to(torch.bfloat16))
        low = fast.low
    elif args.workload == "vae":  # next-1 of SURVEY.md section 8(f): the SDXL VAE decoder on one 128 x 128 latent -> 1024 x 1024 image
        from refiners_amd.engine.vae import CompiledVAEDecoder
        from refiners_amd.latent_diffusion.vae import SDXLAutoencoder

        vae = SDXLAutoencoder(device="meta")
        bench.gpu_weights(vae, seed=7, dtype=torch.bfloat16, device=dev)
        dec = CompiledVAEDecoder(vae)
        with torch.no_grad():
            dec(torch.randn(1, 4, 128, 128, device=dev).to(torch.bfloat16) * 0.13)
        low = dec.low
    else:
        unet, specs, bare_sd, pipe, _ = bench.build_pipeline(args.workload, args.images, 0, dev, torch.bfloat16, args.lora_mode, use_graph=False, broadcast=False)
        pipe.step(0)
        low = pipe.engine.low
    torch.cuda.synchronize()
    ops = low.step
    classes: dict[str, list] = {}
    for e in ops:
        if e[0] is not None and e[2].startswith("mi355x_gemm"):
            a = e[1][0]._obj
            classes.setdefault(native.gemm_signature(a), []).append(a)
    base = step_ms(ops)
    print(f"step program: {len(ops)} entries, {sum(len(v) for v in classes.values())} GEMM/conv launches in {len(classes)} classes; baseline {base:.3f} ms", flush=True)
    # each class's own replay time, to order them
    own = {}
    for sig, items in classes.items():
        sub = [e for e in ops if e[0] is not None and e[2].startswith("mi355x_gemm") and native.gemm_signature(e[1][0]._obj) == sig]
        own[sig] = bench.time_ops(sub, iters=3) * 1e3
    order = sorted(classes, key=lambda s: -own[s])
    choices: dict[str, list[int]] = {}
    log = []
    cur = base
    for sig in order:
        if time.time() - t_start > args.budget_s:
            print("time budget reached, stopping", flush=True)
            break
        if args.only_class and args.only_class not in sig:
            continue
        items = classes[sig]
        a0 = items[0]
        whole = own[sig] / base >= args.min_share
        sub = None if whole else [e for e in ops if e[0] is not None and e[2].startswith("mi355x_gemm") and native.gemm_signature(e[1][0]._obj) == sig]
        if not whole and own[sig] < 0.02:
            continue
        start = (int(a0.tile), int(a0.stages))
        ks0 = int(a0.ksplit)
        results = {}

        def apply(tile, st):
            for a in items:
                a.tile, a.stages = tile, st
                a.ksplit = 1 if tile in (7, 8, 9) else ks0
                if tile == 8:
                    native.attach_streamk(a, low._sk)

        for tile, st in [(0, 0)] + candidates(a0, only):
            if ks0 > 1 and tile == 0:
                tile, st = start  # conv launches the lowering split along K carry an explicit tile
            apply(tile, st)
            try:
                t = step_ms(ops, iters=3, reps=3) if whole else bench.time_ops(sub, iters=5) * 1e3
            except Exception as exc:  # noqa: BLE001 -- a refused configuration
                t = float("inf")
                print(f"   {sig}: tile {tile} stages {st} refused: {exc}")
            results[(tile, st)] = t
        ref_key = start if ks0 > 1 else (0, 0)
        best = min(results, key=results.get)
        gain = results[ref_key] - results[best]
        thresh = 0.0015 * base if whole else 0.03 * results[ref_key]
        if best != ref_key and gain > thresh:
            # confirm against the reference once more (noise guard)
            apply(*ref_key)
            t_ref = step_ms(ops, iters=3, reps=3) if whole else bench.time_ops(sub, iters=5) * 1e3
            apply(*best)
            t_best = step_ms(ops, iters=3, reps=3) if whole else bench.time_ops(sub, iters=5) * 1e3
            if t_ref - t_best > thresh * 0.5:
                choices[sig] = [best[0], best[1]]
                cur = t_best if whole else cur
            else:
                best = ref_key
        else:
            best = ref_key
        apply(*best)
        row = {"class": sig, "launches": len(items), "own_ms": round(own[sig], 4), "mode": "whole-step" if whole else "class-only", "kept": list(best),
               "ms": {f"{k[0]}/{k[1]}": round(v, 4) for k, v in sorted(results.items(), key=lambda kv: kv[1])[:6]}}
        log.append(row)
        print(json.dumps(row), flush=True)
    final = step_ms(ops)
    print(f"baseline {base:.3f} ms -> tuned {final:.3f} ms ({len(choices)} classes changed)", flush=True)
    table = {}
    if args.merge and Path(args.out).exists():
        table = json.loads(Path(args.out).read_text()).get("choices", {})
    table.update(choices)
    out = {"device": native.device_info(), "how": "tools/autotune.py: greedy per-class search, whole recorded SDXL step replayed per candidate (bf16, weight prefetch on)",
           "workload": args.workload, "images_per_gpu": args.images, "baseline_ms": round(base, 4), "tuned_ms": round(final, 4), "choices": table, "log": log}
    Path(args.out).write_text(json.dumps(out, indent=1))
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / f"autotune_log_{args.lora_mode}_{args.workload}_{args.images}.json").write_text(json.dumps(out, indent=1))
    (ROOT / "gpurun_out" / "tuning_gfx950.json").write_text(json.dumps({k: out[k] for k in ("device", "how", "choices")}, indent=1))


if __name__ == "__main__":
    main()
