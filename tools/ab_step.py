"""A/B of the lowering switches on the SDXL step, one process, one set of weights: step time (HIP graph) per variant and the
per-family replay of the last one.  Variants are given as name=ENV1:val,ENV2:val ...; flags read at lowering time
(REFINERS_AMD_LN_FUSE, REFINERS_AMD_QKV_MERGE, REFINERS_AMD_TUNING, REFINERS_AMD_KBLOCK, REFINERS_AMD_WEIGHT_PREFETCH,
REFINERS_AMD_TIME_BATCH, REFINERS_AMD_ATTN_PIPE, REFINERS_AMD_GN_STATS, REFINERS_AMD_TIME_TABLE, REFINERS_AMD_CAT_FUSE).

    python tools/ab_step.py --workload lora_ip base=REFINERS_AMD_LN_FUSE:0,REFINERS_AMD_QKV_MERGE:0,REFINERS_AMD_TUNING:0 all=
"""
from __future__ import annotations

import argparse
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402
from refiners_amd import native  # noqa: E402
from refiners_amd.engine import tuning  # noqa: E402
from refiners_amd.engine.compiled import CompiledSDXL  # noqa: E402

KEYS = ("REFINERS_AMD_LN_FUSE", "REFINERS_AMD_QKV_MERGE", "REFINERS_AMD_TUNING", "REFINERS_AMD_KBLOCK", "REFINERS_AMD_WEIGHT_PREFETCH", "REFINERS_AMD_TIME_BATCH", "REFINERS_AMD_ATTN_PIPE",
        "REFINERS_AMD_GN_STATS", "REFINERS_AMD_TIME_TABLE", "REFINERS_AMD_CAT_FUSE", "REFINERS_AMD_TUNING_TABLE", "REFINERS_AMD_PF_BLOCKS", "REFINERS_AMD_LORA_G8", "REFINERS_AMD_CFG_SPLIT", "REFINERS_AMD_CFG_SPLIT_LEAD")


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="lora_ip")
    ap.add_argument("--images", type=int, default=1)
    ap.add_argument("--lora-mode", default="merged")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--lib", default=None, help="another build of libmi355x_refiners.so to run instead of the in-tree one (compile-time A/B across two processes)")
    ap.add_argument("variants", nargs="*")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    native.load(args.lib)
    unet, specs, bare_sd, pipe0, _ = bench.build_pipeline(args.workload, args.images, 0, dev, torch.bfloat16, args.lora_mode, use_graph=True, broadcast=False)
    inputs, x0 = pipe0.inputs, pipe0.x.clone()
    del pipe0
    variants = [v.split("=", 1) for v in (args.variants or ["default="])]
    pipes = {}
    after2: dict = {}
    for name, envs in variants:
        for k in KEYS:
            os.environ.pop(k, None)
        for kv in filter(None, envs.split(",")):
            k, v = kv.split(":")
            os.environ[k] = v
        native.attention_pipeline_from_env()
        tuning.enabled = os.environ.get("REFINERS_AMD_TUNING", "1") != "0"
        tuning._table = None
        tuning.lora_g8 = os.environ.get("REFINERS_AMD_LORA_G8", "1") != "0"
        libtag = None
        if "%" in name:  # "name%prio=...": lowered against csrc/variants/libmi355x_refiners_prio.so (refiners_amd.build_native.build_variant)
            name_, libtag = name.split("%", 1)
            native.switch_library(ROOT / "refiners_amd" / "csrc" / "variants" / f"libmi355x_refiners_{libtag.split('@')[0]}.so")
        mode = name.split("@", 1)[1] if "@" in name else args.lora_mode  # "name@merged=..." / "name@fused=...": the variant's LoRA mode (same process, same weights)
        p = CompiledSDXL(unet, num_inference_steps=50, condition_scale=5.0, use_graph=True, lora_mode=mode)
        p.inputs, p.x = inputs, x0.clone()
        p._tables(dev)
        p.step(0)
        p.step(1)
        torch.cuda.synchronize()
        pipes[name] = p
        after2[name] = p.x.float().clone()
        first = after2[next(iter(after2))]
        print(f"{name}: {p.engine.stats['step_ops']} launches/step{' x 2 programs (split CFG pair)' if p.engine_c is not None else ''}, tuning {p.engine.stats.get('gemm_tuning')}, "
              f"x after 2 steps vs the first variant: rel l2 {float((after2[name] - first).norm() / first.norm()):.3e}", flush=True)
        if libtag is not None:
            native.switch_library(None)
    res = {n: [] for n in pipes}
    for _ in range(args.rounds):  # interleaved rounds
        for name, p in pipes.items():
            res[name].append(bench.timed_steps(p, args.steps, 2, 1, dev) / args.steps * 1e3)
    for name, vals in res.items():
        print(f"{name:24s} ms/step: " + "  ".join(f"{v:.3f}" for v in vals) + f"   min {min(vals):.3f}", flush=True)
    last = list(pipes)[-1]
    roof = bench.family_roofline(pipes[last], args.workload, args.images, min(res[last]))
    print(json.dumps({"variant": last, "ms_per_step": min(res[last]), "families": roof["families"], "step": roof["step"]}), flush=True)
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "ab_step.json").write_text(json.dumps({"ms": res, "families_of_last": roof["families"]}, indent=1))


if __name__ == "__main__":
    main()
