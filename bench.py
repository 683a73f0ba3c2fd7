"""bench.py -- SDXL-base denoising-step benchmark of the MI355X hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one DDIM denoising step of one 1024x1024 image per GPU: SDXL-base UNet forward on the classifier-free-
guidance pair (batch 2, 4x128x128 latents, 77 text tokens) + CFG combine + DDIM update, bfloat16, replayed as one HIP
graph (BASELINE.json configs[1]; `--workload lora_ip` runs configs[2]: two rank-16 LoRAs on all 722 transformer
Linears + IP-Adapter).  Weights are random-init of the real architecture, inputs synthetic, both resident in HBM before
the timed region.  Multi-GPU: one process per GPU, independent prompts per rank (weak scaling), ONE RCCL broadcast of
the weights at start-up, no per-step collective.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

STEP_TFLOP = {"bare": 13.522, "lora_ip": 13.895, "control": 19.563}  # algorithmic FLOPs of one CFG-pair UNet forward
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (/opt/skills/guides/MI355X_MICROARCH.md)
LATENT = (128, 128)


def gpu_weights(unet, seed: int, dtype: torch.dtype, device: torch.device) -> None:
    """Random-init every parameter directly in HBM (same per-kind scaling rule as refiners_amd.synth)."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for k, v in unet.state_dict().items():
        shape = tuple(v.shape)
        n = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        leaf, kind = k.split(".")[-2:]
        if "Norm" in leaf:
            t = 1 + 0.1 * n if kind == "weight" else 0.1 * n
        elif kind == "bias" or len(shape) < 2:
            t = 0.1 * n
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = n / fan_in ** 0.5
        sd[k] = t.to(dtype)
    unet.load_state_dict(sd, assign=True)


def op_flops(entry) -> float:
    """Algorithmic FLOPs of one recorded launch (GEMM / implicit-GEMM conv: 2 M N sum K; attention: 4 B H Lq Lk D)."""
    fn, args, what, _ = entry
    if fn is None:
        return 0.0
    a = getattr(args[0], "_obj", None)
    if what.startswith("mi355x_gemm"):
        k = 0
        for s in range(a.nseg):
            sg = a.seg[s]
            k += sg.k * (sg.ksize * sg.ksize if a.conv else 1)
        return 2.0 * a.M * a.N * k
    if what == "mi355x_attention":
        return sum(4.0 * a.B * a.H * a.Lq * a.kv[s].Lk * a.D for s in range(a.nstream))
    return 0.0


def pmc_traffic(family: str):
    """HBM-side bytes per launch of a kernel family from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE are
    separate profiler runs, they cannot be taken inside this process): FETCH_SIZE doubled per MI355X_MICROARCH.md, KiB -> B."""
    try:
        f = sorted((ROOT / "profiles").glob("r*_pmc_traffic.json"))[-1]  # the latest committed pass
        fam = json.loads(f.read_text())["families"][family]
        return {"fetch_bytes_per_launch": round(fam["FETCH_SIZE"]["bytes_per_launch"]), "write_bytes_per_launch": round(fam["WRITE_SIZE"]["bytes_per_launch"]),
                "source": f"profiles/{f.name}"}
    except Exception:  # noqa: BLE001 -- no committed PMC pass for this family
        return None


def time_ops(ops, iters: int = 5) -> float:
    """Seconds per replay of a list of recorded launches, HIP events on the launch stream."""
    from refiners_amd import native

    native.replay(ops)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        native.replay(ops)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["bare", "lora_ip", "control"], default="bare",
                    help="bare = BASELINE configs[1]; lora_ip = configs[2]; control = configs[3] (ControlLora canny; use --images-per-gpu 4 for its 32-prompt / 8-GPU shape)")
    ap.add_argument("--images-per-gpu", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the informational extras (VAE decode, 4-images-per-GPU point): for profiler passes")
    ap.add_argument("--lora-mode", choices=["fused", "merged"], default="merged",
                    help="merged: W' = W + sum s B A formed at lowering time (one launch per adapted layer); fused: run-time LoRA K segments")
    args = ap.parse_args()

    import refiners_amd
    from refiners_amd import native, parallel, synth
    from refiners_amd.engine.compiled import CompiledSDXL
    from refiners_amd.latent_diffusion.sdxl import SDXLUNet

    rank, world, local = parallel.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run --nproc-per-node {args.gpus})"
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the product path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    native.load()
    dtype = torch.bfloat16

    # ---- model: rank 0 draws the weights, everyone else receives them over RCCL / xGMI --------------------------------
    t0 = time.time()
    unet = SDXLUNet(4, device="meta")
    gpu_weights(unet, seed=0 if rank == 0 else 1000 + rank, dtype=dtype, device=dev)
    specs = {"loras": [], "ip": None, "control": []}
    if args.workload == "lora_ip":
        shapes = synth.model_shapes(unet)
        specs = {"loras": [synth.lora_spec(shapes, "l1", 1.0, seed=5), synth.lora_spec(shapes, "l2", 0.8, seed=5)],
                 "ip": synth.ip_spec(shapes, 0.6, batch=2 * args.images_per_gpu, seed=5), "control": []}
        synth.apply_adapters(unet, refiners_amd.namespace(), device=dev, dtype=dtype, **specs)
    if args.workload == "control":
        specs = {"loras": [], "ip": None, "control": [synth.control_spec("canny", 1.0, 2 * args.images_per_gpu, LATENT, seed=5)]}
        synth.apply_adapters(unet, refiners_amd.namespace(), device=dev, dtype=dtype, **specs)
    torch.cuda.synchronize()
    tb = time.time()
    n_bcast = parallel.broadcast_module(unet, src=0)
    torch.cuda.synchronize()
    bcast_s = time.time() - tb
    n_params = sum(p.numel() for p in unet.parameters())

    # ---- inputs: independent prompts per rank, resident in HBM -------------------------------------------------------
    n_img = args.images_per_gpu
    inp = synth.sdxl_inputs(n_img, LATENT, seed=100 + rank)
    pipe = CompiledSDXL(unet, num_inference_steps=50, condition_scale=5.0, use_graph=not args.no_graph, lora_mode=args.lora_mode)
    kw = {}
    if specs["ip"] is not None:
        kw["clip_image_embedding"] = specs["ip"]["tokens"].to(dev)
    if specs["control"]:
        kw["conditions"] = {c["name"]: c["condition"].to(dev) for c in specs["control"]}
    pipe.set_inputs(inp["x"].to(dev), clip_text_embedding=inp["text"].to(dev), pooled_text_embedding=inp["pooled"].to(dev),
                    time_ids=inp["time_ids"].to(dev), **kw)
    for i in range(args.warmup):
        pipe.step(i % 50)
    torch.cuda.synchronize()
    setup_s = time.time() - t0

    # ---- timed region: exactly K steps between barrier + synchronize on both sides --------------------------------
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(args.steps):
        pipe.step(i % 50)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t1, device=dev)
    finite = bool(torch.isfinite(pipe.x.float()).all())

    if rank != 0:
        if world > 1:
            torch.distributed.barrier()
        return

    ms_per_step = elapsed / args.steps * 1e3
    images_per_s = world * n_img / (ms_per_step * 1e-3 * 50)
    low = pipe.engine.low
    # ---- roofline of the dominant kernel family, measured live (outside the timed region) ---------------------------
    groups: dict[str, list] = {}
    for e in low.step:
        if e[0] is not None:
            groups.setdefault(e[2].split("@")[0], []).append(e)  # launches on the side stream belong to the same family
    fam = {}
    for name, ops in groups.items():
        sec = time_ops(ops)
        fl = sum(op_flops(e) for e in ops)
        fam[name] = {"launches": len(ops), "ms": round(sec * 1e3, 4), "avg_us": round(sec / len(ops) * 1e6, 2), "tflop": round(fl / 1e12, 4),
                     "tflops": round(fl / sec / 1e12, 1) if fl else None}
    dom = max((n for n in fam if fam[n]["tflop"]), key=lambda n: fam[n]["ms"])
    executed_tflop = sum(f["tflop"] for f in fam.values())
    roofline = {
        "bound": "mfma", "kernel": dom, "launches_per_step": fam[dom]["launches"], "avg_launch_us": fam[dom]["avg_us"],
        "achieved": fam[dom]["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(fam[dom]["tflops"] / PEAK_BF16_TFLOPS, 4),
        "traffic": pmc_traffic(dom),
        "step": {"algorithmic_tflop": STEP_TFLOP[args.workload] * n_img, "executed_tflop": round(executed_tflop, 3),
                 "achieved": round(STEP_TFLOP[args.workload] * n_img / (ms_per_step * 1e-3), 1),
                 "frac": round(STEP_TFLOP[args.workload] * n_img / (ms_per_step * 1e-3) / PEAK_BF16_TFLOPS, 4)},
        "families": fam,
    }

    # ---- CPU baseline: the oracle (float32 port of the reference's algorithm) on this host's cores, one step -------
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import unet_oracle as O

        if specs["loras"] or specs["ip"]:
            bare = {}  # the oracle wants bare-model keys: adapters are passed as specs
            raise_keys = None
        sd_cpu = None
        if args.workload == "bare":
            sd_cpu = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
        if sd_cpu is not None:
            cin = synth.sdxl_inputs(1, LATENT, seed=100)
            O.sdxl_cfg_step(sd_cpu, cin["x"][:, :, :32, :32], 0, 50, cin["text"], cin["pooled"], cin["time_ids"])  # page-in / warm-up at 32x32
            tc = time.perf_counter()
            O.sdxl_cfg_step(sd_cpu, cin["x"], 0, 50, cin["text"], cin["pooled"], cin["time_ids"])
            cpu_s = time.perf_counter() - tc
            cpu = {"value": round(1.0 / (cpu_s * 50), 6), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                   "ms_per_step": round(cpu_s * 1e3, 1),
                   "sample": "1 of the 50 DDIM steps of one 1024x1024 image (CFG pair), float32, oracle/unet_oracle.py; images/s extrapolated x50"}
            del sd_cpu

    # ---- next-1 (outside the metric): VAE decode of the finished latents, for an end-to-end images/s figure ---------
    vae_ms = None
    try:
        if args.no_extra:
            raise RuntimeError("skipped (--no-extra)")
        from refiners_amd.engine.vae import CompiledVAEDecoder
        from refiners_amd.latent_diffusion.vae import SDXLAutoencoder

        vae = SDXLAutoencoder(device="meta")
        gpu_weights(vae, seed=7, dtype=dtype, device=dev)
        dec = CompiledVAEDecoder(vae)
        z = pipe.x[:1] * 0.13
        dec(z)
        torch.cuda.synchronize()
        tv = time.perf_counter()
        for _ in range(3):
            dec(z)
        torch.cuda.synchronize()
        vae_ms = (time.perf_counter() - tv) / 3 * 1e3
    except Exception as exc:  # noqa: BLE001 -- the VAE is outside the benchmarked path; report, do not fail the bench
        vae_ms = f"failed: {type(exc).__name__}: {exc}"

    # ---- throughput-oriented operating point (outside the metric): 4 images per GPU through the same engine -------------
    batched = None
    if world == 1 and n_img == 1 and args.workload == "bare" and not args.no_extra:
        try:
            inp4 = synth.sdxl_inputs(4, LATENT, seed=300)
            pipe4 = CompiledSDXL(unet, num_inference_steps=50, condition_scale=5.0, use_graph=not args.no_graph, lora_mode=args.lora_mode)
            pipe4.set_inputs(inp4["x"].to(dev), clip_text_embedding=inp4["text"].to(dev), pooled_text_embedding=inp4["pooled"].to(dev), time_ids=inp4["time_ids"].to(dev))
            for i in range(2):
                pipe4.step(i)
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            for i in range(10):
                pipe4.step(i)
            torch.cuda.synchronize()
            ms4 = (time.perf_counter() - t4) / 10 * 1e3
            batched = {"images_per_gpu": 4, "ms_per_step": round(ms4, 3), "images_per_s": round(4 / (ms4 * 1e-3 * 50), 4),
                       "step_tflops": round(4 * STEP_TFLOP["bare"] / (ms4 * 1e-3), 1), "frac_of_peak": round(4 * STEP_TFLOP["bare"] / (ms4 * 1e-3) / PEAK_BF16_TFLOPS, 4)}
            del pipe4
        except Exception as exc:  # noqa: BLE001
            batched = f"failed: {type(exc).__name__}: {exc}"

    line = {
        "metric": "sdxl_base_1024px_images_per_sec_50_ddim_steps", "value": round(images_per_s, 4), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "SDXL-base UNet CFG step, 1024x1024 (latent 2x4x128x128, 77 text tokens), DDIM-50" +
                   {"bare": "", "lora_ip": " + 2 LoRA r16 (722 Linears) + IP-Adapter", "control": " + ControlLora (canny)"}[args.workload],
                   "baseline_config": {"bare": "configs[1]", "lora_ip": "configs[2]", "control": "configs[3]"}[args.workload], "images_per_gpu": n_img,
                   "parallelism": f"replica x{world} (independent prompts, weights broadcast once)", "hip_graph": not args.no_graph,
                   "lora_mode": args.lora_mode if args.workload != "bare" else None},
        "step_latency_ms": round(ms_per_step, 3),
        "roofline": roofline, "cpu_baseline": cpu,
        "extra": {"params": n_params, "launches_per_step": pipe.engine.stats["step_ops"], "prologue_launches": pipe.engine.stats["prologue_ops"],
                  "fallback_nodes": pipe.engine.stats["fallback_nodes"], "arena_bytes": pipe.engine.stats["pool_bytes"],
                  "weight_prefetch": pipe.engine.stats.get("weight_prefetch"),
                  "weights_broadcast_s": round(bcast_s, 3), "broadcast_launches": n_bcast, "setup_s": round(setup_s, 1),
                  "output_finite": finite, "device": native.device_info(),
                  "throughput_operating_point": batched,
                  "vae_decode_ms_per_image": round(vae_ms, 2) if isinstance(vae_ms, float) else vae_ms,
                  "end_to_end_images_per_s_incl_vae": round(world * n_img / (ms_per_step * 1e-3 * 50 + n_img * vae_ms * 1e-3), 4) if isinstance(vae_ms, float) else None},
    }
    print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()


if __name__ == "__main__":
    main()
