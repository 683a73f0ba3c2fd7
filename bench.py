"""bench.py -- SDXL-base denoising-step benchmark of the MI355X hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one DDIM denoising step of one 1024x1024 image per GPU: SDXL-base UNet forward on the classifier-free-
guidance pair (batch 2, 4x128x128 latents, 77 text tokens) + CFG combine + DDIM update, bfloat16, replayed as one HIP
graph.  The default workload is the north star's TARGET, BASELINE.json configs[2]: two rank-16 LoRAs on all 722
transformer Linears + IP-Adapter injected through Adapter.inject(); `--workload bare` is configs[1] (also measured by the
default run and reported under `extra.configs1_bare`), `--workload control --images-per-gpu 4` configs[3]'s per-GPU shape.
Weights are random-init of the real architecture, inputs synthetic, both resident in HBM before the timed region.
Multi-GPU: one process per GPU, independent prompts per rank (weak scaling), ONE RCCL broadcast of the weights at
start-up, no per-step collective; `python bench.py --gpus N` without a torchrun environment re-launches itself through
torch.distributed.run.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import socket
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

STEP_TFLOP = {"bare": 13.522, "lora_ip": 13.895, "control": 19.563}  # algorithmic FLOPs of one CFG-pair UNet forward
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (/opt/skills/guides/MI355X_MICROARCH.md)
PEAK_HBM_GBPS = 8000.0     # HBM3E peak of the same guide (6.3 TB/s is what a float4 copy reaches)
LATENT = (128, 128)


_T0 = time.time()


def note(msg: str) -> None:
    """Progress on stderr (the one JSON line on stdout comes last: a leg that takes minutes should be findable in the log of a run that was cut short)."""
    print(f"[bench {time.time() - _T0:7.1f} s] {msg}", file=sys.stderr, flush=True)


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


def respawn_under_torchrun(n: int) -> None:
    """`python bench.py --gpus N` (N > 1) outside a torchrun environment: become `python -m torch.distributed.run ...` with
    the same arguments, one rank per GPU, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(Path(__file__).resolve()), *sys.argv[1:]]
    os.execv(sys.executable, cmd)


def pmc_mfma_file(family: str) -> Path:
    """The committed MFMA counter pass to quote: the LATEST one by name (deterministic; round 3 picked the least-perturbed of several passes,
    which the advisor rightly called best-of-N selection)."""
    return sorted((ROOT / "profiles").glob("r*_pmc_mfma.json"))[-1]


def pmc_mfma_util(family: str, family_tflop_per_step: float = 0.0):
    """MFMA utilisation of a kernel family from the latest committed counter pass (tools/profile_round.py: a separate
    `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ...` run of this same command)."""
    try:
        f = pmc_mfma_file(family)
        doc = json.loads(f.read_text())
        fam = doc["families"][family]
        out = {"mfma_util": round(fam.get("mfma_util_by_duration", fam["mfma_util"]), 4), "SQ_VALU_MFMA_BUSY_CYCLES": fam["SQ_VALU_MFMA_BUSY_CYCLES"],
               "definition": "fraction of the family's kernel time with the matrix pipes busy: SQ_VALU_MFMA_BUSY_CYCLES per ns of dispatch duration, anchored on a "
                             "calibration launch of known MFMA count and duration (see the file)",
               "scope": doc.get("scope", "whole process"), "source": f"profiles/{f.name}",
               "how": doc.get("how", "") + "  (counter passes launch the recorded program directly, --no-graph: rocprofv3 attributes counters per dispatch; the timed region replays "
                                           "the same launches as one HIP graph)"}
        reps = re.search(r"(\d+) full replay", doc.get("scope", ""))
        if reps and fam.get("DURATION_NS") and family_tflop_per_step:
            # boxes differ by several per cent: the FLOP-based fraction of THE SAME profiled run is what mfma_util has to agree with
            out["flop_frac_of_that_run"] = round(family_tflop_per_step * int(reps.group(1)) / (fam["DURATION_NS"] * 1e-9) / PEAK_BF16_TFLOPS, 4)
        if "mfma_util_by_duration" in fam:
            # the round-2 definition (busy / GRBM_GUI_ACTIVE): GUI-active also ticks through the profiler's per-dispatch counter start / stop,
            # which dilutes 20-40 us launches
            out["mfma_util_over_gui_active"] = round(fam["mfma_util"], 4)
        return out
    except Exception:  # noqa: BLE001 -- no committed pass
        return None


def program_entry(e) -> dict:
    """One launch of the recorded step program: entry point + shape class (the key the in-place profile aggregates on)."""
    from refiners_amd import native

    fn, args, what, _ = e
    a = getattr(args[0], "_obj", None)
    if what.startswith("mi355x_gemm"):
        return {"what": what, "key": native.gemm_signature(a) + (f":tile{a.tile}/{a.stages}" if a.tile else ""), "ksplit": int(a.ksplit)}
    if what == "mi355x_attention":
        return {"what": what, "key": f"attention:B{a.B}:H{a.H}:Lq{a.Lq}:Lk{a.kv[0].Lk}" + (f"+{a.kv[1].Lk}" if a.nstream > 1 else "")}
    if what == "mi355x_layernorm":
        return {"what": what, "key": f"layernorm:{a.M}x{a.C}"}
    if what == "mi355x_groupnorm":
        return {"what": what, "key": f"groupnorm:B{a.B}:HW{a.HW}:C{a.C}" + (":cs" if a.colstats else "")}
    return {"what": what, "key": what}


def pmc_traffic(family: str):
    """HBM-side bytes per launch of a kernel family from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE are
    separate profiler runs, they cannot be taken inside this process): FETCH_SIZE doubled per MI355X_MICROARCH.md, KiB -> B."""
    try:
        f = sorted((ROOT / "profiles").glob("r*_pmc_traffic.json"))[-1]  # the latest committed pass
        doc = json.loads(f.read_text())
        fam = doc["families"][family]
        return {"fetch_bytes_per_launch": round(fam["FETCH_SIZE"]["bytes_per_launch"]), "write_bytes_per_launch": round(fam["WRITE_SIZE"]["bytes_per_launch"]),
                "scope": doc.get("scope", "whole process"), "source": f"profiles/{f.name}"}
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


WORKLOADS = {
    "bare": ("configs[1]", ""),
    "lora_ip": ("configs[2]", " + 2 LoRA r16 (722 Linears) + IP-Adapter"),
    "control": ("configs[3]", " + ControlLora (canny)"),
}


def stage_inputs(pipe, specs: dict, n_img: int, seed: int, dev: torch.device) -> None:
    from refiners_amd import synth

    inp = synth.sdxl_inputs(n_img, LATENT, seed=seed)
    kw = {}
    if specs["ip"] is not None:
        kw["clip_image_embedding"] = specs["ip"]["tokens"].to(dev)
    if specs["control"]:
        kw["conditions"] = {c["name"]: c["condition"].to(dev) for c in specs["control"]}
    pipe.set_inputs(inp["x"].to(dev), clip_text_embedding=inp["text"].to(dev), pooled_text_embedding=inp["pooled"].to(dev),
                    time_ids=inp["time_ids"].to(dev), **kw)


def replica_check(pipe, specs: dict, n_img: int, dev: torch.device) -> dict:
    """N > 1 only, after the timed region: every rank runs ONE step on the SAME inputs and the outputs are compared across ranks -- the
    replicas hold broadcast weights and broadcast packed weights (K-blocked / merged / folded on rank 0 only), so a hand-over bug shows up
    here as a rank whose step differs.  The kernels are deterministic (fixed-order split-K), so equal GPUs give equal bits; the bar is 1e-3 relative."""
    stage_inputs(pipe, specs, n_img, 100, dev)
    pipe.step(0)
    x = pipe.x.double()
    mine = torch.stack([x.sum(), x.abs().sum(), x.square().sum()]).to(dev)
    from refiners_amd import parallel

    world = torch.distributed.get_world_size()
    got = parallel.all_gather(mine)
    ref = got[0]
    dev_max = max(float(((g - ref).abs() / ref.abs().clamp_min(1e-30)).max()) for g in got)
    ok = bool(dev_max < 1e-3)
    if not ok:  # loud, but after the timed region and without losing the measured line: the JSON carries ok = false and the checksums
        print(f"bench.py: REPLICAS DISAGREE on the same inputs: checksums {[g.tolist() for g in got]}", file=sys.stderr, flush=True)
    return {"ok": ok, "max_rel_checksum_deviation": dev_max, "ranks": world, **({} if ok else {"checksums": [g.tolist() for g in got]})}


def build_pipeline(workload: str, n_img: int, rank: int, dev: torch.device, dtype: torch.dtype, lora_mode: str, use_graph: bool, broadcast: bool = True, packs: str = "broadcast"):
    """UNet (random init in HBM) + adapters injected through the Chain API + one CompiledSDXL with its inputs staged."""
    import refiners_amd
    from refiners_amd import parallel, synth
    from refiners_amd.engine.compiled import CompiledSDXL
    from refiners_amd.latent_diffusion.sdxl import SDXLUNet

    unet = SDXLUNet(4, device="meta")
    gpu_weights(unet, seed=0 if rank == 0 else 1000 + rank, dtype=dtype, device=dev)  # rank 0 draws; the others receive (below)
    bare_sd = dict(unet.state_dict())  # references to the bare model's tensors, for the CPU baseline
    specs = {"loras": [], "ip": None, "control": []}
    shapes = synth.model_shapes(unet)
    if workload == "lora_ip":
        specs = {"loras": [synth.lora_spec(shapes, "l1", 1.0, seed=5), synth.lora_spec(shapes, "l2", 0.8, seed=5)],
                 "ip": synth.ip_spec(shapes, 0.6, batch=2 * n_img, seed=5), "control": []}
    if workload == "control":
        specs = {"loras": [], "ip": None, "control": [synth.control_spec("canny", 1.0, 2 * n_img, LATENT, seed=5)]}
    if workload != "bare":
        synth.apply_adapters(unet, refiners_amd.namespace(), device=dev, dtype=dtype, **specs)
    multi = broadcast and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
    torch.cuda.synchronize()
    if multi:
        torch.distributed.barrier()  # the clock below times the collective, not the other ranks' weight drawing
    b0 = parallel.moved["bytes"]
    tb = time.time()
    n_bcast = parallel.broadcast_module(unet, src=0) if broadcast else 0
    torch.cuda.synchronize()
    bcast_s = time.time() - tb
    wbytes = parallel.moved["bytes"] - b0
    pipe = CompiledSDXL(unet, num_inference_steps=50, condition_scale=5.0, use_graph=use_graph, lora_mode=lora_mode)
    stage_inputs(pipe, specs, n_img, 100 + rank, dev)
    info = {"weights_broadcast_s": round(bcast_s, 3), "broadcast_launches": n_bcast, "weights_broadcast_bytes": wbytes,
            "weights_broadcast_gbps": round(wbytes / bcast_s / 1e9, 2) if wbytes and bcast_s > 0 else None}
    if multi:
        # every rank lowers its own program (it holds local addresses).  --packs broadcast (default): only rank 0 K-blocks / merges /
        # LayerNorm-folds the weights, the packed copies then travel over xGMI like the weights did (refiners_amd.parallel.broadcast_packs);
        # --packs local: every rank packs for itself (the A/B for the driver's 8-GPU run: ~11 GB over the links against seconds of torch work per rank)
        torch.cuda.synchronize()
        torch.distributed.barrier()
        b1 = parallel.moved["bytes"]
        tp = time.time()
        if packs == "broadcast":
            n_pk = parallel.broadcast_packs(pipe.lower_now, pipe.engine.cache, src=0)
        else:
            pipe.lower_now()
            n_pk = 0
        torch.cuda.synchronize()
        pk_s = time.time() - tp
        pbytes = parallel.moved["bytes"] - b1
        info.update(packs=packs, packs_broadcast_s=round(pk_s, 3), packs_broadcast_launches=n_pk, packs_broadcast_bytes=pbytes,
                    packs_broadcast_gbps=round(pbytes / pk_s / 1e9, 2) if pbytes and pk_s > 0 else None)
    return unet, specs, bare_sd, pipe, info


def timed_steps(pipe, steps: int, warmup: int, world: int, dev: torch.device) -> float:
    """W untimed steps, then exactly K steps between barrier + synchronize pairs; seconds, max over ranks."""
    from refiners_amd import parallel

    sync = torch.cuda.synchronize if torch.device(dev).type == "cuda" else (lambda: None)  # (CPU: the gloo dry run of tests/test_parallel_cpu.py)
    for i in range(warmup):
        pipe.step(i % 50)
    sync()
    if world > 1:
        torch.distributed.barrier()
    sync()
    t1 = time.perf_counter()
    for i in range(steps):
        pipe.step(i % 50)
    sync()
    if world > 1:
        torch.distributed.barrier()
    mine = time.perf_counter() - t1
    timed_steps.last_local_s = mine  # this rank's own clock (bench.py gathers them into extra.per_rank_ms_per_step)
    return parallel.max_over_ranks(mine, device=dev)


def family_roofline(pipe, workload: str, n_img: int, ms_per_step: float) -> dict:
    """Per-entry-point replay of the recorded step (HIP events on the launch stream, outside the timed region)."""
    low = pipe.engine.low
    groups: dict[str, list] = {}
    for e in low.step:
        if e[0] is not None:
            groups.setdefault(e[2], []).append(e)
    fam = {}
    for name, ops in groups.items():
        sec = time_ops(ops)
        fl = sum(op_flops(e) for e in ops)
        fam[name] = {"launches": len(ops), "ms": round(sec * 1e3, 4), "avg_us": round(sec / len(ops) * 1e6, 2), "tflop": round(fl / 1e12, 4),
                     "tflops": round(fl / sec / 1e12, 1) if fl else None}
    # GroupNorm (HBM-bound: partial sums, finalize, apply + SiLU): algorithmic bytes = two reads + one write of the tensor; one read + one write
    # where the statistics come from the epilogue of the launch that produced the tensor (mi355x_gemm_args.colstats_out)
    gn_ops = groups.get("mi355x_groupnorm", [])
    if gn_ops:
        gbytes = sum((2.0 if a.colstats else 3.0) * a.B * a.HW * a.C * (4 if a.dtype == 0 else 2) for a in (e[1][0]._obj for e in gn_ops))
        gsec = fam["mi355x_groupnorm"]["ms"] * 1e-3
        fam["mi355x_groupnorm"].update(algorithmic_gb=round(gbytes / 1e9, 4), hbm_gbps=round(gbytes / gsec / 1e9, 1), frac_of_hbm_peak=round(gbytes / gsec / 1e9 / PEAK_HBM_GBPS, 4),
                                       stats_from_producer=sum(1 for e in gn_ops if e[1][0]._obj.colstats))
    # the five shape classes that take the most time, each replayed on its own (entry point + shape = the key of profiles/*_inplace_by_shape.md)
    classes: dict[str, list] = {}
    for e in low.step:
        if e[0] is not None and op_flops(e):
            classes.setdefault(program_entry(e)["key"], []).append(e)
    cls = []
    for key, ops in classes.items():
        fl = sum(op_flops(e) for e in ops)
        cls.append((key, ops, fl))
    timed = []
    for key, ops, fl in sorted(cls, key=lambda c: -c[2])[:8]:  # by FLOPs first (cheap), then timed
        sec = time_ops(ops, iters=3)
        timed.append({"class": key, "launches": len(ops), "ms": round(sec * 1e3, 4), "avg_us": round(sec / len(ops) * 1e6, 2), "tflops": round(fl / sec / 1e12, 1),
                      "frac": round(fl / sec / 1e12 / PEAK_BF16_TFLOPS, 4)})
    top_classes = sorted(timed, key=lambda c: -c["ms"])[:5]
    try:  # the matrix-pipe utilisation of the same classes from the latest committed counter pass (step program only)
        pm = json.loads(pmc_mfma_file("mi355x_gemm").read_text()).get("classes", {})
        for c in top_classes:
            if c["class"] in pm:
                c["mfma_util_pmc"] = round(pm[c["class"]].get("mfma_util_by_duration", pm[c["class"]]["mfma_util"]), 4)
    except Exception:  # noqa: BLE001 -- no committed pass
        pass
    dom = max((n for n in fam if fam[n]["tflop"]), key=lambda n: fam[n]["ms"])
    executed_tflop = sum(f["tflop"] for f in fam.values())
    algo = STEP_TFLOP[workload] * n_img
    return {
        "bound": "mfma", "kernel": dom, "launches_per_step": fam[dom]["launches"], "avg_launch_us": fam[dom]["avg_us"],
        "achieved": fam[dom]["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(fam[dom]["tflops"] / PEAK_BF16_TFLOPS, 4),
        "traffic": pmc_traffic(dom), "mfma_util": pmc_mfma_util(dom, fam[dom]["tflop"]),
        "step": {"algorithmic_tflop": algo, "executed_tflop": round(executed_tflop, 3), "achieved": round(algo / (ms_per_step * 1e-3), 1),
                 "frac": round(algo / (ms_per_step * 1e-3) / PEAK_BF16_TFLOPS, 4)},
        "families": fam, "top_classes": top_classes,
    }


def reference_checkout():
    """refiners' own package for the CPU baseline leg: REFINERS_SRC, else the copy __graft_entry__.build() staged under oracle/_ref (it travels to
    the GPU box with the snapshot).  Test / baseline infrastructure only: the timed GPU region never imports it."""
    for cand in (os.environ.get("REFINERS_SRC"), ROOT / "oracle" / "_ref" / "src"):
        if cand and (Path(cand) / "refiners").exists():
            return Path(cand)
    return None


def _cpu_step_fn(api_kind: str, bare_sd: dict, specs: dict, workload: str, ref_src=None):
    """(step(x, text, pooled, time_ids) -> x_next, description) on CPU float32: one denoising step the way `LatentDiffusionModel.forward` does it
    (foundationals/latent_diffusion/model.py:128-159): contexts set on the UNet, Chain forward on the CFG pair, CFG combine, DDIM.
    api_kind "reference": refiners' OWN SDXLUNet / adapters / DDIM; "mirror": refiners_amd.fluxion's unfused Chain forward (the same ATen calls)."""
    from types import SimpleNamespace

    from refiners_amd import synth

    cpu = torch.device("cpu")
    if api_kind == "reference":
        sys.path[:0] = [p for p in (str(ROOT / "oracle" / "shim"), str(ref_src)) if p not in sys.path]
        import refiners.fluxion.layers as rfl
        from refiners.fluxion.adapters.lora import Conv2dLora, LinearLora, LoraAdapter
        from refiners.foundationals.latent_diffusion.solvers import DDIM as RefDDIM
        from refiners.foundationals.latent_diffusion.stable_diffusion_xl.control_lora import ConditionEncoder, ControlLoraAdapter, ZeroConvolution
        from refiners.foundationals.latent_diffusion.stable_diffusion_xl.image_prompt import SDXLIPAdapter
        from refiners.foundationals.latent_diffusion.stable_diffusion_xl.unet import SDXLUNet as RefUNet

        assert Path(rfl.__file__).resolve().is_relative_to(Path(ref_src).resolve())
        api = SimpleNamespace(fl=rfl, LinearLora=LinearLora, Conv2dLora=Conv2dLora, LoraAdapter=LoraAdapter, SDXLIPAdapter=SDXLIPAdapter,
                              ControlLoraAdapter=ControlLoraAdapter, ConditionEncoder=ConditionEncoder, ZeroConvolution=ZeroConvolution)
        unet, solver = RefUNet(4, device="meta"), RefDDIM(num_inference_steps=50)
        desc = "finegrain-ai/refiners itself (oracle/_ref): refiners.foundationals SDXLUNet Chain forward, adapters injected through its own API, its own DDIM"
    else:
        import refiners_amd
        from refiners_amd.latent_diffusion.sampling import DDIM
        from refiners_amd.latent_diffusion.sdxl import SDXLUNet

        api, unet, solver = refiners_amd.namespace(), SDXLUNet(4, device="meta"), DDIM(50, device=cpu, dtype=torch.float32)
        desc = "mirror-of-reference ATen path: refiners_amd.fluxion Chain forward, unfused, the same adapters injected (no oracle, no native kernels)"
    unet.load_state_dict({k: v.detach().to(device=cpu, dtype=torch.float32) for k, v in bare_sd.items()}, assign=True)
    if workload != "bare":
        synth.apply_adapters(unet, api, device=cpu, dtype=torch.float32, **specs)

    def step(x, text, pooled, time_ids):
        unet.set_timestep(solver.timesteps[0].unsqueeze(0))
        unet.set_clip_text_embedding(text)
        unet.set_pooled_text_embedding(pooled)
        unet.set_time_ids(time_ids)
        u, c = unet(torch.cat((x, x))).chunk(2)
        return solver(x, predicted_noise=u + 5.0 * (c - u), step=0)

    return step, desc


def physical_cores() -> int:
    """Physical cores of this host (unique (package, core) pairs of /proc/cpuinfo); os.cpu_count() when that cannot be read."""
    try:
        pairs, pkg = set(), "0"
        for ln in Path("/proc/cpuinfo").read_text().splitlines():
            if ln.startswith("physical id"):
                pkg = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                pairs.add((pkg, ln.split(":")[1].strip()))
        return len(pairs) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def usable_cpus() -> int:
    """CPUs this PROCESS may run on: the scheduler affinity mask, cut by the cgroup's CPU quota where one is set (a container on a 256-thread host may own a
    fraction of it; intra-op threads beyond that only queue behind each other)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline_step(bare_sd: dict, specs: dict, workload: str, threads: int, budget_s: float = 75.0) -> dict:
    """The benchmarked workload's denoising step on this host's cores, float32.  With refiners' own package at hand (reference_checkout) the timed
    step runs refiners ITSELF (`kind: "reference"`); the mirror's unfused Chain forward -- the stand-in of earlier rounds, and the fallback when no
    checkout is present (`kind: "port"`) -- is timed beside it on a 32x32-latent sample of the same step, where both are also compared value for value.
    Threads: `threads` > 0 as given; 0 = the best of a sweep over {physical / 4, physical / 2, physical, logical} intra-op threads on a 64x64-latent
    sample of the same step (BASELINE.md section 4 asks for the host's physical cores; on a 2-socket 256-thread host the round-5 guess of 64 was slower
    than the survey's 8-core box).  Timing: one warm-up (16x16 latents: page-in, thread pool) + up to 3 timed full-size steps, median -- the second and
    third only while the time spent stays under `budget_s` (the default run must finish within minutes; `--cpu-budget 600` times all three anywhere)."""
    from refiners_amd import synth

    phys, logical = physical_cores(), os.cpu_count() or 1
    usable = usable_cpus()
    ref_src = reference_checkout() if workload in ("bare", "lora_ip", "control") else None
    if workload == "control" and specs["control"] and specs["control"][0]["condition"].shape[0] != 2:
        ref_src = None  # the bounded sample below is one image
    kind = "reference" if ref_src is not None else "port"
    step, desc = _cpu_step_fn("reference" if ref_src is not None else "mirror", bare_sd, specs, workload, ref_src)
    cin = synth.sdxl_inputs(1, LATENT, seed=100)
    small = [cin["x"][:, :, :32, :32], cin["text"], cin["pooled"], cin["time_ids"]]
    mid = [cin["x"][:, :, :64, :64], *small[1:]]
    note(f"cpu baseline: {kind}, {phys} physical / {logical} logical / {usable} usable cpus")
    with torch.no_grad():
        torch.set_num_threads(threads or min(phys, usable))
        step(cin["x"][:, :, :16, :16], *small[1:])  # page-in / thread-pool warm-up on a 16x16 latent
        note("cpu baseline: warm-up done")
        sweep = None
        if not threads:
            sweep = {}
            for n in sorted({min(usable, n) for n in (max(1, phys // 4), max(1, phys // 2), phys, logical)} | {max(1, min(usable, 16))}):
                torch.set_num_threads(n)
                step(cin["x"][:, :, :16, :16], *small[1:])  # (the pool is rebuilt at the new size)
                ts = time.perf_counter()
                step(*mid)
                sweep[n] = round(time.perf_counter() - ts, 3)
                note(f"cpu baseline: sweep {n} threads: {sweep[n]} s on 64x64 latents")
            threads = min(sweep, key=sweep.get)
            torch.set_num_threads(threads)
            step(cin["x"][:, :, :16, :16], *small[1:])
        times, spent = [], 0.0
        for _ in range(3):
            tc = time.perf_counter()
            out = step(cin["x"], *small[1:])
            times.append(time.perf_counter() - tc)
            spent += times[-1]
            note(f"cpu baseline: full-size step {len(times)}: {times[-1]:.1f} s at {threads} threads")
            if spent + times[-1] > budget_s:
                break
        cpu_s = sorted(times)[len(times) // 2]
        assert bool(torch.isfinite(out).all())
        res = {"value": round(1.0 / (cpu_s * 50), 6), "unit": "images/s", "cores": threads, "kind": kind, "path": desc, "dtype": "f32",
               "ms_per_step": round(cpu_s * 1e3, 1), "timed_steps_s": [round(t, 2) for t in times], "host_cpus": logical, "physical_cores": phys, "usable_cpus": usable,
               "thread_sweep_s_on_64x64_latents": sweep,
               "sample": f"1 warm-up + {len(times)} timed step(s) (median) of the 50 DDIM steps of one 1024x1024 image (CFG pair) of this workload, intra-op threads = "
                         f"{'the best of the sweep' if sweep else 'as given'}; images/s extrapolated x50"}
        if kind == "reference":
            try:  # the mirror beside it, on a bounded sample (32x32 latents): same step, same weights, same adapters
                t1 = time.perf_counter()
                y_ref = step(*small)
                ref_small = time.perf_counter() - t1
                mstep, mdesc = _cpu_step_fn("mirror", bare_sd, specs, workload)
                mstep(cin["x"][:, :, :16, :16], *small[1:])
                t2 = time.perf_counter()
                y_mir = mstep(*small)
                mir_small = time.perf_counter() - t2
                res["mirror"] = {"path": mdesc, "sample": "the same step on 32x32 latents", "reference_ms": round(ref_small * 1e3, 1), "mirror_ms": round(mir_small * 1e3, 1),
                                 "mirror_over_reference": round(mir_small / ref_small, 3), "rel_l2_mirror_vs_reference": float((y_mir - y_ref).norm() / y_ref.norm())}
            except Exception as exc:  # noqa: BLE001
                res["mirror"] = f"failed: {type(exc).__name__}: {exc}"
    return res


def sd15_cpu_point(threads: int) -> dict:
    """BASELINE configs[0]: SD1.5 UNet single forward, 1x4x64x64 latent, float32, the reference's CPU Chain (no GPU, no adapters): refiners' OWN SD1UNet
    (stable_diffusion_1/unet.py:165-249) where its package is at hand (oracle/_ref), else the mirror's unfused Chain; 1 warm-up + 3 timed, median."""
    from refiners_amd import synth

    ref_src = reference_checkout()
    if ref_src is not None:
        sys.path[:0] = [p for p in (str(ROOT / "oracle" / "shim"), str(ref_src)) if p not in sys.path]
        from refiners.foundationals.latent_diffusion.stable_diffusion_1.unet import SD1UNet as U

        kind, desc = "reference", "finegrain-ai/refiners itself (oracle/_ref): SD1UNet Chain forward"
    else:
        from refiners_amd.latent_diffusion.sd1 import SD1UNet as U

        kind, desc = "port", "mirror-of-reference ATen path: refiners_amd.fluxion SD1UNet Chain forward, unfused"
    unet = U(4, device="meta")
    unet.load_state_dict(synth.synth_state_dict(synth.model_shapes(unet), seed=0), assign=True)
    x = torch.randn((1, 4, 64, 64), generator=synth._gen("in.x", 3))
    text = torch.randn((1, 77, 768), generator=synth._gen("in.text", 3))
    torch.set_num_threads(threads)
    times = []
    with torch.no_grad():
        for i in range(4):
            unet.set_timestep(torch.tensor([500]))
            unet.set_clip_text_embedding(text)
            tc = time.perf_counter()
            y = unet(x)
            if i:
                times.append(time.perf_counter() - tc)
        assert bool(torch.isfinite(y).all())
    med = sorted(times)[1]
    return {"workload": "SD1.5 UNet single forward, 1x4x64x64 latent, 77 text tokens, float32, CPU Chain, no adapters", "baseline_config": "configs[0]", "kind": kind, "path": desc,
            "s_per_forward": round(med, 3), "timed_s": [round(t, 3) for t in times], "cores": threads, "algorithmic_tflop": 0.803, "tflops": round(0.803 / med, 3),
            "sample": "1 warm-up + 3 timed forwards, median; synthetic weights (refiners_amd.synth, seed 0)"}


def parity_point(dev: torch.device) -> dict:
    """The benchmarked dtype's error, reported alongside: ONE step of configs[2] (2 LoRAs x 722 Linears + IP-Adapter, 128x128 latents, CFG
    pair, live LoRAs) in bfloat16 through the engine, against the same step as refiners ITSELF computed it on CPU in float32 -- the committed fixture
    tests/golden/full_size_reference.safetensors (recipe `lora_ip_step7`, written in the build container by oracle/make_golden_full_size_reference.py; weights,
    adapters and inputs are drawn by refiners_amd.synth from the recipe's seeds, so nothing but the fixture's bytes is read here).  The float32 engine is held to
    <= 1e-3 of the same tensor by tests/test_engine_gpu.py::test_full_size_lora_ip_step_matches_oracle."""
    import hashlib

    from safetensors import safe_open

    import refiners_amd
    from refiners_amd import synth
    from refiners_amd.engine.compiled import CompiledSDXL
    from refiners_amd.latent_diffusion.sdxl import SDXLUNet

    name, path = "lora_ip_step7", ROOT / "tests" / "golden" / "full_size_reference.safetensors"
    with safe_open(str(path), framework="pt") as f:
        meta = f.metadata() or {}
        r = json.loads(meta[name])
        if meta.get("synth") != hashlib.sha256((ROOT / "refiners_amd" / "synth.py").read_bytes()).hexdigest():
            return {"parity_bf16_rel_l2": None, "why": "refiners_amd/synth.py changed since the fixture was written (re-run oracle/make_golden_full_size_reference.py)"}
        ref = f.get_tensor(name)
    out = {}
    for dt, tag in ((torch.bfloat16, "bf16"),):
        unet = SDXLUNet(4, device="meta")
        shapes = synth.model_shapes(unet)
        sd = synth.synth_state_dict(shapes, seed=r["weight_seed"])
        unet.load_state_dict({k: v.to(device=dev, dtype=dt) for k, v in sd.items()}, assign=True)
        del sd
        specs = {"loras": [synth.lora_spec(shapes, "l1", 1.0, seed=5), synth.lora_spec(shapes, "l2", 0.8, seed=5)], "ip": synth.ip_spec(shapes, 0.6, batch=2, seed=5), "control": []}
        synth.apply_adapters(unet, refiners_amd.namespace(), device=dev, dtype=dt, **specs)
        inp = synth.sdxl_inputs(r["images"], LATENT, seed=r["input_seed"])
        pipe = CompiledSDXL(unet, num_inference_steps=r["num_steps"], condition_scale=r["condition_scale"], lora_mode="fused")
        pipe.set_inputs(inp["x"].to(dev), clip_text_embedding=inp["text"].to(dev), pooled_text_embedding=inp["pooled"].to(dev), time_ids=inp["time_ids"].to(dev),
                        clip_image_embedding=specs["ip"]["tokens"].to(dev))
        got = pipe.step(r["step"]).float().cpu()
        out[f"parity_{tag}_rel_l2"] = float((got - ref).norm() / ref.norm())
        out[f"parity_{tag}_max_abs_over_max"] = float((got - ref).abs().max() / ref.abs().max())
        del pipe, unet
        torch.cuda.empty_cache()
    out.update(golden="tests/golden/full_size_reference.safetensors:lora_ip_step7 (refiners itself, CPU float32)", recipe=r, lora_mode="fused",
               what="x_next of one CFG + DDIM step, configs[2] at 128x128 latents; f32-I/O contract <= 1e-3 is held by the float32 engine tests against the same tensor")
    return out


def sam_point(dev: torch.device, dtype: torch.dtype) -> dict:
    """configs[4]: SAM ViT-H image encoder with the HQ-SAM early-embedding hook on one 1024x1024 image (random-init weights in HBM)."""
    from refiners_amd.engine.sam import CompiledSAMViT
    from refiners_amd.segment_anything import SAMViTAdapter, SAMViTH

    vit = SAMViTH(device="meta")
    gpu_weights(vit, seed=11, dtype=dtype, device=dev)
    ad = SAMViTAdapter(vit).inject()
    ad.set_context("hq_sam", {"early_vit_embedding": None})
    x = torch.rand(1, 3, 1024, 1024, device=dev).to(dtype)
    fast = CompiledSAMViT(vit)
    with torch.no_grad():
        y = fast(x)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            fast(x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / 5 * 1e3
    return {"workload": "SAM ViT-H image encoder + HQ-SAM hook, 1x3x1024x1024", "ms_per_image": round(ms, 3), "algorithmic_tflop": 5.96, "tflops": round(5.96 / (ms * 1e-3), 1),
            "launches": fast.stats["step_ops"], "fallback_nodes": fast.stats["fallback_nodes"], "output_finite": bool(torch.isfinite(y.float()).all())}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=list(WORKLOADS), default="lora_ip",
                    help="lora_ip = BASELINE configs[2] (the north star's target, default); bare = configs[1]; control = configs[3] (use --images-per-gpu 4 for its 32-prompt / 8-GPU shape)")
    ap.add_argument("--images-per-gpu", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="intra-op threads of the CPU baseline (0 = the best of a sweep over physical / 4, / 2, physical, logical)")
    ap.add_argument("--cpu-budget", type=float, default=75.0, help="seconds of full-size CPU steps after which no further timed step is started (1 warm-up + up to 3 timed, median)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the informational extras (configs[1] line, fused-LoRA line, VAE decode, 4-images-per-GPU point)")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-family replay (profiler passes: the kernel table then holds the timed steps only)")
    ap.add_argument("--dump-program", default=None, help="write the recorded step program (one entry per launch: entry point + shape class) as JSON: tools/profile_round.py")
    ap.add_argument("--packs", choices=["broadcast", "local"], default="broadcast",
                    help="N > 1: packed weights (K-blocked / merged / LayerNorm-folded copies) computed on rank 0 and broadcast (default), or computed by every rank for itself")
    ap.add_argument("--lora-mode", choices=["fused", "merged"], default="fused",
                    help="fused (default, the engine's default and the north star's kernel): run-time LoRA inside the parent launch, adapters stay live; "
                         "merged: W' = W + sum s B A formed at lowering time (reported under extra.lora_mode_merged)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args.gpus)

    from refiners_amd import native, parallel

    rank, world, local = parallel.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run --nproc-per-node {args.gpus})"
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the product path)"
    if os.environ.get("REFINERS_AMD_DIST_BACKEND") == "gloo":
        # rehearsal on a one-GPU box: every rank drives cuda:0 and the collectives go through gloo (host-staged).  It exercises the N > 1
        # control flow (arena broadcast, pack broadcast, barriers, max-over-ranks clock) end to end; its throughput means nothing.
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    native.load()
    dtype = torch.bfloat16
    n_img = args.images_per_gpu
    use_graph = not args.no_graph

    t0 = time.time()
    unet, specs, bare_sd, pipe, bc = build_pipeline(args.workload, n_img, rank, dev, dtype, args.lora_mode, use_graph, packs=args.packs)
    n_params = sum(p.numel() for p in unet.parameters())
    note("pipeline built, timing")
    elapsed = timed_steps(pipe, args.steps, args.warmup, world, dev)
    note(f"timed region done: {elapsed / args.steps * 1e3:.3f} ms per step")
    if world > 1:  # every rank's own clock around the same K steps (value uses the max): a slow GPU / link shows up here
        mine = torch.tensor([timed_steps.last_local_s / args.steps * 1e3], dtype=torch.float64, device=dev)
        bc["per_rank_ms_per_step"] = [round(float(t), 3) for t in parallel.all_gather(mine)]
    setup_s = time.time() - t0 - elapsed
    finite = bool(torch.isfinite(pipe.x.float()).all())
    if world > 1:
        bc["replica_check"] = replica_check(pipe, specs, n_img, dev)

    if rank != 0:
        if world > 1:
            torch.distributed.barrier()
        return

    ms_per_step = elapsed / args.steps * 1e3
    images_per_s = world * n_img / (ms_per_step * 1e-3 * 50)
    if args.dump_program:
        Path(args.dump_program).write_text(json.dumps([program_entry(e) for e in pipe.engine.low.step if e[0] is not None]))
    roofline = None if args.no_roofline else family_roofline(pipe, args.workload, n_img, ms_per_step)
    stats = dict(pipe.engine.stats)

    extra: dict = {"params": n_params, "launches_per_step": stats["step_ops"], "prologue_launches": stats["prologue_ops"], "fallback_nodes": stats["fallback_nodes"],
                   "arena_bytes": stats["pool_bytes"], "weight_prefetch": stats.get("weight_prefetch"), **bc, "setup_s": round(setup_s, 1),
                   "output_finite": finite, "device": native.device_info(), "gemm_tuning": stats.get("gemm_tuning")}

    # ---- CPU baseline: refiners itself where its package is at hand (oracle/_ref), else the mirror's Chain forward; same adapters, one step ----
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline_step(bare_sd, specs, args.workload, args.cpu_threads, args.cpu_budget)
        except Exception as exc:  # noqa: BLE001 -- a baseline failure must not lose the measured line
            cpu = {"value": None, "unit": "images/s", "cores": args.cpu_threads, "kind": "port", "sample": f"failed: {type(exc).__name__}: {exc}"}
        if not args.no_extra:
            try:  # BASELINE configs[0]: the reference's SD1.5 UNet on the same cores
                note("configs[0]: SD1.5 UNet on the CPU")
                extra["configs0_sd15_cpu"] = sd15_cpu_point(int(cpu.get("cores") or physical_cores()))
            except Exception as exc:  # noqa: BLE001
                extra["configs0_sd15_cpu"] = f"failed: {type(exc).__name__}: {exc}"

    if world == 1 and n_img == 1 and not args.no_extra:
        # ---- the same target workload with run-time (exact-order) LoRA instead of merged weights ------------------------
        if args.workload == "lora_ip":
            try:
                other = "fused" if args.lora_mode == "merged" else "merged"
                from refiners_amd.engine.compiled import CompiledSDXL

                p2 = CompiledSDXL(unet, num_inference_steps=50, condition_scale=5.0, use_graph=use_graph, lora_mode=other)
                p2.inputs, p2.x = pipe.inputs, pipe.x.clone()
                p2._tables(dev)
                s2 = timed_steps(p2, 10, 2, 1, dev)
                extra[f"lora_mode_{other}"] = {"ms_per_step": round(s2 / 10 * 1e3, 3), "launches_per_step": p2.engine.stats["step_ops"]}
                del p2
            except Exception as exc:  # noqa: BLE001
                extra["lora_mode_other"] = f"failed: {type(exc).__name__}: {exc}"
        # ---- the headline configuration's throughput points: configs[2] with LIVE LoRAs at 4 and 8 images per GPU (UNet batch 8 / 16) ----
        if args.workload == "lora_ip":
            note("configs[2] throughput points (4 and 8 images per GPU)")
            pts = []
            for n4 in (4, 8):
                try:
                    un4, _, _, pipe4, _ = build_pipeline("lora_ip", n4, rank, dev, dtype, args.lora_mode, use_graph, broadcast=False)
                    s4 = timed_steps(pipe4, 6, 2, 1, dev)
                    ms4 = s4 / 6 * 1e3
                    pts.append({"workload": f"configs[2] (2 LoRA r16 + IP-Adapter), lora_mode {args.lora_mode}", "images_per_gpu": n4, "ms_per_step": round(ms4, 3),
                                "images_per_s": round(n4 / (ms4 * 1e-3 * 50), 4), "step_tflops": round(n4 * STEP_TFLOP["lora_ip"] / (ms4 * 1e-3), 1),
                                "frac_of_peak": round(n4 * STEP_TFLOP["lora_ip"] / (ms4 * 1e-3) / PEAK_BF16_TFLOPS, 4), "launches_per_step": pipe4.engine.stats["step_ops"]})
                    del pipe4, un4
                    torch.cuda.empty_cache()
                except Exception as exc:  # noqa: BLE001
                    pts.append({"images_per_gpu": n4, "failed": f"{type(exc).__name__}: {exc}"})
            extra["throughput_operating_point_configs2"] = pts
        # ---- the benchmarked dtype's error against refiners' own full-size step (reported alongside: BASELINE.md section 4) ----
        try:
            note("bf16 parity point against the reference-written full-size step")
            extra["parity"] = parity_point(dev)
        except Exception as exc:  # noqa: BLE001
            extra["parity"] = {"parity_bf16_rel_l2": None, "why": f"failed: {type(exc).__name__}: {exc}"}
        # ---- next-1 (outside the metric): VAE decode of the finished latents, for an end-to-end images/s figure ---------
        try:
            from refiners_amd.engine.vae import CompiledVAEDecoder
            from refiners_amd.latent_diffusion.vae import SDXLAutoencoder

            vae = SDXLAutoencoder(device="meta")
            gpu_weights(vae, seed=7, dtype=dtype, device=dev)
            note("VAE decode")
            dec = CompiledVAEDecoder(vae)
            z = pipe.x[:1] * 0.13
            dec(z)
            torch.cuda.synchronize()
            tv = time.perf_counter()
            for _ in range(3):
                dec(z)
            torch.cuda.synchronize()
            vae_ms = (time.perf_counter() - tv) / 3 * 1e3
            extra["vae_decode_ms_per_image"] = round(vae_ms, 2)
            extra["vae_fallback_nodes"] = dec.stats.get("fallback_nodes") if hasattr(dec, "stats") else None
            extra["end_to_end_images_per_s_incl_vae"] = round(world * n_img / (ms_per_step * 1e-3 * 50 + n_img * vae_ms * 1e-3), 4)
            del dec, vae
        except Exception as exc:  # noqa: BLE001 -- the VAE is outside the benchmarked path; report, do not fail the bench
            extra["vae_decode_ms_per_image"] = f"failed: {type(exc).__name__}: {exc}"
        # ---- BASELINE configs[1] (no adapters) and the 4-images-per-GPU operating point, through the same engine ----------
        del pipe
        if args.workload != "bare":
            del unet, bare_sd
            try:
                torch.cuda.empty_cache()
                note("configs[1] bare")
                unet_b, _, _, pipe_b, _ = build_pipeline("bare", 1, rank, dev, dtype, args.lora_mode, use_graph, broadcast=False)
                sb = timed_steps(pipe_b, 20, 3, 1, dev)
                msb = sb / 20 * 1e3
                extra["configs1_bare"] = {"ms_per_step": round(msb, 3), "images_per_s": round(1 / (msb * 1e-3 * 50), 4), "launches_per_step": pipe_b.engine.stats["step_ops"],
                                          "step_tflops": round(STEP_TFLOP["bare"] / (msb * 1e-3), 1)}
                del pipe_b
                unet = unet_b
            except Exception as exc:  # noqa: BLE001
                extra["configs1_bare"] = f"failed: {type(exc).__name__}: {exc}"
                unet = None
        if unet is not None and args.workload in ("bare", "lora_ip"):  # `unet` is a bare SDXL UNet here
            try:
                from refiners_amd import synth
                from refiners_amd.engine.compiled import CompiledSDXL

                inp4 = synth.sdxl_inputs(4, LATENT, seed=300)
                note("bare x4 operating point")
                pipe4 = CompiledSDXL(unet, num_inference_steps=50, condition_scale=5.0, use_graph=use_graph, lora_mode=args.lora_mode)
                pipe4.set_inputs(inp4["x"].to(dev), clip_text_embedding=inp4["text"].to(dev), pooled_text_embedding=inp4["pooled"].to(dev), time_ids=inp4["time_ids"].to(dev))
                s4 = timed_steps(pipe4, 10, 2, 1, dev)
                ms4 = s4 / 10 * 1e3
                extra["throughput_operating_point"] = {"workload": "configs[1] (bare)", "images_per_gpu": 4, "ms_per_step": round(ms4, 3), "images_per_s": round(4 / (ms4 * 1e-3 * 50), 4),
                                                       "step_tflops": round(4 * STEP_TFLOP["bare"] / (ms4 * 1e-3), 1),
                                                       "frac_of_peak": round(4 * STEP_TFLOP["bare"] / (ms4 * 1e-3) / PEAK_BF16_TFLOPS, 4)}
                del pipe4
            except Exception as exc:  # noqa: BLE001
                extra["throughput_operating_point"] = f"failed: {type(exc).__name__}: {exc}"

    if world == 1 and n_img == 1 and not args.no_extra:
        # ---- BASELINE configs[3]'s per-GPU shape: ControlLora (canny), 4 prompts per GPU (UNet batch 8) ------------------------------
        try:
            note("configs[3] per-GPU shape")
            unet_c, _, _, pipe_c, _ = build_pipeline("control", 4, rank, dev, dtype, args.lora_mode, use_graph, broadcast=False)
            sc = timed_steps(pipe_c, 6, 2, 1, dev)
            msc = sc / 6 * 1e3
            extra["configs3_per_gpu_shape"] = {"workload": "SDXL-base + ControlLora (canny), 4 images per GPU (32 prompts over 8 GPUs)", "ms_per_step": round(msc, 3),
                                               "images_per_s": round(4 / (msc * 1e-3 * 50), 4), "launches_per_step": pipe_c.engine.stats["step_ops"],
                                               "step_tflops": round(4 * STEP_TFLOP["control"] / (msc * 1e-3), 1), "fallback_nodes": pipe_c.engine.stats["fallback_nodes"]}
            del pipe_c, unet_c
            torch.cuda.empty_cache()
        except Exception as exc:  # noqa: BLE001
            extra["configs3_per_gpu_shape"] = f"failed: {type(exc).__name__}: {exc}"
        # ---- BASELINE configs[4]: SegmentAnything ViT-H image encoder + HQ-SAM hook, 1024x1024, bf16 ---------------------------------
        try:
            note("configs[4] SAM ViT-H")
            extra["configs4_sam_vit_h"] = sam_point(dev, dtype)
        except Exception as exc:  # noqa: BLE001
            extra["configs4_sam_vit_h"] = f"failed: {type(exc).__name__}: {exc}"

    line = {
        "metric": "sdxl_base_1024px_images_per_sec_50_ddim_steps", "value": round(images_per_s, 4), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "SDXL-base UNet CFG step, 1024x1024 (latent 2x4x128x128, 77 text tokens), DDIM-50" + WORKLOADS[args.workload][1],
                   "baseline_config": WORKLOADS[args.workload][0], "images_per_gpu": n_img,
                   "parallelism": f"replica x{world} (independent prompts, weights broadcast once)", "hip_graph": use_graph,
                   "lora_mode": args.lora_mode if args.workload != "bare" else None},
        "step_latency_ms": round(ms_per_step, 3),
        "parity_bf16_rel_l2": (extra.get("parity") or {}).get("parity_bf16_rel_l2"),
        "roofline": roofline, "cpu_baseline": cpu, "extra": extra,
    }
    if os.environ.get("REFINERS_AMD_DIST_BACKEND") == "gloo":
        line["config"]["rehearsal"] = "all ranks on one GPU over gloo: control-flow check only, not a measurement"
    print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()


if __name__ == "__main__":
    main()
