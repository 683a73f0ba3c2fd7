"""Where the SDXL step's time goes, by kernel AND shape: replays the launches of each (entry point, shape) class of the
recorded step program on their own (HIP events), for the bare config-2 workload.  Output: one line per class, sorted by time."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402
from refiners_amd import native, synth  # noqa: E402
from refiners_amd.engine.compiled import CompiledSDXL  # noqa: E402
from refiners_amd.latent_diffusion.sdxl import SDXLUNet  # noqa: E402


def shape_key(e):
    fn, args, what, _ = e
    a = getattr(args[0], "_obj", None)
    if what.startswith("mi355x_gemm"):
        k = sum(a.seg[s].k * (a.seg[s].ksize ** 2 if a.conv else 1) for s in range(a.nseg))
        extra = ("geglu" if a.geglu == 1 else "gelu" if a.geglu else "") + (" res" if a.res else "") + (f" ksplit{a.ksplit}" if a.ksplit > 1 else "") + (f" nseg{a.nseg}" if a.nseg > 1 else "")
        return f"{what} M={a.M} N={a.N} K={k} {extra}".strip()
    if what == "mi355x_attention":
        return f"{what} B={a.B} H={a.H} Lq={a.Lq} Lk={a.kv[0].Lk}"
    if what == "mi355x_layernorm":
        return f"{what} M={a.M} C={a.C}"
    if what == "mi355x_groupnorm":
        return f"{what} B={a.B} HW={a.HW} C={a.C}"
    return what


def main():
    dev = torch.device("cuda", 0)
    native.load()
    dtype = torch.bfloat16
    unet = SDXLUNet(4, device="meta")
    bench.gpu_weights(unet, seed=0, dtype=dtype, device=dev)
    inp = synth.sdxl_inputs(1, bench.LATENT, seed=100)
    pipe = CompiledSDXL(unet, num_inference_steps=50, condition_scale=5.0, use_graph=False, lora_mode="merged")
    pipe.set_inputs(inp["x"].to(dev), clip_text_embedding=inp["text"].to(dev), pooled_text_embedding=inp["pooled"].to(dev), time_ids=inp["time_ids"].to(dev))
    pipe.step(0)
    torch.cuda.synchronize()
    print("weight prefetch:", pipe.engine.stats.get("weight_prefetch"))
    # whole-step replay time (no graph), the number the per-class sums should add up to
    print(f"whole step program, plain replay: {bench.time_ops(pipe.engine.low.step, iters=10) * 1e3:.2f} ms")
    groups = {}
    for e in pipe.engine.low.step:
        if e[0] is not None:
            groups.setdefault(shape_key(e), []).append(e)
    variants = [("default", 0, 0)] + [(f"tile{t}_stages{st}", t, st) for t, st in (tuple(int(v) for v in a.split(",")) for a in sys.argv[1:])]
    for label, tile, stages in variants:
        native.load().mi355x_set_option(b"tile", tile)
        native.load().mi355x_set_option(b"stages", stages)
        rows = []
        for key, ops in groups.items():
            if label != "default" and not key.startswith("mi355x_gemm"):
                continue
            sec = bench.time_ops(ops, iters=10)
            fl = sum(bench.op_flops(e) for e in ops)
            rows.append(dict(cls=key, launches=len(ops), ms=sec * 1e3, us_each=sec / len(ops) * 1e6, tflops=fl / sec / 1e12 if fl else 0.0))
        rows.sort(key=lambda r: -r["ms"])
        total = sum(r["ms"] for r in rows)
        print(f"== {label}: sum over classes: {total:.2f} ms, {sum(r['launches'] for r in rows)} launches")
        for r in rows[: (200 if label == "default" else 14)]:
            print(f"{r['ms']:7.3f} ms {100 * r['ms'] / total:5.1f}%  x{r['launches']:3d}  {r['us_each']:7.1f} us  {r['tflops']:6.0f} TF  {r['cls']}")
    native.load().mi355x_set_option(b"tile", 0)
    native.load().mi355x_set_option(b"stages", 0)


if __name__ == "__main__":
    main()
