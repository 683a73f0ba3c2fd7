"""Whole-step replay time of the bare SDXL config under different weight-prefetch schedules (re-linking the same program)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402
from refiners_amd import native, synth  # noqa: E402
from refiners_amd.engine.compiled import CompiledSDXL  # noqa: E402
from refiners_amd.latent_diffusion.sdxl import SDXLUNet  # noqa: E402


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
    ops = pipe.engine.low.step

    def t():
        return min(bench.time_ops(ops, iters=10) for _ in range(2)) * 1e3

    native.link_weight_prefetch(ops, enable=False)
    print(f"no prefetch: {t():.2f} ms", flush=True)
    native.link_weight_prefetch(ops)
    print(f"prefetch on, replay without a graph: {t():.2f} ms  (python glue ops in the program: {sum(1 for e in ops if e[0] is None)})", flush=True)
    # the number that matters: one HIP-graph launch per step
    import time
    for i in range(3):
        pipe2 = CompiledSDXL(unet, num_inference_steps=50, condition_scale=5.0, use_graph=True, lora_mode="merged")
        pipe2.set_inputs(inp["x"].to(dev), clip_text_embedding=inp["text"].to(dev), pooled_text_embedding=inp["pooled"].to(dev), time_ids=inp["time_ids"].to(dev))
        for k in range(3):
            pipe2.step(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(20):
            pipe2.step(k)
        torch.cuda.synchronize()
        print(f"graph replay: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step", flush=True)


if __name__ == "__main__":
    main()
