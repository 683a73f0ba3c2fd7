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
    for legacy in (1, 0, 1, 0):
        native.load().mi355x_set_option(b"legacy", legacy)
        print(f"{'per-iteration address arithmetic (old loader)' if legacy else 'hoisted row pointers (new loader)       '}: {t():.2f} ms", flush=True)
    native.load().mi355x_set_option(b"legacy", 0)


if __name__ == "__main__":
    main()
