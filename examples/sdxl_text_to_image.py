"""Text -> image with every stage on the MI355X engine (the pipeline of tests/test_end_to_end_gpu.py as a script):

    prompt --BPE (host)--> token ids --CompiledDoubleTextEncoder--> embeddings --CompiledSDXL (CFG + solver, one HIP graph per
    step)--> latents --CompiledVAEDecoder--> image tensor in [-1, 1]

Weights: pass refiners-format safetensors (the files refiners' conversion scripts write; state-dict keys are identical) with
--unet / --text-encoder / --vae; without them the models get seeded random weights, which exercises the whole path but draws noise.
The tokenizer needs CLIP's BPE vocabulary (--vocab or REFINERS_AMD_CLIP_VOCAB; it ships with refiners and with openai/CLIP).

    python examples/sdxl_text_to_image.py --prompt "a photo of a cat" --steps 30 --solver ddim --out cat.pt
"""
from __future__ import annotations

import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402


def random_weights(module, seed: int, dtype, device) -> None:
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for k, p in module.state_dict().items():
        n = torch.randn(tuple(p.shape), generator=g, device=device)
        leaf, kind = (k.split(".") + ["", ""])[-2:] if "." in k else ("", k)
        fan_in = 1
        for d in p.shape[1:]:
            fan_in *= d
        t = (1 + 0.1 * n) if ("Norm" in leaf and kind == "weight") else (0.1 * n if p.dim() < 2 else n / max(fan_in, 1) ** 0.5)
        sd[k] = t.to(dtype)
    module.load_state_dict(sd, assign=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompt", default="a photograph of an astronaut riding a horse")
    ap.add_argument("--negative-prompt", default="")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--solver", choices=["ddim", "euler", "dpm"], default="ddim")
    ap.add_argument("--guidance", type=float, default=5.0)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--unet"), ap.add_argument("--text-encoder"), ap.add_argument("--vae"), ap.add_argument("--vocab")
    ap.add_argument("--out", default="image.pt")
    args = ap.parse_args()

    from refiners_amd import native
    from refiners_amd.clip import CLIPTokenizer
    from refiners_amd.engine.compiled import CompiledSDXL
    from refiners_amd.engine.text import CompiledDoubleTextEncoder
    from refiners_amd.engine.vae import CompiledVAEDecoder
    from refiners_amd.latent_diffusion.prompt import DoubleTextEncoder
    from refiners_amd.latent_diffusion.sdxl import SDXLUNet
    from refiners_amd.latent_diffusion.solvers import DPMSolver, Euler
    from refiners_amd.latent_diffusion.vae import SDXLAutoencoder

    assert torch.cuda.is_available(), "this script needs an MI355X (the product path has no CPU fallback)"
    native.load()
    dev, dtype = torch.device("cuda", 0), torch.bfloat16
    models = {}
    for name, cls, path, seed in (("text", DoubleTextEncoder, args.text_encoder, 1), ("unet", lambda **kw: SDXLUNet(4, **kw), args.unet, 2),
                                  ("vae", SDXLAutoencoder, args.vae, 3)):
        if path:
            models[name] = cls(device=dev, dtype=dtype).load_from_safetensors(path)
        else:
            models[name] = cls(device="meta")
            random_weights(models[name], seed, dtype, dev)
    if args.vocab:
        for tk in [m for m in models["text"].modules() if isinstance(m, CLIPTokenizer)]:
            tk.vocabulary_path = Path(args.vocab)

    prompts = [args.negative_prompt, args.prompt]  # [negative ; conditional], the order the CFG step expects
    t0 = time.perf_counter()
    emb, pooled = CompiledDoubleTextEncoder(models["text"])(prompts)
    solver = {"ddim": None, "euler": Euler(args.steps, device=dev), "dpm": DPMSolver(args.steps, device=dev)}[args.solver]
    sd = CompiledSDXL(models["unet"], num_inference_steps=args.steps, condition_scale=args.guidance, lora_mode="merged", solver=solver)
    h = w = args.size // 8
    x = torch.randn((1, 4, h, w), generator=torch.Generator().manual_seed(args.seed)).to(dev, dtype)
    if args.solver == "euler":
        x = x * float(solver.init_noise_sigma)
    time_ids = torch.tensor([[args.size, args.size, 0, 0, args.size, args.size]] * 2, device=dev, dtype=torch.float32)
    sd.set_inputs(x, clip_text_embedding=emb, pooled_text_embedding=pooled, time_ids=time_ids)
    latents = sd.sample()
    image = CompiledVAEDecoder(models["vae"])(latents)
    torch.cuda.synchronize()
    print(f"{args.steps} steps + text encoders + VAE decode: {time.perf_counter() - t0:.2f} s (first call: includes lowering and graph capture)")
    torch.save(image.float().cpu(), args.out)
    print("saved", args.out, tuple(image.shape), "range", float(image.min()), float(image.max()))


if __name__ == "__main__":
    main()
