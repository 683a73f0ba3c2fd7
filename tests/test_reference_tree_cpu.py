"""Only in the build container (skipped where /root/reference is absent): the engine's lowering accepts trees built from
refiners' OWN classes (it matches on class names), and produces the same program as for the mirror."""
import sys
from collections import Counter
from pathlib import Path
from types import SimpleNamespace

import pytest
import torch

REF = Path("/root/reference/src")
pytestmark = pytest.mark.skipif(not REF.exists(), reason="the reference checkout is only present in the build container")


def test_lowering_accepts_the_real_refiners_tree():
    root = Path(__file__).resolve().parent.parent
    sys.path[:0] = [str(root / "oracle" / "shim"), str(REF)]
    import refiners.fluxion.layers as rfl
    from refiners.fluxion.adapters.lora import Conv2dLora, LinearLora, LoraAdapter
    from refiners.foundationals.latent_diffusion.stable_diffusion_xl.control_lora import ConditionEncoder, ControlLoraAdapter, ZeroConvolution
    from refiners.foundationals.latent_diffusion.stable_diffusion_xl.image_prompt import SDXLIPAdapter
    from refiners.foundationals.latent_diffusion.stable_diffusion_xl.unet import SDXLUNet as RefUNet

    import refiners_amd
    from refiners_amd import synth
    from refiners_amd.latent_diffusion.sdxl import SDXLUNet
    from tests import support as S
    from tests.test_lowering_cpu import _dry

    ref_api = SimpleNamespace(fl=rfl, LinearLora=LinearLora, Conv2dLora=Conv2dLora, LoraAdapter=LoraAdapter, SDXLIPAdapter=SDXLIPAdapter,
                              ControlLoraAdapter=ControlLoraAdapter, ConditionEncoder=ConditionEncoder, ZeroConvolution=ZeroConvolution)
    for case in ("sdxl_lora_ip", "sdxl_control"):
        cfg = S.CASES[case]
        specs = S.build_specs(cfg, S.key_shapes("sdxl"))
        tokens = {("cross_attention_block", "clip_text_embedding"): (77, 2048)}
        if specs["ip"] is not None:
            tokens[("ip_adapter", "clip_image_embedding")] = (4, 2048)
        conds = [f"control_lora_{c['name']}" for c in specs["control"]]
        programs = []
        for cls, api in ((RefUNet, ref_api), (SDXLUNet, refiners_amd.namespace())):
            unet = cls(4, device="meta", dtype=torch.bfloat16)
            synth.apply_adapters(unet, api, device="meta", dtype=torch.bfloat16, **specs)
            low = _dry(unet, 2, *cfg["latent_hw"], torch.bfloat16, tokens, conditions=conds)
            programs.append((Counter(e[2] for e in low.step), Counter(e[2] for e in low.prologue), low.stats["lora_sites"], low.stats["ip_sites"], low.stats["fallback_nodes"]))
        assert programs[0] == programs[1], case


def _programs_equal(a, b):
    from collections import Counter as C

    return C(e[2] for e in a.step) == C(e[2] for e in b.step) and a.stats["fallback_nodes"] == b.stats["fallback_nodes"]


def test_prompt_side_lowerings_accept_the_real_refiners_trees():
    """DoubleTextEncoder, CLIPImageEncoderH + ImageProjection, SAM ViT-H, the VAE and an SDXL UNet carrying a T2I-Adapter, built
    from refiners' own classes, lower to the same launch programs as the mirror's (dry run on the meta device)."""
    root = Path(__file__).resolve().parent.parent
    sys.path[:0] = [str(root / "oracle" / "shim"), str(REF)]
    from refiners.foundationals.clip.image_encoder import CLIPImageEncoderH as RefImg
    from refiners.foundationals.latent_diffusion.image_prompt import ImageProjection as RefProj
    from refiners.foundationals.latent_diffusion.stable_diffusion_xl.model import SDXLAutoencoder as RefVAE
    from refiners.foundationals.latent_diffusion.stable_diffusion_xl.t2i_adapter import SDXLT2IAdapter as RefT2I
    from refiners.foundationals.latent_diffusion.stable_diffusion_xl.text_encoder import DoubleTextEncoder as RefDTE
    from refiners.foundationals.latent_diffusion.stable_diffusion_xl.unet import SDXLUNet as RefUNet
    from refiners.foundationals.segment_anything.image_encoder import SAMViTH as RefSAM

    from refiners_amd.clip_image import CLIPImageEncoderH
    from refiners_amd.engine.image_prompt import ImagePromptLowering
    from refiners_amd.engine.unet_lowering import UNetIO, UNetLowering
    from refiners_amd.engine.sam import SAMLowering
    from refiners_amd.engine.text import TextLowering
    from refiners_amd.engine.vae import VAEDecoderLowering
    from refiners_amd.latent_diffusion.adapters import ImageProjection
    from refiners_amd.latent_diffusion.prompt import DoubleTextEncoder
    from refiners_amd.latent_diffusion.sdxl import SDXLUNet
    from refiners_amd.latent_diffusion.t2i import SDXLT2IAdapter
    from refiners_amd.latent_diffusion.vae import SDXLAutoencoder
    from refiners_amd.segment_anything import SAMViTH

    dev, dt = torch.device("meta"), torch.bfloat16
    e = lambda *shape, dtype=dt: torch.empty(*shape, device=dev, dtype=dtype)  # noqa: E731
    ti = lambda n: torch.empty(n, device=dev, dtype=torch.int32)  # noqa: E731

    def text(cls):
        low = TextLowering(dev, dt, None, "merged")
        low.lower_double(cls(device="meta", dtype=dt), ti(154), ti(154), ti(2), 2, 77, e(154, 2048), e(2, 1280))
        return low

    def image(enc_cls, proj_cls):
        low = ImagePromptLowering(dev, dt, None, "merged")
        both = e(2, 1024)
        low.lower_image_encoder(enc_cls(device="meta", dtype=dt), e(1, 3, 224, 224), ti(1), both[1:])
        low.lower_image_projection(proj_cls(clip_image_embedding_dim=1024, clip_text_embedding_dim=2048, num_tokens=4, device="meta", dtype=dt), both, e(8, 2048))
        return low

    def sam(cls):
        low = SAMLowering(dev, dt, None, "merged")
        low.lower(cls(device="meta", dtype=dt), e(1, 3, 1024, 1024), e(1, 256, 64, 64), None)
        return low

    def vae(cls):
        low = VAEDecoderLowering(dev, dt, None, "merged")
        model = cls(device="meta", dtype=dt)
        low.lower_decoder(list(model._modules.values())[1], e(1, 4, 32, 32), e(1, 3, 256, 256), float(model.encoder_scale))
        return low

    def t2i(unet_cls, adapter_cls):
        unet = unet_cls(4, device="meta", dtype=dt)
        adapter_cls(unet, name="depth", scale=0.5).inject()
        io = UNetIO(x=e(2, 4, 32, 32), timestep=e(2, dtype=torch.float32), out=e(2, 4, 32, 32))
        io.pooled, io.time_ids = e(2, 1280), e(2, 6, dtype=torch.float32)
        io.tokens[("cross_attention_block", "clip_text_embedding")] = (e(256, 2048), 77)
        io.t2i["depth"] = [e(1, 320, 16, 16), e(1, 640, 16, 16), e(1, 1280, 8, 8), e(1, 1280, 8, 8)]
        low = UNetLowering(dev, dt, None, "merged")
        low.lower(unet, io)
        assert low.stats.get("t2i_sites") == 4
        return low

    assert _programs_equal(text(RefDTE), text(DoubleTextEncoder))
    assert _programs_equal(image(RefImg, RefProj), image(CLIPImageEncoderH, ImageProjection))
    assert _programs_equal(sam(RefSAM), sam(SAMViTH))
    assert _programs_equal(vae(RefVAE), vae(SDXLAutoencoder))
    assert _programs_equal(t2i(RefUNet, RefT2I), t2i(SDXLUNet, SDXLT2IAdapter))
