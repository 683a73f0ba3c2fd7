"""pytest -m gpu: the node-level drop-ins (FusedResidualBlock / FusedCrossAttentionBlock2d adapters) inside an otherwise
unfused tree: same outputs as the plain Chain forward, inject -> eject restores the tree."""
import pytest
import torch

import refiners_amd
from refiners_amd.engine.fused import FusedCrossAttentionBlock2d, FusedResidualBlock, fuse, unfuse
from refiners_amd.latent_diffusion.sampling import DDIM
from refiners_amd.latent_diffusion.sdxl import SDXLUNet
from tests import support as S

pytestmark = pytest.mark.gpu


def test_fused_nodes_inside_the_chain_tree(gpu_device):
    case = "sdxl_lora_ip"
    cfg = S.CASES[case]
    unet = SDXLUNet(4, device="meta")
    S.load_mirror_weights(unet, S.weights("sdxl", cfg["weight_seed"]), device="cuda", dtype=torch.float32)
    specs = S.build_specs(cfg, S.key_shapes("sdxl"))
    S.synth.apply_adapters(unet, refiners_amd.namespace(), device="cuda", dtype=torch.float32, **specs)
    inp = {k: v.cuda() for k, v in S.synth.sdxl_inputs(cfg["images"], cfg["latent_hw"], cfg["input_seed"]).items()}

    def run():
        unet.set_timestep(DDIM(cfg["num_steps"]).timesteps[cfg["step"]].unsqueeze(0).cuda())
        unet.set_clip_text_embedding(inp["text"])
        unet.set_pooled_text_embedding(inp["pooled"])
        unet.set_time_ids(inp["time_ids"])
        with torch.no_grad():
            return unet(torch.cat((inp["x"], inp["x"])))

    before = repr(unet)
    made = fuse(unet)
    assert sum(isinstance(m, FusedResidualBlock) for m in made) == 17
    assert sum(isinstance(m, FusedCrossAttentionBlock2d) for m in made) == 11
    y = run()
    l2, mx = S.rel_err(y, S.golden(case)["unet_out"])
    assert l2 < 1e-3 and mx < 1e-3, (l2, mx)
    y2 = run()  # (not bit-compared: the unfused remainder of the tree runs MIOpen convs, whose first call auto-tunes)
    l2, mx = S.rel_err(y2, y)
    assert l2 < 1e-5 and mx < 1e-5, (l2, mx)
    assert unfuse(unet) == 28
    assert repr(unet) == before
