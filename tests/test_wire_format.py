"""Weight wire format: a CivitAI-style SDXL LoRA file (kohya key names, down / up pairs,
alphas) goes through the mirror's SDLoraManager to exactly the layers the REAL reference's manager picks
(tests/golden/lora_wire_sdxl.json, written by oracle/make_golden_wire.py), and a Chain round-trips through safetensors
under the reference's state-dict keys."""
import hashlib
import re
import json
from types import SimpleNamespace

import torch

from refiners_amd.fluxion.adapters import Lora, LoraAdapter
from refiners_amd.latent_diffusion.adapters import SDLoraManager
from refiners_amd.latent_diffusion.sdxl import SDXLUNet
from tests import support as S
from tests.golden_cases import kohya_sdxl_lora_keys


def test_lora_file_attaches_where_the_reference_manager_attaches_it():
    gold = json.loads((S.GOLD / "lora_wire_sdxl.json").read_text())
    keys = kohya_sdxl_lora_keys(rank=8)
    assert [[k, list(s)] for k, s in keys] == gold["keys"]
    tensors = {k: torch.empty(shape, device="meta") for k, shape in keys}
    unet = SDXLUNet(4, device="meta")
    manager = SDLoraManager(SimpleNamespace(unet=unet, device=torch.device("meta"), dtype=torch.float32))
    manager.add_loras("style", tensors=tensors, scale=0.75)
    attached = [[a.get_path(), [list(lr.down.weight.shape) for lr in a.loras.values()]] for a in unet.layers(LoraAdapter)]
    assert attached == gold["attached"]
    assert hashlib.sha256(re.sub(r"Lambda\(.*\)", "Lambda", repr(unet)).encode()).hexdigest() == gold["repr_sha256"]
    assert manager.scales == gold["scales"] and manager.names == gold["names"]
    # the key -> layer map of the two-step route (sorted keys, res / downsample / upsample pre-pass)
    unet2 = SDXLUNet(4, device="meta")
    manager2 = SDLoraManager(SimpleNamespace(unet=unet2, device=torch.device("meta"), dtype=torch.float32))
    loras = Lora.from_dict("style", state_dict=tensors)
    loras = {k: loras[k] for k in sorted(loras, key=SDLoraManager.sort_keys)}
    key_map: list[tuple[str, str]] = []
    manager2.add_loras_to_unet(loras, debug_map=key_map)
    assert [list(kv) for kv in key_map] == gold["key_map"]
    # and back out: ejecting every adapter restores the bare tree
    bare = repr(SDXLUNet(4, device="meta"))
    manager.remove_all()
    assert repr(unet) == bare


def test_chain_round_trips_through_safetensors_under_reference_keys(tmp_path):
    from safetensors.torch import save_file

    from refiners_amd import synth
    from refiners_amd.latent_diffusion.vae import SDXLAutoencoder

    shapes = {k: tuple(v) for k, v in json.loads((S.GOLD / "vae_keys.json").read_text()).items()}
    sd = synth.synth_state_dict(shapes, 3)
    path = tmp_path / "vae.safetensors"
    save_file({k: v.contiguous() for k, v in sd.items()}, str(path))  # a file keyed like refiners' converted checkpoints
    vae = SDXLAutoencoder(device="cpu").load_from_safetensors(path)
    got = vae.state_dict()
    assert list(got) == list(sd) and all(torch.equal(got[k], sd[k]) for k in sd)


def test_ip_adapter_checkpoint_lands_where_the_reference_puts_it():
    """`image_proj.*` + `ip_adapter.NNN.*` (image_prompt.py:395-410): the NNN-th pair of tensors -- in FILE order -- becomes the
    key / value projection of the NNN-th text cross-attention.  The tensors' second dimension is a tag (golden_cases.ip_adapter_file),
    so the shapes read back from the loaded adapter say which file tensor landed where; compared with the real reference's result."""
    from refiners_amd.latent_diffusion.adapters import SDXLIPAdapter
    from tests.golden_cases import ip_adapter_file, tagged

    gold = json.loads((S.GOLD / "adapter_wire_sdxl.json").read_text())["ip_adapter"]
    keys = ip_adapter_file()
    assert [[k, list(s)] for k, s in keys] == gold["keys"]
    unet = SDXLUNet(4, device="meta")
    ad = SDXLIPAdapter(target=unet, clip_image_encoder=SimpleNamespace(output_dim=1024, embedding_dim=1280), weights=tagged(keys, device="meta"))
    landed = [[i, list(sub.image_key_projection.weight.shape), list(sub.image_value_projection.weight.shape)] for i, sub in enumerate(ad.sub_adapters)]
    assert landed == gold["landed"]
    assert [[k, list(v.shape)] for k, v in ad.image_proj.state_dict().items()] == gold["image_proj_keys"]
    # a SHUFFLED file keeps working as long as each index's key tensor precedes its value tensor (the format is positional)
    swapped = dict(reversed(list(tagged(keys, device="meta").items())))
    ad2 = SDXLIPAdapter(target=SDXLUNet(4, device="meta"), clip_image_encoder=SimpleNamespace(output_dim=1024, embedding_dim=1280), weights=swapped)
    assert [list(s.image_key_projection.weight.shape) for s in ad2.sub_adapters] == [row[2] for row in gold["landed"]]  # reversed order -> value tensor first


def test_control_lora_checkpoint_lands_where_the_reference_puts_it():
    """`ControlLora.<path>.{down,up}` / `ZeroConvolution_NN.*` / `ConditionEncoder.*` (xl/control_lora.py:345-411)."""
    from refiners_amd.latent_diffusion.adapters import ConditionEncoder, ControlLoraAdapter, ZeroConvolution
    from tests.golden_cases import control_lora_file, tagged

    gold = json.loads((S.GOLD / "adapter_wire_sdxl.json").read_text())["control_lora"]
    keys = control_lora_file()
    assert [[k, list(s)] for k, s in keys] == gold["keys"]
    unet = SDXLUNet(4, device="meta")
    cad = ControlLoraAdapter(name="canny", target=unet, scale=0.7, weights=tagged(keys, device="meta"))
    cl = cad.control_lora
    attached = [[a.get_path(), [list(lr.down.weight.shape) for lr in a.loras.values()], list(a.names)] for a in cl.layers(LoraAdapter)]
    assert attached == gold["attached"]
    assert [[list(z.state_dict()), [list(v.shape) for v in z.state_dict().values()]] for z in cl.layers(ZeroConvolution)] == gold["zero"]
    assert [[k, list(v.shape)] for k, v in cl.ensure_find(ConditionEncoder).state_dict().items()] == gold["encoder"]
    assert hashlib.sha256(re.sub(r"Lambda\(.*\)", "Lambda", repr(cl)).encode()).hexdigest() == gold["repr_sha256"]
    # values: real tensors through the same path end up in the parameters they are keyed for
    small = {k: torch.randn(shape) for k, shape in keys if k.startswith(("ZeroConvolution_03", "ConditionEncoder.Chain_1"))}
    zc = list(cl.layers(ZeroConvolution))[2]
    zc.to_empty(device="cpu")
    zc.load_state_dict({k.removeprefix("ZeroConvolution_03."): v for k, v in small.items() if k.startswith("ZeroConvolution_03")})
    assert torch.equal(zc.state_dict()["Conv2d.weight"], small["ZeroConvolution_03.Conv2d.weight"])
