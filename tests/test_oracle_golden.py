"""The CPU oracle must reproduce what the REAL reference produced (tests/golden/*, written by oracle/make_golden.py
from finegrain-ai/refiners itself) from the seeds alone.  This is what pins the oracle; float32 CPU, tolerance 2e-4
relative (torch CPU kernels may differ in summation order between machines; observed 1e-6)."""
import pytest
import torch

from oracle import unet_oracle as O
from tests import support as S

TOL = 2e-4


@pytest.mark.parametrize("case", [c for c, cfg in S.CASES.items() if cfg["family"] == "sdxl"])
def test_sdxl_oracle_matches_reference(case):
    cfg = S.CASES[case]
    sd = S.weights("sdxl", cfg["weight_seed"])
    gold = S.golden(case)
    inp = S.synth.sdxl_inputs(cfg["images"], cfg["latent_hw"], cfg["input_seed"])
    specs = S.build_specs(cfg, S.key_shapes("sdxl"))
    ts, _ = O.ddim_tables(cfg["num_steps"])
    assert float(ts[cfg["step"]]) == float(gold["timestep"])
    y = O.sdxl_unet(sd, torch.cat((inp["x"], inp["x"])), ts[cfg["step"]].unsqueeze(0), inp["text"], inp["pooled"], inp["time_ids"],
                    **S.oracle_adapters(specs))
    l2, mx = S.rel_err(y, gold["unet_out"])
    assert l2 < TOL and mx < TOL, (case, l2, mx)
    x_next = O.sdxl_cfg_step(sd, inp["x"], cfg["step"], cfg["num_steps"], inp["text"], inp["pooled"], inp["time_ids"],
                             condition_scale=cfg["condition_scale"], **S.oracle_adapters(specs))
    l2, mx = S.rel_err(x_next, gold["x_next"])
    assert l2 < TOL and mx < TOL, (case, "x_next", l2, mx)


def test_sd1_oracle_matches_reference():
    cfg = S.CASES["sd1_bare"]
    sd = S.weights("sd1", cfg["weight_seed"])
    gold = S.golden("sd1_bare")
    x = torch.randn((1, 4, *cfg["latent_hw"]), generator=S.synth._gen("in.x", cfg["input_seed"]))
    text = torch.randn((1, 77, 768), generator=S.synth._gen("in.text", cfg["input_seed"]))
    y = O.sd1_unet(sd, x, torch.tensor([cfg["timestep"]]), text)
    l2, mx = S.rel_err(y, gold["unet_out"])
    assert l2 < TOL and mx < TOL, (l2, mx)
    assert torch.equal(gold["unet_out"], gold["unet_out_again"])  # the reference itself is bit-reproducible


def test_oracle_matches_reference_at_full_size():
    """The oracle pinned to the reference AT THE BENCHMARKED GEOMETRY (128 x 128 latents): refiners' own steps of the FULL_SIZE recipes
    (tests/golden/full_size_reference.safetensors, written in the build container by oracle/make_golden_full_size_reference.py -- configs[1] bare, configs[2] with
    2 LoRAs x 722 Linears + IP-Adapter, configs[3]'s ControlLora image) against the oracle's committed steps of the same recipes.  Two committed tensors per
    recipe, no computation: until round 6 every reference-written golden was 32 x 32."""
    for name in S.FULL_SIZE:
        ref = S.full_size_reference(name)
        assert ref is not None, f"{name}: re-run `python oracle/make_golden_full_size_reference.py` (recipe or synth.py changed)"
        l2, mx = S.rel_err(S.full_size_oracle(name), ref)
        print(f"{name}: oracle vs reference at full size: l2 {l2:.2e} max {mx:.2e}")
        assert l2 < TOL and mx < TOL, (name, l2, mx)


def test_full_size_fixture_is_current():
    """tests/golden/full_size_oracle.safetensors holds the oracle's steps at the benchmarked geometry (the GPU tests compare the engine with them).  An entry is only
    served when its recipe AND the digest of the sources it was computed from (oracle/*.py, refiners_amd/synth.py) match the tree: a stale file would otherwise cost every
    GPU run minutes of host arithmetic -- or, before the digest existed, silently keep a reference the edited oracle no longer produces (round-4 advisor)."""
    import json

    from safetensors import safe_open

    with safe_open(str(S.GOLD / "full_size_oracle.safetensors"), framework="pt") as f:
        meta = f.metadata() or {}
        assert meta.get("sources") == S.oracle_sources_digest(), "re-run `python oracle/make_golden_full_size.py` (oracle or synth sources changed)"
        for name, recipe in S.FULL_SIZE.items():
            assert name in f.keys() and json.loads(meta.get(name, "null")) == recipe, name
            assert tuple(f.get_slice(name).get_shape()) == (1, 4, 128, 128)
