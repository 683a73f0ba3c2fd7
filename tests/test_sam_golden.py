"""BASELINE.json config 5 (SegmentAnything ViT-H image encoder + HQ-SAM encoder hook): the CPU oracle and the host mirror
against outputs of the real reference (tests/golden/sam_vit_h.safetensors, written by oracle/make_golden_sam.py)."""
import json

import pytest
import torch

from oracle import sam_oracle
from refiners_amd import synth
from refiners_amd.segment_anything import SAMViTAdapter, SAMViTH
from tests import support as S
from tests.golden_cases import SAM_CASE, sam_sample

TOL = 2e-4


@pytest.fixture(scope="module")
def sam_inputs():
    shapes = {k: tuple(v) for k, v in json.loads((S.GOLD / "sam_vit_h_keys.json").read_text()).items()}
    sd = synth.synth_state_dict(shapes, SAM_CASE["weight_seed"])
    image = torch.rand((1, 3, 1024, 1024), generator=synth._gen("sam.image", SAM_CASE["input_seed"]))
    return shapes, sd, image


def _check(neck, early):
    gold = S.golden("sam_vit_h")
    got = sam_sample(neck, early)
    for k in ("neck", "early", "stats"):
        l2, mx = S.rel_err(got[k], gold[k])
        assert l2 < TOL and mx < TOL, (k, l2, mx)


def test_sam_oracle_matches_reference(sam_inputs):
    _, sd, image = sam_inputs
    neck, early = sam_oracle.sam_vit(sd, image)
    _check(neck, early)


def test_sam_mirror_matches_reference(sam_inputs):
    shapes, sd, image = sam_inputs
    vit = SAMViTH(device="meta")
    assert {k: tuple(v.shape) for k, v in vit.state_dict().items()} == shapes and list(vit.state_dict()) == list(shapes)
    vit.load_state_dict(sd, assign=True)
    before = repr(vit)
    adapter = SAMViTAdapter(vit).inject()
    adapter.set_context("hq_sam", {"early_vit_embedding": None})
    with torch.no_grad():
        neck = adapter(image)
    early = vit.layer(("Transformer", 7), torch.nn.Module).use_context("hq_sam")["early_vit_embedding"]
    _check(neck, early)
    adapter.eject()
    assert repr(vit) == before
