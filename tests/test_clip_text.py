"""Prompt encoder: CPU oracle and host mirror vs the real reference's DoubleTextEncoder;
the re-implemented BPE tokenizer vs the reference's (build container only: needs the vocabulary file)."""
import json
from pathlib import Path

import pytest
import torch

from oracle import clip_oracle
from refiners_amd import synth
from refiners_amd.clip import CLIPTokenizer
from refiners_amd.latent_diffusion.prompt import DoubleTextEncoder
from tests import support as S
from tests.golden_cases import CLIP_CASE

TOL = 2e-4
REF_VOCAB = Path("/root/reference/src/refiners/foundationals/clip/bpe_simple_vocab_16e6.txt.gz")


@pytest.fixture(scope="module")
def clip_inputs():
    shapes = {k: tuple(v) for k, v in json.loads((S.GOLD / "double_text_encoder_keys.json").read_text()).items()}
    return shapes, synth.synth_state_dict(shapes, CLIP_CASE["weight_seed"]), S.golden("double_text_encoder")


def test_prompt_encoder_oracle_matches_reference(clip_inputs):
    _, sd, gold = clip_inputs
    emb, pooled = clip_oracle.double_text_encoder(sd, gold["tokens_l"], gold["tokens_g"])
    for got, want in ((emb, gold["text_embedding"]), (pooled, gold["pooled"])):
        l2, mx = S.rel_err(got, want)
        assert l2 < TOL and mx < TOL, (l2, mx)


def test_prompt_encoder_mirror_matches_reference(clip_inputs):
    shapes, sd, gold = clip_inputs
    enc = DoubleTextEncoder(device="meta")
    assert {k: tuple(v.shape) for k, v in enc.state_dict().items()} == shapes and list(enc.state_dict()) == list(shapes)
    enc.load_state_dict(sd, assign=True)
    # feed the golden token ids through the tree below the tokenizers (the vocabulary file does not travel)
    for tok in [m for m in enc.modules() if isinstance(m, CLIPTokenizer)]:
        tok.forward = (lambda t: (lambda _text: gold["tokens_g" if t.pad_token_id == 0 else "tokens_l"].long()))(tok)  # type: ignore[method-assign]
    with torch.no_grad():
        emb, pooled = enc(list(CLIP_CASE["prompts"]))
    for got, want in ((emb, gold["text_embedding"]), (pooled, gold["pooled"])):
        l2, mx = S.rel_err(got, want)
        assert l2 < TOL and mx < TOL, (l2, mx)


@pytest.mark.skipif(not REF_VOCAB.is_file(), reason="needs the CLIP BPE vocabulary that ships with refiners")
def test_tokenizer_matches_reference_token_ids():
    gold = S.golden("double_text_encoder")
    for pad, key in ((49407, "tokens_l"), (0, "tokens_g")):
        tok = CLIPTokenizer(vocabulary_path=REF_VOCAB, pad_token_id=pad)
        assert torch.equal(tok(list(CLIP_CASE["prompts"])).to(torch.int32), gold[key])
    tok = CLIPTokenizer(vocabulary_path=REF_VOCAB)
    # known-answer ids of openai/CLIP's tokenizer for a few words (start, ..., end)
    assert tok.encode("a cute cat").tolist() == [49406, 320, 2242, 2368, 49407]
    assert tok("hello world").shape == (1, 77) and int(tok("x" * 500).shape[1]) == 77


# ------------------------------------------------------------------------------------------------ image prompt side
@pytest.fixture(scope="module")
def image_inputs():
    from tests.golden_cases import CLIP_IMAGE_CASE

    keys = json.loads((S.GOLD / "clip_image_h_keys.json").read_text())
    shapes = {k: tuple(v) for k, v in keys["encoder"].items()}
    pshapes = {k: tuple(v) for k, v in keys["image_proj"].items()}
    sd = synth.synth_state_dict(shapes, CLIP_IMAGE_CASE["weight_seed"])
    psd = synth.synth_state_dict(pshapes, CLIP_IMAGE_CASE["weight_seed"] + 1)
    image = torch.randn((1, 3, 224, 224), generator=synth._gen("clip.image", CLIP_IMAGE_CASE["input_seed"]))
    return shapes, pshapes, sd, psd, image, S.golden("clip_image_h")


def test_image_prompt_oracle_matches_reference(image_inputs):
    _, _, sd, psd, image, gold = image_inputs
    emb = clip_oracle.clip_image_encoder(sd, image)
    l2, mx = S.rel_err(emb, gold["embedding"])
    assert l2 < TOL and mx < TOL, (l2, mx)
    l2, mx = S.rel_err(clip_oracle.image_prompt_tokens(psd, emb), gold["clip_image_embedding"])
    assert l2 < TOL and mx < TOL, (l2, mx)


def test_image_prompt_mirror_matches_reference(image_inputs):
    from refiners_amd.clip_image import CLIPImageEncoderH
    from refiners_amd.latent_diffusion.adapters import ImageProjection

    shapes, pshapes, sd, psd, image, gold = image_inputs
    enc = CLIPImageEncoderH(device="meta")
    proj = ImageProjection(clip_image_embedding_dim=1024, clip_text_embedding_dim=2048, num_tokens=4, device="meta")
    assert list(enc.state_dict()) == list(shapes) and {k: tuple(v.shape) for k, v in enc.state_dict().items()} == shapes
    assert {k: tuple(v.shape) for k, v in proj.state_dict().items()} == pshapes
    enc.load_state_dict(sd, assign=True)
    proj.load_state_dict(psd, assign=True)
    with torch.no_grad():
        emb = enc(image)
        tokens = torch.cat((proj(torch.zeros_like(emb)), proj(emb)))
    for got, want in ((emb, gold["embedding"]), (tokens, gold["clip_image_embedding"])):
        l2, mx = S.rel_err(got, want)
        assert l2 < TOL and mx < TOL, (l2, mx)


def test_fine_grained_image_prompt_mirror_matches_reference(image_inputs):
    """The "plus" IP-Adapter (image_prompt.py:81-234, 516-525, 553-564): grid-feature encoder + PerceiverResampler mirror vs
    the REAL reference's (2, 16, 2048) tokens; state-dict keys equal to the reference's; SDXLIPAdapter(fine_grained=True) builds
    the same resampler and computes the same tokens."""
    from refiners_amd.clip_image import CLIPImageEncoderH
    from refiners_amd.latent_diffusion.adapters import PerceiverResampler, SDXLIPAdapter, convert_to_grid_features
    from refiners_amd.latent_diffusion.sdxl import SDXLUNet
    from tests.golden_cases import CLIP_IMAGE_CASE

    shapes, _, sd, _, image, gold = image_inputs
    rshapes = {k: tuple(v) for k, v in json.loads((S.GOLD / "clip_image_h_keys.json").read_text())["perceiver"].items()}
    enc = CLIPImageEncoderH(device="meta")
    enc.load_state_dict(sd, assign=True)
    ad = SDXLIPAdapter(target=SDXLUNet(4, device="meta"), clip_image_encoder=enc, fine_grained=True)
    res = ad.image_proj
    assert isinstance(res, PerceiverResampler) and list(res.state_dict()) == list(rshapes) and {k: tuple(v.shape) for k, v in res.state_dict().items()} == rshapes
    res.load_state_dict(synth.synth_state_dict(rshapes, CLIP_IMAGE_CASE["weight_seed"] + 2), assign=True)
    grid = ad.grid_image_encoder
    assert len(grid) == 3 and len(grid[-1]) == 31 and len(enc[2]) == 32  # a structural copy: the adapter's own encoder is untouched
    with torch.no_grad():
        feats = grid(image)
        tokens = ad.compute_image_tokens(image)
    l2, mx = S.rel_err(feats[:, ::16, ::16], gold["grid_features_sample"])
    assert l2 < TOL and mx < TOL, (l2, mx)
    l2, mx = S.rel_err(tokens, gold["plus_image_embedding"])
    assert l2 < TOL and mx < TOL, (l2, mx)
    assert next(convert_to_grid_features(enc).parameters()) is next(enc.parameters())  # structural_copy shares the weighted leaves, like the reference's
