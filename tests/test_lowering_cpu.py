"""CPU checks of the host side of the engine: the C-ABI library loads and exports what the header declares, the
lowering matches every golden-case tree (dry run on the meta device: argument structs are built, nothing is launched),
and the drop-in reads its inputs where refiners' context store really keeps them."""
import re
from collections import Counter
from pathlib import Path

import pytest
import torch

import refiners_amd
import refiners_amd.fluxion.layers as fl
from refiners_amd import native, synth
from refiners_amd.engine.compiled import CompiledUNet
from refiners_amd.engine.packing import launches
from refiners_amd.engine.unet_lowering import UNetIO, UNetLowering
from refiners_amd.fluxion.tree import tree_epoch
from refiners_amd.latent_diffusion.sd1 import SD1UNet
from refiners_amd.latent_diffusion.sdxl import SDXLUNet
from tests import support as S

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module", autouse=True)
def _lib():
    from refiners_amd.build_native import build_native

    build_native()


def test_library_exports_every_symbol_of_the_header():
    header = (ROOT / "include" / "mi355x_refiners.h").read_text()
    declared = set(re.findall(r"\b(mi355x_[a-z0-9_]+)\s*\(", header))
    assert declared == set(native.EXPORTS), declared ^ set(native.EXPORTS)
    lib = native.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mi355x_abi_version() == 7


def _dry(unet, B, H, W, dtype, tokens, pooled=True, conditions=(), condition_rows=None):
    dev = torch.device("meta")
    io = UNetIO(x=torch.empty(B, 4, H, W, device=dev, dtype=dtype), timestep=torch.empty(B, device=dev), out=torch.empty(B, 4, H, W, device=dev, dtype=dtype))
    if pooled:
        io.pooled = torch.empty(B, 1280, device=dev, dtype=dtype)
        io.time_ids = torch.empty(B, 6, device=dev)
    for ck, (L, width) in tokens.items():
        io.tokens[ck] = (torch.zeros(B * ((L + 63) // 64 * 64), width, device=dev, dtype=dtype), L)
    for name in conditions:
        io.conditions[name] = torch.empty(condition_rows or B, 3, 8 * H, 8 * W, device=dev, dtype=dtype)
    low = UNetLowering(dev, dtype)
    low.lower(unet, io)
    return low


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [c for c, cfg in S.CASES.items() if cfg["family"] == "sdxl"])
def test_every_golden_tree_lowers_without_fallback(case, dtype):
    cfg = S.CASES[case]
    unet = SDXLUNet(4, device="meta", dtype=dtype)
    specs = S.build_specs(cfg, S.key_shapes("sdxl"))
    synth.apply_adapters(unet, refiners_amd.namespace(), device="meta", dtype=dtype, **specs)
    tokens = {("cross_attention_block", "clip_text_embedding"): (77, 2048)}
    if specs["ip"] is not None:
        tokens[("ip_adapter", "clip_image_embedding")] = (4, 2048)
    low = _dry(unet, 2, *cfg["latent_hw"], dtype, tokens, conditions=[f"control_lora_{c['name']}" for c in specs["control"]])
    kinds = Counter(e[2] for e in low.step)
    nc = len(specs["control"])  # every ControlLora runs its own copy of the encoder half: 68 attentions, 102 LayerNorms
    assert kinds["mi355x_attention"] == 140 + 68 * nc
    assert kinds["mi355x_layernorm"] == 210 + 102 * nc
    if nc == 2:  # stacked adapters keep their own contexts and zero convolutions (xl/control_lora.py:251-411)
        assert len({k for k in low.io.conditions}) == 2 and sum(1 for e in low.prologue if e[2] == "mi355x_gemm(conv)") == 16
    # the nine ResidualConcatenator outputs are never written (two-source GroupNorm + two shortcut segments): no concat launch at all
    # (a ResidualBlock whose conv2 carries a LoRA on the two-launch path -- in-launch LoRA is a device-only decision, off on meta -- has no
    # segment left for the second shortcut part: there the concatenation is written after all)
    mat = low.stats.get("concat_materialised", 0)
    assert kinds["mi355x_groupnorm"] >= 46 and kinds["mi355x_concat2"] == mat and (mat == 0 or case == "sdxl_conv_lora")
    gn2 = [e[1][0]._obj for e in low.step if e[2] == "mi355x_groupnorm" and e[1][0]._obj.C1 > 0]  # (pointers are 0 on the meta device: C1 marks the two-source form)
    assert len(gn2) == 9 - mat and (mat or sorted(int(g.C) for g in gn2) == sorted([2560, 2560, 1920, 1920, 1280, 960, 960, 640, 640]))
    # the text / image K and V^T projections are hoisted out of the per-step program
    assert sum(1 for e in low.prologue if e[2] == "mi355x_gemm") >= 140
    assert [f for f in low.stats["fallback_nodes"] if "ConditionEncoder" not in f] == []
    if case == "sdxl_lora_ip":
        assert low.stats["lora_sites"] == 722 and low.stats["ip_sites"] == 70
    if case == "sdxl_bare":
        assert len(low.step) < 1000  # the reference issues ~2 700 module calls for the same work


def test_one_control_picture_for_the_whole_batch_costs_no_step_launch():
    """A batch-1 control picture (what the reference's Sum broadcasts, and the only form its Self-Attention Guidance pass can take) is encoded
    once and repeated per prompt in the PROLOGUE: the step program is the one of a 2n-row picture."""
    cfg = S.CASES["sdxl_control"]
    progs = []
    for rows in (None, 1):
        unet = SDXLUNet(4, device="meta", dtype=torch.bfloat16)
        specs = S.build_specs(cfg, S.key_shapes("sdxl"))
        synth.apply_adapters(unet, refiners_amd.namespace(), device="meta", dtype=torch.bfloat16, **specs)
        progs.append(_dry(unet, 2, *cfg["latent_hw"], torch.bfloat16, {("cross_attention_block", "clip_text_embedding"): (77, 2048)},
                          conditions=[f"control_lora_{c['name']}" for c in specs["control"]], condition_rows=rows))
    assert [e[2] for e in progs[0].step] == [e[2] for e in progs[1].step]
    assert launches(progs[1].prologue) == launches(progs[0].prologue) + 2  # the two row copies


def test_merged_lora_mode_adds_no_step_launch():
    cfg = S.CASES["sdxl_lora_ip"]
    unet = SDXLUNet(4, device="meta", dtype=torch.bfloat16)
    specs = S.build_specs(cfg, S.key_shapes("sdxl"))
    synth.apply_adapters(unet, refiners_amd.namespace(), device="meta", dtype=torch.bfloat16, **specs)
    dev = torch.device("meta")
    io = UNetIO(x=torch.empty(2, 4, 32, 32, device=dev, dtype=torch.bfloat16), timestep=torch.empty(2, device=dev), out=torch.empty(2, 4, 32, 32, device=dev, dtype=torch.bfloat16))
    io.pooled, io.time_ids = torch.empty(2, 1280, device=dev, dtype=torch.bfloat16), torch.empty(2, 6, device=dev)
    io.tokens[("cross_attention_block", "clip_text_embedding")] = (torch.zeros(256, 2048, device=dev, dtype=torch.bfloat16), 77)
    io.tokens[("ip_adapter", "clip_image_embedding")] = (torch.zeros(128, 2048, device=dev, dtype=torch.bfloat16), 4)
    low = UNetLowering(dev, torch.bfloat16, None, "merged")
    low.lower(unet, io)
    # 981 launches of the bare tree (LayerNorm folding / Q|K|V merging are device-only decisions, off on meta) minus the 17 per-ResidualBlock
    # time projections, which ride in one batched launch
    assert len(low.step) == 981 - 17 + 1 - 9 and low.stats["lora_sites"] == 722 and low.stats["ip_sites"] == 70  # (- 9: no concat launches)
    assert low.stats["time_bias_batched"] == 17


def test_time_projection_batching_can_be_switched_off(monkeypatch):
    """REFINERS_AMD_TIME_BATCH=0: one 2-row GEMM per ResidualBlock again (the A/B lever of tools/ab_step.py); same program otherwise."""
    monkeypatch.setenv("REFINERS_AMD_TIME_BATCH", "0")
    dev = torch.device("meta")
    unet = SDXLUNet(4, device="meta", dtype=torch.bfloat16)
    io = UNetIO(x=torch.empty(2, 4, 32, 32, device=dev, dtype=torch.bfloat16), timestep=torch.empty(2, device=dev), out=torch.empty(2, 4, 32, 32, device=dev, dtype=torch.bfloat16))
    io.pooled, io.time_ids = torch.empty(2, 1280, device=dev, dtype=torch.bfloat16), torch.empty(2, 6, device=dev)
    io.tokens[("cross_attention_block", "clip_text_embedding")] = (torch.zeros(256, 2048, device=dev, dtype=torch.bfloat16), 77)
    low = UNetLowering(dev, torch.bfloat16, None, "merged")
    low.lower(unet, io)
    assert len(low.step) == 981 - 9 and "time_bias_batched" not in low.stats


def test_sd1_tree_lowers_onto_the_general_attention_kernel_for_its_head_dims():
    low = _dry(SD1UNet(4, device="meta"), 1, 32, 32, torch.float32, {("cross_attention_block", "clip_text_embedding"): (77, 768)}, pooled=False)
    assert low.stats["fallback_nodes"] == []
    names = Counter(op[2] for op in low.step)
    assert names["mi355x_attention_general"] == 32 and names["mi355x_attention"] == 0


def test_inputs_are_read_where_the_context_store_keeps_them():
    """set_context replaces the top-level dict and merges below (reference context.py:16-46): after the four set_* calls
    `unet.provider` only shows the last one; the drop-in must still see all of them."""
    unet = SDXLUNet(4, device="meta")
    unet.set_timestep(torch.tensor([981]))
    unet.set_clip_text_embedding(torch.zeros(2, 77, 2048))
    unet.set_pooled_text_embedding(torch.zeros(2, 1280))
    unet.set_time_ids(torch.zeros(2, 6))
    assert set(unet.provider.contexts["diffusion"]) == {"time_ids"}
    got = CompiledUNet(unet)._gather()
    assert got["timestep"] is not None and got["pooled"] is not None and got["time_ids"] is not None
    assert list(got["tokens"]) == [("cross_attention_block", "clip_text_embedding")]


def test_scale_and_structure_changes_move_the_tree_epoch():
    import refiners_amd.fluxion.layers as fl
    from refiners_amd.fluxion.adapters import LinearLora, LoraAdapter

    chain = fl.Chain(fl.Linear(8, 8))
    e0 = tree_epoch()
    lora = LinearLora("a", in_features=8, out_features=8, rank=2)
    adapter = LoraAdapter(chain[0], lora).inject(chain)
    e1 = tree_epoch()
    lora.scale = 0.5
    e2 = tree_epoch()
    adapter.eject()
    e3 = tree_epoch()
    assert e0 < e1 < e2 < e3


def test_prompt_encoder_lowers_without_fallbacks():
    """DoubleTextEncoder (CLIP-L + CLIP-G) -> 395 launches, 43 causal attentions, nothing left on torch."""
    from refiners_amd.engine.text import TextLowering
    from refiners_amd.latent_diffusion.prompt import DoubleTextEncoder

    dev, dt = torch.device("meta"), torch.bfloat16
    enc = DoubleTextEncoder(device="meta", dtype=dt)
    ti = lambda n: torch.empty(n, device=dev, dtype=torch.int32)  # noqa: E731
    low = TextLowering(dev, dt, None, "merged")
    low.lower_double(enc, ti(154), ti(154), ti(2), 2, 77, torch.empty(154, 2048, device=dev, dtype=dt), torch.empty(2, 1280, device=dev, dtype=dt))
    names = Counter(op[2] for op in low.step)
    assert names["mi355x_attention_general"] == 43 and names["mi355x_layernorm"] == 87 and low.stats["fallback_nodes"] == []
    assert all(op[0] is not None for op in low.step)  # no Python glue in the program


def test_image_prompt_encoder_lowers_without_fallbacks():
    """CLIPImageEncoderH + ImageProjection -> 265 launches, 32 bidirectional attentions over 257 tokens (heads of 80)."""
    from refiners_amd.clip_image import CLIPImageEncoderH
    from refiners_amd.engine.image_prompt import ImagePromptLowering
    from refiners_amd.latent_diffusion.adapters import ImageProjection

    dev, dt = torch.device("meta"), torch.bfloat16
    enc, proj = CLIPImageEncoderH(device="meta", dtype=dt), ImageProjection(1024, 2048, 4, device="meta", dtype=dt)
    low = ImagePromptLowering(dev, dt, None, "merged")
    both = torch.empty(2, 1024, device=dev, dtype=dt)
    low.lower_image_encoder(enc, torch.empty(1, 3, 224, 224, device=dev, dtype=dt), torch.empty(1, device=dev, dtype=torch.int32), both[1:])
    low.lower_image_projection(proj, both, torch.empty(8, 2048, device=dev, dtype=dt))
    names = Counter(op[2] for op in low.step)
    assert names["mi355x_attention_general"] == 32 and low.stats["fallback_nodes"] == [] and all(op[0] is not None for op in low.step)


def test_weight_prefetch_links_every_gemm_to_the_next_ones_weights():
    """native.link_weight_prefetch: launch i carries the weight span(s) of launch i+1 (wrapping around), spans stay inside
    the operand, transposed projections contribute their x-slot weights, and enable=False clears everything."""
    import ctypes as C

    from refiners_amd import native

    def op(M, N, K, w_ptr, x_ptr=0x1000, ldw=None, weight_is_x=False, conv=0, ksize=1):
        a = native.GemmArgs()
        a.dtype, a.M, a.N, a.nseg, a.conv = 1, M, N, 1, conv
        a.seg[0].k, a.seg[0].ksize, a.seg[0].w, a.seg[0].ldw = K, ksize, w_ptr, ldw or K * ksize * ksize
        a.seg[0].x, a.seg[0].ldx = x_ptr, K
        a.weight_is_x = weight_is_x
        return (object(), (C.byref(a),), "mi355x_gemm(conv)" if conv else "mi355x_gemm", ())

    ops = [op(2048, 1280, 1280, 0x10000000), op(2048, 10240, 1280, 0x20000000), ((None, lambda: None, "python", ())),
           op(1280, 2048, 1280, 0x50000000, x_ptr=0x30000000, weight_is_x=True), op(2048, 1280, 320, 0x40000000, conv=1, ksize=3),
           op(2048, 640, 640, 0x60000000, ldw=2048)]
    st = native.link_weight_prefetch(ops)
    g = [e[1][0]._obj for e in ops if e[0] is not None]
    assert st["linked"] == 5 and st["launches"] == 5
    assert (g[0].prefetch[0], g[0].prefetch_bytes[0]) == (0x20000000, 10240 * 1280 * 2)
    assert (g[1].prefetch[0], g[1].prefetch_bytes[0]) == (0x30000000, 1280 * 1280 * 2)        # the x slot holds the parameters there
    assert (g[2].prefetch[0], g[2].prefetch_bytes[0]) == (0x40000000, 1280 * 9 * 320 * 2)     # conv: N x (ksize^2 * C)
    assert (g[3].prefetch[0], g[3].prefetch_bytes[0]) == (0x60000000, (639 * 2048 + 640) * 2)  # column slice: last row to K only
    assert (g[4].prefetch[0], g[4].prefetch_bytes[0]) == (0x10000000, 1280 * 1280 * 2)        # wraps around for the next replay
    assert all(32 <= a.prefetch_blocks <= 128 for a in g)
    import os

    os.environ["REFINERS_AMD_PF_BLOCKS"] = "16-64"  # the A/B lever on the number of prefetch workgroups
    try:
        native.link_weight_prefetch(ops)
        assert [a.prefetch_blocks for a in g] == [64, 25, 57, 20, 25]  # ceil(bytes / 128 KB) clamped to [16, 64]
    finally:
        del os.environ["REFINERS_AMD_PF_BLOCKS"]
    native.link_weight_prefetch(ops, enable=False)
    assert all(not a.prefetch[0] and a.prefetch_blocks == 0 for a in g)


def test_kblocked_wrapper_round_trips_and_is_what_prefetch_sees():
    """native.KBlocked: [N, K] -> [K / block][N][block] and back; a recorded launch with a K-blocked weight reports N * K bytes."""
    import ctypes as C

    from refiners_amd import native

    for dtype, blk in ((torch.bfloat16, 64), (torch.float32, 32)):
        w = torch.randn(37, 5 * blk).to(dtype)
        kb = native.KBlocked(w)
        assert kb.shape == (37, 5 * blk) and tuple(kb.t.shape) == (5, 37, blk) and kb.dtype == dtype
        assert torch.equal(kb.dense(), w)
        assert torch.equal(kb.t[2, 11], w[11, 2 * blk : 3 * blk])  # K block 2 of row 11
        adopted = native.KBlocked.adopt(kb.t.reshape(-1), 37, 5 * blk)
        assert torch.equal(adopted.dense(), w)
    a = native.GemmArgs()
    a.dtype, a.M, a.N, a.nseg, a.conv = 1, 2048, 1280, 1, 0
    a.seg[0].k, a.seg[0].ksize, a.seg[0].w, a.seg[0].ldw, a.seg[0].kblocked = 5120, 1, 0x1000, 5120, 1
    assert native.weight_spans(a) == [(0x1000, 1280 * 5120 * 2)]


# ---- the section 8(b) error convention: unknown sub-trees fall back, they never make the call fail (fluxion/layers/chain.py:226-243) -------------
class _Scale(fl.Module):
    """An out-of-scope, context-free layer someone appended to a UNet stage."""

    def __init__(self, s: float) -> None:
        super().__init__()
        self.s = s

    def forward(self, x):
        return x * self.s


def _bare_io(dev, dtype=torch.float32):
    io = UNetIO(x=torch.empty(2, 4, 32, 32, device=dev, dtype=dtype), timestep=torch.empty(2, device=dev), out=torch.empty(2, 4, 32, 32, device=dev, dtype=dtype))
    io.pooled, io.time_ids = torch.empty(2, 1280, device=dev, dtype=dtype), torch.empty(2, 6, device=dev)
    io.tokens[("cross_attention_block", "clip_text_embedding")] = (torch.zeros(256, 2048, device=dev, dtype=dtype), 77)
    return io


def test_a_context_free_unknown_layer_becomes_a_torch_node_of_the_program():
    from refiners_amd.latent_diffusion.blocks import ResidualBlock

    unet = SDXLUNet(4, device="meta")
    unet.layer(("DownBlocks", 1), fl.Chain).insert_after_type(ResidualBlock, _Scale(0.9))
    low = UNetLowering(torch.device("meta"), torch.float32)
    low.lower(unet, _bare_io(torch.device("meta")))
    assert low.stats["fallback_nodes"] == ["_Scale"]
    assert sum(1 for e in low.step if e[0] is None and e[2] == "torch:_Scale") == 1


def test_an_unknown_layer_that_needs_the_context_store_refuses_the_whole_tree():
    """FreeU's concatenator shape (latent_diffusion/freeu.py:57-72): Concatenate(backbone features, UseContext(unet.residuals)[n] -> filter).  The
    recorded program keeps the residual slots to itself, so this node cannot run inside it: lowering raises Unsupported and CompiledUNet
    runs the stock forward (tests/test_engine_gpu.py, tests/test_reference_tree_gpu.py)."""
    from refiners_amd.engine.packing import Unsupported
    from refiners_amd.latent_diffusion.blocks import ResidualConcatenator

    class SkipFilter(fl.Concatenate):
        def __init__(self, n: int) -> None:
            super().__init__(fl.Identity(), fl.Chain(fl.UseContext(context="unet", key="residuals").compose(lambda r: r[n]), fl.Lambda(lambda t: t * 0.5)), dim=1)

    unet = SDXLUNet(4, device="meta")
    block = unet.layer(("UpBlocks", 0), fl.Chain)
    block.replace(block.ensure_find(ResidualConcatenator), SkipFilter(-2))
    low = UNetLowering(torch.device("meta"), torch.float32)
    with pytest.raises(Unsupported, match="SkipFilter.*UseContext"):
        low.lower(unet, _bare_io(torch.device("meta")))


def test_groupnorm_takes_its_statistics_from_the_producing_launch_where_there_is_one(monkeypatch):
    """ResidualBlock = GroupNorm -> SiLU -> Conv2d -> GroupNorm -> SiLU -> Conv2d (latent_diffusion/unet.py:6-51): the second GroupNorm's input is
    the first convolution's output, so that launch writes the column statistics (mi355x_gemm_args.colstats_out) and the GroupNorm call carries
    them (two kernels instead of three); the first GroupNorm's input comes from outside (no producer in this program): statistics pass.
    The block's output carries its statistics for whoever normalises it next (CrossAttentionBlock2d's GroupNorm here) -- in its OWN buffer, so
    they are still the right ones when the tensor is normalised again after a later launch produced another tensor of the same shape (a skip
    tensor inside a ResidualConcatenator's output).  Recorded on the CPU device: nothing is launched."""
    from refiners_amd.engine.lowering_blocks import BlockLowering
    from refiners_amd.engine.packing import Act
    from refiners_amd.engine.unet_lowering import UNetContext
    from refiners_amd.latent_diffusion.blocks import CrossAttentionBlock2d, ResidualBlock

    torch.manual_seed(0)
    tree = fl.Chain(ResidualBlock(64, 128), CrossAttentionBlock2d(channels=128, context_embedding_dim=64, context_key="clip_text_embedding", num_attention_heads=2, use_bias=False, use_linear_projection=True))
    B, H, W = 2, 8, 8

    def lower(flag: str):
        monkeypatch.setenv("REFINERS_AMD_GN_STATS", flag)
        low = BlockLowering(torch.device("cpu"), torch.float32)
        ctx = UNetContext(low, B)
        ctx.text[("cross_attention_block", "clip_text_embedding")] = (torch.zeros(B * 64, 64), 7)
        with low.in_step():
            a = Act(low.pool.get(B * H * W, 64), B, H, W)
            a1 = low.residual_block(tree[0], a, ctx)
            assert (a1.cs is not None) == (flag == "1")
            out = low.cross_attention_2d(tree[1], a1, ctx)
            stale = low.groupnorm(a1, kids_of(tree[1])[0], silu=False) if flag == "1" else None  # a1 again, after the transformer's proj_out produced a tensor of the same shape
        return low, out, stale

    def kids_of(m):
        return list(list(m._modules.values())[0]._modules.values())

    low, out, stale = lower("1")
    gns = [e[1][0]._obj for e in low.step if e[2] == "mi355x_groupnorm"]
    assert [bool(g.colstats) for g in gns] == [False, True, True, True]  # GN1 (outside input), GN2 (conv1), transformer GN (conv2 + shortcut), a1 once more
    assert gns[2].colstats == gns[3].colstats != gns[1].colstats and low.stats["gn_from_producer"] == 3 and out.cs is not None
    producers = [e[1][0]._obj for e in low.step if e[2].startswith("mi355x_gemm") and e[1][0]._obj.colstats_out]
    assert len(producers) == 3 and {int(p.N) for p in producers} == {128} and len({int(p.colstats_out) for p in producers}) == 3  # conv1, conv2, proj_out: a buffer each
    low0, _, _ = lower("0")
    assert all(not e[1][0]._obj.colstats for e in low0.step if e[2] == "mi355x_groupnorm") and "gn_from_producer" not in low0.stats
    assert [e[2] for e in low0.step] == [e[2] for e in low.step][:-1]  # same program otherwise


def test_a_residual_block_reads_a_concatenation_that_was_never_written(monkeypatch):
    """ResidualConcatenator -> ResidualBlock (latent_diffusion/unet.py:69-85): the block's first GroupNorm takes the two parts as two sources (with both
    producers' column statistics where they exist), its 1x1 shortcut as two K segments of conv2 -- no concat launch, the skip tensor stays where it is.
    REFINERS_AMD_CAT_FUSE=0 writes the concatenation again.  Recorded on the CPU device: nothing is launched."""
    from refiners_amd.engine.lowering_blocks import BlockLowering
    from refiners_amd.engine.packing import Act, CatAct
    from refiners_amd.engine.unet_lowering import UNetContext
    from refiners_amd.latent_diffusion.blocks import ResidualBlock

    torch.manual_seed(0)
    first, block = ResidualBlock(64, 64), ResidualBlock(64 + 128, 128)
    B, H, W = 2, 8, 8

    def lower(flag: str):
        monkeypatch.setenv("REFINERS_AMD_CAT_FUSE", flag)
        low = BlockLowering(torch.device("cpu"), torch.float32)
        ctx = UNetContext(low, B)
        with low.in_step():
            x = low.residual_block(first, Act(low.pool.get(B * H * W, 64), B, H, W), ctx)      # carries its producer's statistics
            skip = Act(low.pool.get(B * H * W, 128), B, H, W)                                  # a skip tensor from somewhere else: none
            out = low.residual_block(block, CatAct(x, skip), ctx)
        return low, out

    low, out = lower("1")
    names = [e[2] for e in low.step]
    assert "mi355x_concat2" not in names and "concat_materialised" not in low.stats and out.C == 128
    gn = [e[1][0]._obj for e in low.step if e[2] == "mi355x_groupnorm"][2]  # the second block's first GroupNorm
    assert gn.C == 192 and gn.C1 == 64 and gn.x2 and not gn.colstats  # two sources; the skip has no statistics, so the pass over both parts runs
    conv2 = [e[1][0]._obj for e in low.step if e[2] == "mi355x_gemm(conv)"][-1]
    assert conv2.nseg == 3 and [int(conv2.seg[i].k) for i in range(3)] == [128, 64, 128] and [int(conv2.seg[i].ksize) for i in range(3)] == [3, 1, 1]
    low0, _ = lower("0")
    names0 = [e[2] for e in low0.step]
    assert names0.count("mi355x_concat2") == 1 and low0.stats["concat_materialised"] == 1 and len(names0) == len(names) + 1
    conv2 = [e[1][0]._obj for e in low0.step if e[2] == "mi355x_gemm(conv)"][-1]
    assert conv2.nseg == 2 and int(conv2.seg[1].k) == 192


def test_tuning_table_lookup_rules(tmp_path, monkeypatch):
    """engine/tuning.py: exact signature first; a `...lora` launch without its own entry takes the un-adapted launch's tile if a LoRA kernel exists for it
    (4-wave tiles, two LDS stages); REFINERS_AMD_TUNING_TABLE selects a candidate table; and the two Q|K|V^T launches of the SDXL step with live LoRAs
    sit on the 128-column tile, the only one wide enough for the three groups' t columns (csrc/gemm_kernel.cuh, GemmP::lora_tt)."""
    import json

    from refiners_amd.engine import tuning

    monkeypatch.delenv("REFINERS_AMD_TUNING_TABLE", raising=False)
    monkeypatch.setattr(tuning, "_table", None)
    monkeypatch.setattr(tuning, "enabled", True)
    for sig in ("gemm:bf16:2048x3840x1280:s1:T2560lnlora", "gemm:bf16:8192x1920x640:s1:T1280lnlora"):
        assert tuning.lookup(sig) == (1, 2), sig
    assert tuning.lookup("gemm:bf16:1x1x1:s1:") == (0, 0) and tuning.lookup("gemm:bf16:1x1x1:s1:", stages=3) == (0, 3)
    cand = tmp_path / "cand.json"
    cand.write_text(json.dumps({"choices": {"gemm:bf16:64x64x64:s1:": [6, 3], "gemm:bf16:128x64x64:s1:": [2, 4], "gemm:bf16:128x64x64:s1:lora": [3, 2]}}))
    monkeypatch.setenv("REFINERS_AMD_TUNING_TABLE", str(cand))
    monkeypatch.setattr(tuning, "_table", None)
    assert tuning.summary()["table"] == "cand.json" and tuning.summary()["entries"] == 3
    assert tuning.lookup("gemm:bf16:64x64x64:s1:") == (6, 3)
    assert tuning.lookup("gemm:bf16:64x64x64:s1:lora") == (1, 2)      # the 8-wave tile has no LoRA kernel: 128 x 128, two stages
    assert tuning.lookup("gemm:bf16:128x64x64:s1:lora") == (3, 2)     # its own entry wins over the un-adapted launch's
    monkeypatch.setattr(tuning, "enabled", False)
    assert tuning.lookup("gemm:bf16:64x64x64:s1:") == (0, 0)
    monkeypatch.setattr(tuning, "_table", None)  # (the next user re-reads the product table)


def test_every_tuned_signature_is_a_launch_of_the_lowered_step(monkeypatch):
    """engine/tuning_gfx950.json is keyed by launch signature; a lowering change that alters a signature (the concat-shortcut split turned two `:s2:`
    convolutions into `:s3:` in round 4) silently orphans its entry.  Every key of the table must be carried by a launch of the bare SDXL step at the
    batch sizes the table was measured on (CFG pair = UNet batch 2; four images per GPU = batch 8; eight = batch 16, round 6), 128 x 128 latents, bf16."""
    import json

    from refiners_amd import native
    from refiners_amd.engine import tuning

    monkeypatch.setattr(tuning, "enabled", False)  # (signatures do not depend on the choice; this keeps tile-8 scratch off the meta device)
    doc = json.loads(tuning.TABLE_PATH.read_text())
    elsewhere = {k for keys in doc.get("other_workloads", {}).values() for k in keys}  # entries measured on another program (the SAM ViT-H encoder): listed by name
    assert elsewhere <= set(doc["choices"])
    table = {k: v for k, v in doc["choices"].items() if k not in elsewhere}
    seen = set()
    for B in (2, 8, 16):
        unet = SDXLUNet(4, device="meta", dtype=torch.bfloat16)
        low = _dry(unet, B, 128, 128, torch.bfloat16, {("cross_attention_block", "clip_text_embedding"): (77, 2048)})
        seen |= {native.gemm_signature(e[1][0]._obj).rsplit(":", 1)[0] for e in low.step if e[0] is not None and e[2].startswith("mi355x_gemm")}
    # (compared without the flag field: on the meta device every pointer is 0, so the flags that mean "this pointer is set" -- ln, st, T -- are missing)
    # not visible in a meta-device lowering: the merged Q|K|V^T launches (flag T: merging is decided on device tensors) and the two-row time-embedding GEMM
    # (the timestep table replaces it wherever the solver's timesteps are known)
    skip = lambda k: "T" in k.rsplit(":", 1)[1] or int(k.split(":")[2].split("x")[0]) <= 8  # noqa: E731
    orphans = sorted(k for k in table if not skip(k) and k.rsplit(":", 1)[0] not in seen)
    assert not orphans, orphans
