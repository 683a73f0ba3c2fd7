"""Host-side tables that decide what runs / what is reported: the measured tile table's fallback rules (refiners_amd/engine/tuning.py) and
which committed counter pass bench.py quotes beside its live roofline (bench.pmc_mfma_file)."""
import json

import bench
from refiners_amd.engine import tuning


def test_tile_table_fallbacks(monkeypatch):
    monkeypatch.setattr(tuning, "_table", {"gemm:bf16:8x8x8:s1:": (6, 2), "gemm:bf16:8x8x8:s1:ln": (3, 4), "gemm:bf16:9x9x9:s1:lora": (2, 2)})
    monkeypatch.setattr(tuning, "enabled", True)
    assert tuning.lookup("gemm:bf16:8x8x8:s1:") == (6, 2)
    assert tuning.lookup("gemm:bf16:1x1x1:s1:", stages=3) == (0, 3)  # unknown shape: the library heuristic, the caller's stage hint kept
    # a LoRA launch without its own entry takes the un-adapted launch's tile, restricted to what the LoRA kernels exist for
    assert tuning.lookup("gemm:bf16:8x8x8:s1:lora") == (1, 2)    # 8-wave tile -> 128x128, two stages
    assert tuning.lookup("gemm:bf16:8x8x8:s1:lnlora") == (3, 2)  # deeper ring -> two stages
    assert tuning.lookup("gemm:bf16:9x9x9:s1:lora") == (2, 2)    # its own measured entry wins
    # the 8-wave loop (7 = whole tiles, 8 = stream-K) takes the LoRAs of a plain one-segment GEMM with one column group, as whole tiles
    monkeypatch.setattr(tuning, "_table", {"gemm:bf16:8x8x8:s1:gegluln": (7, 0), "gemm:bf16:8x8x8:s1:st": (8, 0), "gemm:bf16:8x24x8:s1:T16ln": (7, 0),
                                           "conv:bf16:8x8x72:s1:": (8, 0), "gemm:bf16:8x8x16:s2:": (7, 0)})
    assert tuning.lookup("gemm:bf16:8x8x8:s1:geglulnlora") == (7, 0)
    assert tuning.lookup("gemm:bf16:8x8x8:s1:stlora") == (7, 0)
    monkeypatch.setitem(tuning._table, "gemm:bf16:8x8x8:s1:ln", (9, 0))
    assert tuning.lookup("gemm:bf16:8x8x8:s1:lnlora") == (9, 0)  # (the 192-row tile of the same loop has the LoRA instance too)
    assert tuning.lookup("gemm:bf16:8x24x8:s1:T16lnlora") == (1, 2)  # a transposed column group (Q | K | V^T: three LoRA groups): the 4-wave kernel
    assert tuning.lookup("conv:bf16:8x8x72:s1:lora") == (1, 2)
    assert tuning.lookup("gemm:bf16:8x8x16:s2:lora") == (1, 2)
    monkeypatch.setattr(tuning, "lora_g8", False)
    assert tuning.lookup("gemm:bf16:8x8x8:s1:stlora") == (1, 2)
    monkeypatch.setattr(tuning, "enabled", False)
    assert tuning.lookup("gemm:bf16:8x8x8:s1:") == (0, 0)


def test_the_shipped_table_only_names_tiles_the_library_has():
    doc = json.loads(tuning.TABLE_PATH.read_text())
    for sig, (tile, stages) in doc["choices"].items():
        assert tile in (0, 1, 2, 3, 4, 6, 7, 8, 9) and stages in (0, 2, 3, 4), (sig, tile, stages)  # (7 / 8 / 9: the 8-wave loop, whole 256-row tiles / stream-K / whole 192-row tiles)
        assert not (tile == 9 and "T" in sig.rsplit(":", 1)[1]), sig  # (the 192-row tile has no transposed form)
        if sig.endswith("lora"):
            assert (tile in (1, 2, 3, 4) and stages == 2) or (tile in (7, 9) and sig.startswith("gemm") and ":s1:" in sig and "T" not in sig.split(":")[-1]), sig


def test_bench_quotes_the_latest_counter_pass_deterministically(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    fam = lambda ns, util: {"families": {"mi355x_gemm": {"mfma_util": util, "mfma_util_by_duration": util + 0.05, "SQ_VALU_MFMA_BUSY_CYCLES": 1.0, "DURATION_NS": ns}},  # noqa: E731
                            "scope": "step program only: 3 full replay(s) of the 686 recorded launches", "classes": {}}
    (prof / "r02_s_pmc_mfma.json").write_text(json.dumps(fam(1.0e6, 0.9)))   # an older round never wins
    (prof / "r03_q_pmc_mfma.json").write_text(json.dumps(fam(5.0e7, 0.15)))
    (prof / "r03_r2_pmc_mfma.json").write_text(json.dumps(fam(7.0e7, 0.11)))  # later by name: quoted, slower box or not (no best-of-N selection)
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    assert bench.pmc_mfma_file("mi355x_gemm").name == "r03_r2_pmc_mfma.json"
    got = bench.pmc_mfma_util("mi355x_gemm", family_tflop_per_step=8.0)
    assert got["source"].endswith("r03_r2_pmc_mfma.json") and abs(got["mfma_util"] - 0.16) < 1e-9 and got["mfma_util_over_gui_active"] == 0.11
    assert abs(got["flop_frac_of_that_run"] - 8.0 * 3 / 7.0e-2 / bench.PEAK_BF16_TFLOPS) < 1e-4
    assert bench.pmc_mfma_util("no_such_family") is None


def test_attention_key_order_puts_a_lanes_eight_keys_in_one_chunk():
    """csrc/common.cuh: k_row_key -- the property the attention kernels rely on, checked on the expression itself (read from the source):
    the S^T blocks 2s and 2s+1 hand lane group g the rows 4g..4g+3 of each; with the K tile's rows loaded in k_row_key order those eight
    P^T values belong to the keys 32s + 8g .. + 7 IN ORDER, i.e. to the 16-byte chunk 4s + g of a bf16 V^T row, and the map is a bijection
    of the tile's 64 keys (nothing is dropped or doubled, so the softmax sums do not change)."""
    import re
    from pathlib import Path

    src = (Path(__file__).resolve().parent.parent / "refiners_amd" / "csrc" / "common.cuh").read_text()
    expr = re.search(r"constexpr int k_row_key\(int row\) \{ return (.+?); \}", src).group(1)
    key = lambda row: eval(expr, {"row": row})  # noqa: E731,S307 -- integer shifts / masks only: the C expression is valid Python
    assert sorted(key(r) for r in range(64)) == list(range(64))
    for s in range(2):
        for g in range(4):
            got = [key(16 * (2 * s) + 4 * g + r) for r in range(4)] + [key(16 * (2 * s + 1) + 4 * g + r) for r in range(4)]
            assert got == list(range(32 * s + 8 * g, 32 * s + 8 * g + 8)), (s, g, got)
            assert (got[0] * 2) // 16 == 4 * s + g  # byte offset of the first key in a bf16 row / 16 = the chunk the kernel reads
