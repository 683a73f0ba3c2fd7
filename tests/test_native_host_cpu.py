"""Host-side rules of refiners_amd.native that need no GPU: the in-launch LoRA's flag arena and its error words (what the engines poll at their host
sync points), the tile / split-K precedence of a launch (caller's explicit choice > measured table > heuristic), bench.py's CPU-count helpers."""
import json

import pytest
import torch

from refiners_amd import native
from refiners_amd.engine import tuning


def test_lora_flag_arena_sites_do_not_overlap_and_error_words_are_found():
    ls = native.LoraSync(torch.device("cpu"))
    assert ls.pending() is None  # no site yet: nothing to look at, no host sync
    sites = [(1, 2048), (3, 2048), (1, 300), (2, 8192), (1, 33)]
    views = [ls.flags(g, m) for g, m in sites]
    spans = []
    for (g, m), v in zip(sites, views):
        assert v.numel() == g * ((m + 31) // 32) + 1 and v.dtype == torch.int32 and int(v.abs().sum()) == 0
        assert v.data_ptr() % 16 == 0
        spans.append((v.data_ptr(), v.data_ptr() + 4 * v.numel()))
    spans.sort()
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), "flag arrays of two sites overlap"
    assert not bool(ls.pending())
    # a flag word carrying an epoch is NOT an error; the word behind a site's last flag is
    views[1][5] = 7
    views[3][0] = 7
    assert not bool(ls.pending())
    views[2][-1] = 1
    assert bool(ls.pending())
    with pytest.raises(native.NativeError):
        ls.check()
    assert not bool(ls.pending()) and int(views[1][5]) == 7  # the raise cleared the error words only
    ls.check()


def test_lora_flag_arena_grows_in_chunks_with_stable_addresses():
    ls = native.LoraSync(torch.device("cpu"))
    ls.CHUNK = 64  # (instance attribute: small chunks for the test)
    first = ls.flags(1, 32 * 40)
    ptr = first.data_ptr()
    more = [ls.flags(1, 32 * 40) for _ in range(5)]
    assert len(ls.chunks) >= 3 and first.data_ptr() == ptr  # earlier sites keep their storage: recorded launches hold the addresses
    more[-1][-1] = 1
    assert bool(ls.pending())
    ls.clear_errors()
    assert not bool(ls.pending())
    ls.reset()
    assert len(ls.chunks) == 1 and ls.pending() is None and int(ls.chunks[0].abs().sum()) == 0


def _args(M=2048, N=1280, K=11520, conv=1):
    a = native.GemmArgs()
    a.dtype, a.M, a.N, a.nseg, a.conv = native.MI355X_BF16, M, N, 1, conv
    a.seg[0].k, a.seg[0].ksize = K // 9 if conv else K, 3 if conv else 1
    return a


def test_explicit_tile_and_split_are_the_callers_but_a_heuristic_split_yields_to_the_table(monkeypatch):
    ws = torch.zeros(16)
    sig = native.gemm_signature(_args())
    monkeypatch.setattr(tuning, "enabled", True)
    monkeypatch.setattr(tuning, "_table", {sig: (9, 0)})
    # a test or probe that asks for split-K on a tabled shape gets split-K (round-5 advisor: the table used to override it)
    a = _args()
    native._fill_split(a, 1, 3, ws)
    assert (a.tile, a.ksplit) == (1, 3) and a.ws == ws.data_ptr()
    # Lowering.conv's three-way split is a heuristic: the table's 8-wave tile replaces the TILE, ksplit / ws stay for mi355x_gemm to fall back on
    a = _args()
    native._fill_split(a, 1, 3, ws, table_may_replace_split=True)
    assert (a.tile, a.ksplit) == (9, 3) and a.ws == ws.data_ptr()
    # no choice at all: the table
    a = _args()
    native._fill_split(a, 0, 1, None)
    assert (a.tile, a.ksplit) == (9, 1) and not a.ws
    # a shape the table does not know: the library heuristic (tile 0)
    a = _args(N=640)
    native._fill_split(a, 0, 1, None)
    assert a.tile == 0
    monkeypatch.setattr(tuning, "_table", {sig: (1, 2)})
    a = _args()
    native._fill_split(a, 1, 3, ws, table_may_replace_split=True)
    assert (a.tile, a.ksplit) == (1, 3)  # (a 4-wave entry does not touch a split launch)


def test_bench_cpu_count_helpers():
    import bench

    phys, usable = bench.physical_cores(), bench.usable_cpus()
    assert 1 <= usable <= (bench.os.cpu_count() or 1) and phys >= 1
