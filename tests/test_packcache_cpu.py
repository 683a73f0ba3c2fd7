"""PackCache hand-over rules (refiners_amd.parallel.broadcast_packs): which entries a receiving rank makes itself, which it only allocates,
and that what it allocates has the shapes / dtypes / devices of the source's values.  Single process: the manifest is passed by hand."""
import torch

from refiners_amd import native
from refiners_amd.engine.packing import LoraPack, PackCache, _leaves


def _fill(cache: PackCache, w: torch.Tensor, b: torch.Tensor, calls: list) -> None:
    def mk(tag, fn):
        def run():
            calls.append(tag)
            return fn()

        return run

    cache.get(("as_is",) + PackCache.ident(w), mk("as_is", lambda: w))                                        # the leaf's own storage
    cache.get(("index", 8), mk("index", lambda: torch.arange(8)))                                            # a function of the key alone
    cache.get(("kblocked",) + PackCache.ident(w), mk("kblocked", lambda: native.KBlocked(w)))                 # fresh, packed
    cache.get(("fold",) + PackCache.ident(w, b), mk("fold", lambda: (w * 2, w.float().sum(1), b)))            # fresh tensors + the leaf itself
    cache.get(("lora",) + PackCache.ident(w), mk("lora", lambda: LoraPack(w[:16].clone(), w[:, :16].clone())))  # a pack object with empty slots
    cache.get(("scalar",) + PackCache.ident(b), mk("scalar", lambda: 3))                                      # no tensor inside


def test_manifest_separates_what_travels_from_what_every_rank_makes():
    torch.manual_seed(0)
    w, b = torch.randn(32, 64), torch.randn(32)
    src, calls = PackCache(), []
    _fill(src, w, b, calls)
    man = src.manifest()
    assert [m[0] for m in man] == ["alias", "alias", "recv", "recv", "recv", "alias"]
    assert src.made == 6 and len(calls) == 6
    # the receiver holds equal leaves (the weight broadcast ran before); it lowers with the manifest adopted
    w2, b2 = w.clone(), b.clone()
    dst, calls2 = PackCache(), []
    dst.build_device = torch.device("cpu")
    dst.adopt(man)
    _fill(dst, w2, b2, calls2)
    assert calls2 == ["as_is", "index", "scalar"] and dst.made == 3  # no K-blocking / folding / stacking on the receiver
    got, want = dst.leaves(man), src.leaves(man)
    assert [(tuple(t.shape), t.dtype, t.device.type) for t in got] == [(tuple(t.shape), t.dtype, t.device.type) for t in want]
    kb = dst.store[dst.order[2]]
    assert isinstance(kb, native.KBlocked) and kb.shape == (32, 64)
    lp = dst.store[dst.order[4]]
    assert isinstance(lp, LoraPack) and lp.a_kb is None and lp.conv is None
    # the "broadcast": copy in place, as parallel.broadcast_tensors(..., repoint=False) does on a receiver
    for t, s in zip(got, want):
        t.copy_(s)
    assert torch.equal(kb.dense(), w) and torch.equal(dst.store[dst.order[3]][1], w.sum(1))
    # afterwards the cache answers normally again (a later re-lowering packs locally)
    dst.get(("later",) + PackCache.ident(b2), lambda: b2 + 1)
    assert dst.made == 4


def test_a_longer_request_sequence_than_the_manifest_is_refused():
    w, b = torch.randn(32, 64), torch.randn(32)
    src = PackCache()
    src.get(("as_is",) + PackCache.ident(w), lambda: w)
    dst = PackCache()
    dst.adopt(src.manifest())
    dst.get(("as_is",) + PackCache.ident(w), lambda: w)
    try:
        dst.get(("more",) + PackCache.ident(b), lambda: b * 2)
    except AssertionError as exc:
        assert "fewer packed weights" in str(exc)
    else:
        raise AssertionError("an unexpected extra entry must not be answered silently")
    assert _leaves(3) == []


def test_a_warm_cache_hands_over_only_what_the_current_lowering_created():
    """Round-3 advisor finding: manifest() used to walk every entry ever created while a receiver consumed from position 0.  After an earlier
    lowering on both ranks (a first hand-over, a LoRA scale change) the second hand-over must publish / consume the NEW entries only."""
    torch.manual_seed(1)
    w, b = torch.randn(32, 64), torch.randn(32)
    w2, b2 = w.clone(), b.clone()
    src, dst = PackCache(), PackCache()
    dst.build_device = torch.device("cpu")
    # first hand-over
    src.mark()
    _fill(src, w, b, [])
    man = src.manifest()
    dst.adopt(man)
    _fill(dst, w2, b2, [])
    for t, s in zip(dst.leaves(man), src.leaves(man)):
        t.copy_(s)
    # a scale change re-lowers: every old entry hits, one new merged weight is made
    calls = []
    src.mark()
    _fill(src, w, b, calls)
    src.get(("merged", 0.5) + PackCache.ident(w), lambda: w * 0.5)
    man2 = src.manifest()
    assert calls == [] and len(man2) == 1 and man2[0][0] == "recv"
    dst.adopt(man2)
    _fill(dst, w2, b2, calls)
    got = dst.get(("merged", 0.5) + PackCache.ident(w2), lambda: calls.append("merged") or w2 * 0.5)
    assert calls == [] and tuple(got.shape) == (32, 64)
    (leaf,), (want,) = dst.leaves(man2), src.leaves(man2)
    leaf.copy_(want)
    assert torch.equal(dst.store[dst.order[-1]], w * 0.5)


def test_a_receiver_that_is_out_of_step_is_refused_before_it_builds_anything():
    w, b = torch.randn(32, 64), torch.randn(32)
    src = PackCache()
    src.get(("kblocked",) + PackCache.ident(w), lambda: native.KBlocked(w))
    dst = PackCache()
    dst.build_device = torch.device("cpu")
    dst.adopt(src.manifest())
    try:
        dst.get(("fold",) + PackCache.ident(w, b), lambda: (w * 2, b))  # same position, another entry: used to get the K-blocked description
    except AssertionError as exc:
        assert "out of step" in str(exc)
    else:
        raise AssertionError("a mismatching key must not be answered from the manifest")
