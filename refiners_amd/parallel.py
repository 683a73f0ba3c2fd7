"""Multi-GPU layer of the hot path: one process per GPU, independent prompts per rank, no per-step collective.

The reference has no multi-device support at all; the sampling loop shards naturally because
every image is an independent 50-step trajectory (batch invariance is one of the reference's own tests,
tests/e2e/test_diffusion.py:1539-1597).  So the only collectives are
  * ONE broadcast of UNet + adapter weights from rank 0 at start-up (RCCL over xGMI; flattened into a few large
    buckets so that each launch moves hundreds of MB and RCCL can keep all links of the ring busy), and
  * an optional gather of the final latents (128 KiB per image).
Works with backend "nccl" (= RCCL on ROCm) on GPUs and "gloo" on CPU (tests).
"""
from __future__ import annotations

import os
from typing import Any, Iterable, Sequence

import torch
import torch.distributed as dist
from torch import Tensor


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's environment; initialises the default process group if needed."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # REFINERS_AMD_DIST_BACKEND=gloo: rehearsal of the N > 1 flow where RCCL cannot run (several ranks on ONE GPU, see bench.py)
            backend = os.environ.get("REFINERS_AMD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


moved = {"bytes": 0, "collectives": 0}  # running totals of broadcast_tensors on this rank (bench.py reads the deltas: GB/s over xGMI = bytes / seconds)


def _staged(t: Tensor) -> bool:
    """gloo moves host memory: device tensors are staged through the host (the one-GPU rehearsal of bench.py; RCCL takes them as they are)."""
    return t.device.type != "cpu" and dist.get_backend() == "gloo"


def _broadcast(t: Tensor, src: int) -> None:
    if not _staged(t):
        dist.broadcast(t, src=src)
        return
    h = t.cpu()
    dist.broadcast(h, src=src)
    if dist.get_rank() != src:
        t.copy_(h)


def all_gather(t: Tensor) -> list[Tensor]:
    """Every rank's `t` (same shape and dtype everywhere), on t's device."""
    h = t.cpu() if _staged(t) else t
    parts = [torch.empty_like(h) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, h)
    return [p.to(t.device) for p in parts]


def broadcast_tensors(tensors: Iterable[Tensor], src: int = 0, bucket_bytes: int = 512 << 20, align: int = 256, repoint: bool = True) -> int:
    """In-place broadcast of many tensors through ONE flat arena per (dtype, device), sent in large bucket-sized pieces.

    No staging copies on the receivers: every tensor's storage is RE-POINTED into the arena (`t.data = arena[off : off + n]`,
    256-byte aligned so the kernels' 16-byte vector loads stay legal) and RCCL writes straight into it; only the source
    rank pays one copy-in pass.  The arena stays the weights' home afterwards (one allocation instead of thousands, which is
    also what a packed / direct-to-GPU checkpoint load wants).  Each `dist.broadcast` moves up to `bucket_bytes`; RCCL
    pipelines a large message over all its channels / xGMI links by itself, so fewer, larger messages are the lever here
    (SURVEY.md section 8(e)).  `repoint=False`: the tensors keep their storage (somebody already holds their addresses -- the packed
    weights of a lowered program) and the receivers copy out of the arena instead.  Returns the number of collective launches."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    rank = dist.get_rank()
    tensors = list(tensors)
    # every rank must walk the SAME list (count, shapes, dtypes, device kinds): arena offsets and the number of collectives follow
    # from it, so a mismatch would hang or silently shift weights.  Compared against the source's manifest before anything moves.
    check_same_on_all_ranks([(tuple(t.shape), str(t.dtype), t.device.type) for t in tensors], "broadcast_tensors: tensor list", src)
    groups: dict[tuple[torch.dtype, torch.device], list[Tensor]] = {}
    for t in tensors:
        groups.setdefault((t.dtype, t.device), []).append(t)
    launches = 0
    for (dtype, device), items in groups.items():
        es = items[0].element_size()
        step = max(align // es, 1)
        offs, total = [], 0
        for t in items:
            offs.append(total)
            total += (t.numel() + step - 1) // step * step
        arena = torch.empty(total, dtype=dtype, device=device)
        for t, off in zip(items, offs):
            view = arena[off : off + t.numel()].view(t.shape)
            if rank == src:
                view.copy_(t.detach())
            if repoint:
                t.data = view  # the parameter now lives in the arena (no copy-back after the collective)
        per = max(bucket_bytes // es, 1)
        for lo in range(0, total, per):
            _broadcast(arena[lo : min(lo + per, total)], src)
            launches += 1
        moved["bytes"] += total * es
        moved["collectives"] += launches
        if not repoint and rank != src:
            for t, off in zip(items, offs):
                t.copy_(arena[off : off + t.numel()].view(t.shape))
    return launches


def check_same_on_all_ranks(value: Any, what: str, src: int = 0) -> None:
    """Raise on EVERY rank when any rank's `value` (a small picklable description) differs from the source's."""
    import hashlib
    import pickle

    mine = hashlib.sha256(pickle.dumps(value)).hexdigest()
    box = [mine]
    dist.broadcast_object_list(box, src=src)
    ok = [None] * dist.get_world_size()
    dist.all_gather_object(ok, box[0] == mine)
    if not all(ok):
        bad = [r for r, o in enumerate(ok) if not o]
        raise RuntimeError(f"{what} differs from rank {src}'s on ranks {bad}")


def propagate_failure(exc: BaseException | None, what: str) -> None:
    """Collective: when any rank passes an exception, every rank raises (the failing one re-raises its own)."""
    msgs: list[Any] = [None] * dist.get_world_size()
    dist.all_gather_object(msgs, None if exc is None else f"{type(exc).__name__}: {exc}")
    if exc is not None:
        raise exc
    bad = [(r, m) for r, m in enumerate(msgs) if m is not None]
    if bad:
        raise RuntimeError(f"{what} failed on rank {bad[0][0]}: {bad[0][1]}")


def broadcast_module(module: torch.nn.Module, src: int = 0, bucket_bytes: int = 512 << 20) -> int:
    """Broadcast every parameter and buffer of a (possibly adapted) Chain tree from `src`.  Programs lowered from the tree
    before the call are invalidated (the weights moved into the broadcast arena)."""
    seen: dict[int, Tensor] = {}
    for t in list(module.parameters()) + list(module.buffers()):
        if t.device.type != "meta":
            seen.setdefault(id(t), t)
    n = broadcast_tensors(seen.values(), src=src, bucket_bytes=bucket_bytes)
    if n:
        from .fluxion.tree import bump_epoch

        bump_epoch()
    return n


def broadcast_packs(lower: Any, cache: Any, src: int = 0, bucket_bytes: int = 512 << 20) -> int:
    """Lower on every rank, pack on ONE: `lower()` builds this rank's launch program(s) against `cache` (a refiners_amd.engine.packing.PackCache);
    the source runs it normally and publishes the cache's manifest, the others run it with the cache in adopt mode (uninitialised storage
    of the published shapes instead of K-blocking / merging / LayerNorm folding / concatenating the weights themselves), then one bucketed
    broadcast fills the packed weights IN PLACE (the programs already hold their addresses).  Call it after broadcast_module (the leaves the
    aliasing entries point at must already hold the source's values).  Returns the number of collective launches (0 single-process)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        lower()
        return 0
    rank = dist.get_rank()
    err: BaseException | None = None
    box: list[Any] = [None]
    try:
        if rank == src:
            cache.mark()  # the hand-over covers what THIS lowering creates (a warm cache keeps its earlier entries to itself)
            lower()
            box[0] = cache.manifest()
    except BaseException as e:  # noqa: BLE001
        err = e
    propagate_failure(err, "broadcast_packs: lowering on the source rank")
    dist.broadcast_object_list(box, src=src)
    manifest = box[0]
    try:
        if rank != src:
            cache.adopt(manifest)
            lower()
    except BaseException as e:  # noqa: BLE001
        err = e
        cache.adopt(None)  # back to normal: a later lowering on this rank packs for itself
    propagate_failure(err, "broadcast_packs: lowering on a receiving rank")
    return broadcast_tensors(cache.leaves(manifest), src=src, bucket_bytes=bucket_bytes, repoint=False)


def load_and_broadcast(module: torch.nn.Module, tensors_path: Any, device: torch.device | str, src: int = 0, strict: bool = True) -> int:
    """Checkpoint -> every GPU without a host-side detour on the receivers: rank `src` reads the safetensors file STRAIGHT onto its
    device (safetensors maps the file and copies each tensor to HBM; the keys are refiners' Chain-path names, which the mirror
    shares) and adopts the tensors as its parameters (`assign=True`, no second copy); the other ranks only allocate uninitialised
    storage; then ONE arena broadcast (broadcast_module) moves the weights over xGMI.  Works for a tree built on the meta device.
    Returns the number of collective launches (0 in a single-process run)."""
    from safetensors.torch import load_file

    multi = dist.is_initialized() and dist.get_world_size() > 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    err: BaseException | None = None
    try:
        if rank == src:
            state = load_file(str(tensors_path), device=str(device))
            want = module.state_dict()
            cast = {k: (v if k not in want or v.dtype == want[k].dtype else v.to(want[k].dtype)) for k, v in state.items()}
            module.load_state_dict(cast, strict=strict, assign=True)
            # strict=False / a tree that was not built on meta: whatever the file did not cover must end up on `device` too, or the
            # source's tensor list (and with it the arena layout) would differ from the receivers'
            _materialise(module, torch.device(device))
        else:
            module.to_empty(device=device)
    except BaseException as e:  # noqa: BLE001 -- a source-only failure (missing file, strict mismatch) must not leave the others in the collective
        if not multi:
            raise
        err = e
    if multi:
        propagate_failure(err, "load_and_broadcast")
    return broadcast_module(module, src=src)


def _materialise(module: torch.nn.Module, device: torch.device) -> None:
    """Every parameter / buffer still on meta gets (uninitialised) storage on `device`; anything on another device moves there.
    Tensors already there keep their identity (id-keyed caches and outside references stay valid): "cuda" and "cuda:0" are the same place."""
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    for mod in module.modules():
        for store in (mod._parameters, mod._buffers):
            for name, t in list(store.items()):
                if t is None or t.device == device:
                    continue
                new = torch.empty_like(t, device=device) if t.device.type == "meta" else t.detach().to(device)
                store[name] = torch.nn.Parameter(new, requires_grad=t.requires_grad) if isinstance(t, torch.nn.Parameter) else new


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous slice of `n_items` prompts for `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def shard(items: Sequence[Any], rank: int, world: int) -> list[Any]:
    return [items[i] for i in shard_range(len(items), rank, world)]


def gather_latents(x: Tensor, dst: int = 0) -> Tensor | None:
    """Concatenate every rank's (n_i, 4, H, W) latents on `dst` (ranks may hold different n_i)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    world = dist.get_world_size()
    counts = all_gather(torch.tensor([x.shape[0]], dtype=torch.int64, device=x.device))
    most = int(max(int(c) for c in counts))
    padded = torch.zeros((most,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    padded[: x.shape[0]] = x
    parts = all_gather(padded)
    if dist.get_rank() != dst:
        return None
    return torch.cat([p[: int(c)] for p, c in zip(parts, counts)])


def max_over_ranks(value: float, device: torch.device | str = "cpu") -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)
