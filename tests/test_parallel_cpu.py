"""world_size-2 gloo tests of the multi-GPU layer (weights broadcast, prompt sharding, latent gather): the same code
path the 8-GPU bench takes with backend nccl (= RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, q) -> None:
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import refiners_amd.fluxion.layers as fl
    from refiners_amd import parallel
    from refiners_amd.fluxion.adapters import LinearLora, LoraAdapter
    from refiners_amd.latent_diffusion.sampling import DDIM, SDXLDenoiser  # noqa: F401

    parallel.init_from_env("gloo")
    torch.manual_seed(100 + rank)  # ranks start with DIFFERENT weights; after the broadcast they must equal rank 0's
    model = fl.Chain(fl.Linear(16, 32), fl.SiLU(), fl.Linear(32, 8), fl.LayerNorm(8))
    lora = LinearLora("l", in_features=16, out_features=32, rank=4)
    torch.nn.init.normal_(lora.up.weight)
    LoraAdapter(model[0], lora).inject(model)
    n = parallel.broadcast_module(model, src=0, bucket_bytes=1024)  # tiny buckets: several launches
    digest = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).double().sum().item()
    prompts = [f"p{i}" for i in range(5)]
    mine = parallel.shard(prompts, rank, world)
    x = torch.full((len(mine), 4, 2, 2), float(rank))
    allx = parallel.gather_latents(x, dst=0)
    slow = parallel.max_over_ranks(1.0 + rank)
    q.put((rank, n, digest, mine, None if allx is None else allx[:, 0, 0, 0].tolist(), slow))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_shard_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, n0, d0, m0, all0, s0), (r1, n1, d1, m1, all1, s1) = got
    assert n0 == n1 and n0 > 1
    assert d0 == d1  # identical weights everywhere after the broadcast
    assert m0 == ["p0", "p1", "p2"] and m1 == ["p3", "p4"]
    assert all0 == [0.0, 0.0, 0.0, 1.0, 1.0] and all1 is None
    assert s0 == s1 == 2.0


def _worker_load(rank: int, world: int, port: int, path: str, q) -> None:
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import refiners_amd.fluxion.layers as fl
    from refiners_amd import parallel

    parallel.init_from_env("gloo")
    model = fl.Chain(fl.Linear(16, 32, device="meta"), fl.SiLU(), fl.Linear(32, 8, device="meta"))  # nobody materialises weights up front
    n = parallel.load_and_broadcast(model, path if rank == 0 else "/nonexistent: only rank 0 reads the file", device="cpu")
    q.put((rank, n, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).double().sum().item(), all(p.device.type == "cpu" for p in model.parameters())))
    dist.barrier()
    dist.destroy_process_group()


def test_checkpoint_is_read_once_and_broadcast(tmp_path):
    """next-3: rank 0 loads the safetensors file directly to its device, the others receive through the arena broadcast."""
    from safetensors.torch import save_file

    import refiners_amd.fluxion.layers as fl

    torch.manual_seed(5)
    ref = fl.Chain(fl.Linear(16, 32), fl.SiLU(), fl.Linear(32, 8))
    path = tmp_path / "m.safetensors"
    save_file({k: v.contiguous() for k, v in ref.state_dict().items()}, str(path))
    want = torch.cat([p.detach().reshape(-1) for p in ref.parameters()]).double().sum().item()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_load, args=(r, 2, port, str(path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(g[1] >= 1 and g[2] == want and g[3] for g in got), got


def _spawn(target, args_for_rank, world=2, timeout=180):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, *args_for_rank(r), q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=timeout) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def _worker_fail(rank: int, world: int, port: int, mode: str, path: str, q) -> None:
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import refiners_amd.fluxion.layers as fl
    from refiners_amd import parallel

    parallel.init_from_env("gloo")
    outcome = "ok"
    try:
        if mode == "missing_file":  # only the source touches the file: its failure must reach the others instead of leaving them in the collective
            model = fl.Chain(fl.Linear(16, 32, device="meta"), fl.Linear(32, 8, device="meta"))
            parallel.load_and_broadcast(model, "/nonexistent/weights.safetensors", device="cpu")
        elif mode == "different_trees":  # rank 1 built another tree: arena layouts would differ -> refuse before anything moves
            model = fl.Chain(fl.Linear(16, 32), fl.Linear(32, 8 if rank == 0 else 4))
            parallel.broadcast_module(model)
        elif mode == "partial":  # strict=False, the file lacks one tensor: it must still end up on the device on EVERY rank, same list everywhere
            model = fl.Chain(fl.Linear(16, 32, device="meta"), fl.Linear(32, 8, device="meta"))
            n = parallel.load_and_broadcast(model, path, device="cpu", strict=False)
            outcome = ("ok", n, sorted((k, tuple(v.shape), v.device.type) for k, v in model.state_dict().items()),
                       float(model.state_dict()["Linear_1.weight"].double().sum()))
    except Exception as e:  # noqa: BLE001
        outcome = f"{type(e).__name__}: {e}"
    q.put((rank, outcome))
    dist.barrier()
    dist.destroy_process_group()


def test_a_source_only_failure_reaches_every_rank():
    got = _spawn(_worker_fail, lambda r: ("missing_file", ""))
    assert all(isinstance(o, str) and o != "ok" for _, o in got), got
    assert "rank 0" in got[1][1]  # the receiver's message names the failing rank


def test_different_tensor_lists_are_refused_before_the_broadcast():
    got = _spawn(_worker_fail, lambda r: ("different_trees", ""))
    assert all(isinstance(o, str) and "differs from rank 0" in o for _, o in got), got


def test_partial_checkpoint_keeps_the_ranks_consistent(tmp_path):
    from safetensors.torch import save_file

    import refiners_amd.fluxion.layers as fl

    torch.manual_seed(6)
    ref = fl.Chain(fl.Linear(16, 32), fl.Linear(32, 8))
    sd = {k: v.contiguous() for k, v in ref.state_dict().items() if k != "Linear_2.bias"}
    path = tmp_path / "partial.safetensors"
    save_file(sd, str(path))
    got = _spawn(_worker_fail, lambda r: ("partial", str(path)))
    (_, a), (_, b) = got
    assert a[0] == b[0] == "ok" and a[1] == b[1] >= 1 and a[2] == b[2] and a[3] == b[3] == float(ref.state_dict()["Linear_1.weight"].double().sum())
    assert all(dev == "cpu" for _, _, dev in a[2])


class _ToyPipe:
    """Stands in for CompiledSDXL in the CPU dry run of bench.py's multi-process flow: x <- 0.9 x + unet(x) per step."""

    def __init__(self, unet, x):
        self.unet, self.x = unet, x

    def step(self, i):
        with torch.no_grad():
            self.x = 0.9 * self.x + 0.01 * self.unet(self.x)
        return self.x


def _worker_bench(rank: int, world: int, port: int, q) -> None:
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    import refiners_amd.fluxion.layers as fl
    from refiners_amd import parallel

    r, w, _ = parallel.init_from_env("gloo")
    torch.manual_seed(1000 + rank)  # like bench.build_pipeline: rank 0 draws the weights, the others receive them
    unet = fl.Chain(fl.Conv2d(4, 8, 3, padding=1), fl.SiLU(), fl.Conv2d(8, 4, 3, padding=1))
    n_b = parallel.broadcast_module(unet, src=0)
    prompts = list(range(6))
    mine = parallel.shard(prompts, r, w)  # independent prompts per rank, no per-step collective
    g = torch.Generator().manual_seed(7)
    x_all = torch.randn(len(prompts), 4, 8, 8, generator=g)
    pipe = _ToyPipe(unet, x_all[mine[0] : mine[-1] + 1].clone())
    elapsed = bench.timed_steps(pipe, steps=4, warmup=1, world=w, dev="cpu")  # barrier + max over ranks, as on the GPUs
    allx = parallel.gather_latents(pipe.x, dst=0)
    want = _ToyPipe(unet, x_all.clone())
    for i in range(5):
        want.step(i)
    q.put((rank, n_b, elapsed, None if allx is None else float((allx - want.x).abs().max())))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_shaped_dry_run_world2():
    """bench.py --gpus 2 on CPU tensors: weights broadcast once, prompts sharded, the timed loop between barriers with the max over
    ranks, latents gathered on rank 0 -- and the gathered result equals the single-process run (the path shards without any exchange)."""
    (_, n0, e0, d0), (_, n1, e1, d1) = _spawn(_worker_bench, lambda r: ())
    assert n0 == n1 >= 1 and e0 == e1 > 0  # both ranks report the same (max-over-ranks) time
    assert d0 is not None and d0 < 1e-6 and d1 is None


def _worker_packs(rank: int, world: int, port: int, q) -> None:
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import refiners_amd.fluxion.layers as fl
    from refiners_amd import native, parallel
    from refiners_amd.engine.lowering_blocks import BlockLowering
    from refiners_amd.engine.packing import Act, PackCache, _leaves
    from refiners_amd.engine.unet_lowering import UNetContext
    from refiners_amd.fluxion.adapters import LinearLora, LoraAdapter
    from refiners_amd.latent_diffusion.blocks import CrossAttentionBlock2d, ResidualBlock

    native.load()
    parallel.init_from_env("gloo")
    torch.manual_seed(100 + rank)  # different weights per rank until the broadcast
    tree = fl.Chain(ResidualBlock(64, 128), CrossAttentionBlock2d(channels=128, context_embedding_dim=64, context_key="clip_text_embedding", num_attention_heads=2, use_bias=False, use_linear_projection=True))
    for lin in [m for m in tree.layers(fl.Linear) if m.in_features == 128 and m.out_features == 128][:3]:  # live LoRAs: stacked / K-blocked packs
        lo = LinearLora("l", in_features=128, out_features=128, rank=8)
        torch.nn.init.normal_(lo.up.weight)
        LoraAdapter(lin, lo).inject()
    parallel.broadcast_module(tree, src=0)
    cache = PackCache()
    B, H, W = 2, 8, 8

    def lower() -> None:
        low = BlockLowering(torch.device("cpu"), torch.float32, cache)
        ctx = UNetContext(low, B)
        ctx.text[("cross_attention_block", "clip_text_embedding")] = (torch.zeros(B * 64, 64), 7)
        with low.in_step():
            a = Act(low.pool.get(B * H * W, 64), B, H, W)
            a = low.residual_block(tree[0], a, ctx)
            low.cross_attention_2d(tree[1], a, ctx)
        cache.sweep()

    n = parallel.broadcast_packs(lower, cache, src=0, bucket_bytes=4096)
    man = cache.manifest()
    digest = sum(float(t.double().sum()) for key in cache.order for t in _leaves(cache.store[key]))
    q.put((rank, n, cache.made, sum(1 for m in man if m[0] == "recv"), len(man), digest))
    dist.barrier()
    dist.destroy_process_group()


def test_only_the_source_rank_packs_the_weights():
    """PackCache contents (K-blocked copies, stacked LoRA rows, LayerNorm-folded weights, merged biases ...) are computed on
    rank 0 only and broadcast; the receivers' lowering makes nothing but the aliasing entries (views of leaves the weight broadcast already
    filled), and ends up with identical packed weights."""
    (_, n0, made0, recv0, len0, d0), (_, n1, made1, recv1, len1, d1) = _spawn(_worker_packs, lambda r: ())
    assert n0 == n1 >= 1 and len0 == len1 and recv0 == recv1 > 5
    assert made0 == len0  # the source made everything ...
    assert made1 == len1 - recv1  # ... the receiver only the aliases
    assert d0 == d1


def test_shard_range_covers_everything():
    from refiners_amd.parallel import shard_range

    for n in (0, 1, 7, 32):
        for world in (1, 2, 3, 8):
            idx = [i for r in range(world) for i in shard_range(n, r, world)]
            assert idx == list(range(n))


def _worker_nccl(rank: int, world: int, port: int, q) -> None:
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import refiners_amd.fluxion.layers as fl
    from refiners_amd import parallel

    parallel.init_from_env("nccl")
    dev = torch.device("cuda", rank)
    torch.manual_seed(100 + rank)
    model = fl.Chain(fl.Linear(256, 512, device=dev, dtype=torch.bfloat16), fl.SiLU(), fl.Linear(512, 128, device=dev, dtype=torch.bfloat16))
    n = parallel.broadcast_module(model, src=0, bucket_bytes=64 << 10)
    digest = torch.cat([p.detach().float().reshape(-1) for p in model.parameters()]).double().sum().item()
    aligned = all(p.data_ptr() % 256 == 0 for p in model.parameters())  # the arena keeps the kernels' 16-byte (here 256-byte) alignment
    allx = parallel.gather_latents(torch.full((1 + rank, 4, 2, 2), float(rank), device=dev), dst=0)
    q.put((rank, n, digest, aligned, None if allx is None else allx[:, 0, 0, 0].tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_broadcast_and_gather_over_rccl_world2():
    """The same path with backend nccl (= RCCL over xGMI) on two GPUs of one node; skipped on single-GPU boxes."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_nccl, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (r0, n0, d0, a0, all0), (r1, n1, d1, a1, all1) = got
    assert n0 == n1 and n0 >= 1 and d0 == d1 and a0 and a1
    assert all0 == [0.0, 1.0, 1.0] and all1 is None
