"""Run-time side of the MI355X engine: static I/O staging, (re)lowering on tree changes, HIP-graph replay.

`CompiledUNet` is what a refiners user switches to:

    unet = SDXLUNet(...)                       # refiners_amd.latent_diffusion (or refiners' own class)
    fast = CompiledUNet(unet)                  # nothing happens yet
    unet.set_timestep(t); unet.set_clip_text_embedding(e); ...   # same context API as always
    y = fast(x)                                # == unet(x), on hand-written gfx950 kernels

It keeps refiners' contract at the UNet boundary (reference latent_diffusion/model.py:128-159 calls `self.unet(latents)`
after `set_unet_context`): inputs come from the UNet's context store, the context is reset after the call exactly like
`Chain.forward` does (chain.py:245-257), adapters may be injected / ejected / rescaled between calls (the lowering is
redone when `tree_epoch()` moved).  `CompiledSDXL` adds the classifier-free-guidance + DDIM update as one more kernel and
replays a whole denoising step as ONE HIP graph launch.

There is no silent fallback: if the native library is missing this module raises.  A tree shape the lowering does not know follows the
reference's own convention for unknown children -- the stock child loop runs (fluxion/layers/chain.py:226-257; SURVEY.md section 8(b):
"unsupported => fall back, never error") -- at two levels, both reported: a context-free sub-tree inside a UNet stage runs through its own
torch forward inside the recorded program (`.stats["fallback_nodes"]`); anything else (a node that needs the Chain's context store at run
time, an unknown top-level layout) makes the WHOLE call run `unet(x)`, with a `RuntimeWarning` naming the reason and
`.stats["whole_fallback"]` set.  Adapters outside SURVEY.md section 8 (FreeU, reference-only, StyleAligned ...) therefore keep working, unfused.
"""
from __future__ import annotations

import os

import warnings
from typing import Any, Optional

import torch
from torch import Tensor

from .. import native
from ..fluxion.tree import tree_epoch
from .packing import PackCache, Unsupported, isa, kids, launches  # noqa: F401
from .unet_lowering import UNetIO, UNetLowering  # noqa: F401

TOKEN_CONTEXTS = (("cross_attention_block", "clip_text_embedding"), ("ip_adapter", "clip_image_embedding"))


def _ident(t: Optional[Tensor]) -> Any:
    """Identity of a prompt-side input.  The key alone is not enough: once the caller drops the tensor the allocator may hand
    the same block (same data_ptr, same _version) to the NEXT prompt's embedding, so whoever stores this key also stores the
    tensor itself (CompiledUNet.prologue_refs) -- a live tensor's address cannot be recycled."""
    return None if t is None else (id(t), t.data_ptr(), t._version, tuple(t.shape), t.dtype)


def _weights_version(unet: Any, cache: Optional[dict] = None, epoch: Any = None) -> int:
    """Changes whenever a parameter is updated IN PLACE (load_state_dict / load_from_safetensors without assign,
    parallel.broadcast_module, optimizer steps): converted, merged and K-blocked copies must then be rebuilt.  `cache` (owned by
    the CompiledUNet, so it dies with it) keeps the parameter list of a mirror tree per tree epoch: the walk over ~2 900 modules
    costs more than the sum."""
    if cache is None or epoch is None:
        return sum(p._version for p in unet.parameters())
    if cache.get("epoch") != epoch:
        cache["epoch"], cache["params"] = epoch, list(unet.parameters())
    return sum(p._version for p in cache["params"])


class Program:
    """A lowered launch list replayed directly the first time and as ONE HIP graph afterwards (every buffer is static)."""

    def __init__(self, ops: list, use_graph: bool, weight_prefetch: Optional[bool] = None, low: Any = None) -> None:
        # `low`: the Lowering the launches came from.  Given, every run ends with a look at its hand-over error words (in-launch LoRA, stream-K):
        # these programs -- text / image encoders, SAM -- run once per prompt or picture and their callers read the result next, so the host
        # round trip costs nothing that is not paid anyway
        self.ops, self.use_graph, self.low = ops, use_graph, low
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        if weight_prefetch is None:
            weight_prefetch = os.environ.get("REFINERS_AMD_WEIGHT_PREFETCH", "1") != "0"
        self.prefetch = native.link_weight_prefetch(ops, enable=weight_prefetch)  # every GEMM pulls the next one's weights into the Infinity Cache

    def run(self) -> None:
        if self.use_graph and self.graph is not None:
            self.graph.replay()
        else:
            native.replay(self.ops)  # also the warm-up: first-launch work (function attributes) must not be captured
            if self.use_graph:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    native.replay(self.ops)
                self.graph = g
        if self.low is not None:
            bad = self.low.handover_pending()
            if bad and bool((bad[0] if len(bad) == 1 else torch.stack(bad).any()).item()):
                self.low.handover_raise()


class CompiledUNet:
    def __init__(self, unet: Any, use_graph: bool = True, lora_mode: str = "fused") -> None:
        native.load()  # fail loudly: there is no fallback for a missing HIP library
        self.unet = unet
        self.use_graph = use_graph
        self.lora_mode = lora_mode  # see Lowering.__init__
        self.weight_prefetch = os.environ.get("REFINERS_AMD_WEIGHT_PREFETCH", "1") != "0"
        self.cache = PackCache()
        self.io_override: Optional[tuple[Tensor, Tensor]] = None  # (x, out) buffers owned by the caller instead of fresh ones (see _build)
        self.low: Optional[UNetLowering] = None
        self.io: Optional[UNetIO] = None
        self.key: Any = None
        self.bad_key: Any = None  # the (tree state, geometry) key whose lowering raised Unsupported: those calls run the stock forward
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.prologue_key: Any = None
        self.prologue_refs: Any = None  # the staged source tensors themselves (see _ident)
        self.stats: dict[str, Any] = {}

    # -- context plumbing ----------------------------------------------------------------------------------
    def _contexts(self) -> dict[str, dict[str, Any]]:
        # Chain.set_context REPLACES the top-level dict but MERGES into every descendant's (context.py:16-46,
        # chain.py:131-156), so after set_timestep / set_time_ids / ... only a child Chain sees
        # all of them -- read where the UseContext nodes read, not at unet.provider.
        child = next((m for m in kids(self.unet) if isa(m, "Chain")), None)
        return (child if child is not None else self.unet).provider.contexts

    def _gather(self) -> dict[str, Any]:
        c = self._contexts()
        diff = c.get("diffusion", {})
        got: dict[str, Any] = {"timestep": diff.get("timestep"), "pooled": diff.get("pooled_text_embedding"), "time_ids": diff.get("time_ids"), "tokens": {}, "conditions": {},
                               "t2i": {}}
        assert got["timestep"] is not None, "context diffusion.timestep is unset (call unet.set_timestep first)"
        for ctx, key in TOKEN_CONTEXTS:
            v = c.get(ctx, {}).get(key)
            if v is not None:
                got["tokens"][(ctx, key)] = v
        for name, d in c.items():
            if name.startswith("control_lora_") and d.get("condition") is not None:
                got["conditions"][name] = d["condition"]
            if name == "controlnet":  # SD1.5 Controlnet adapters share one context, one key per adapter (sd1/controlnet.py:200-204)
                for key, v in d.items():
                    if key.startswith("condition_") and v is not None:
                        got["conditions"][f"controlnet.{key}"] = v
        for key, feats in c.get("t2iadapter", {}).items():  # T2IAdapter.set_condition_features (t2i_adapter.py:201-202)
            if key.startswith("condition_features_") and feats is not None:
                got["t2i"][key.removeprefix("condition_features_")] = tuple(feats)
        return got

    # -- lowering ----------------------------------------------------------------------------------------------
    def _build(self, x_shape: tuple, dev: torch.device, got: dict[str, Any]) -> None:
        dtype = self.unet.dtype
        assert dtype in (torch.float32, torch.bfloat16), f"the MI355X path computes in float32 or bfloat16, not {dtype}"
        B, _, H, W = x_shape
        xbuf, obuf = self.io_override if self.io_override is not None else (None, None)  # CompiledSDXL's split CFG pair: halves of ONE 2n-row buffer
        if xbuf is None:
            xbuf = torch.empty(tuple(x_shape), device=dev, dtype=dtype)
            obuf = torch.empty(B, self._out_channels(), H, W, device=dev, dtype=dtype)
        assert tuple(xbuf.shape) == tuple(x_shape) and tuple(obuf.shape) == (B, self._out_channels(), H, W) and xbuf.is_contiguous() and obuf.is_contiguous() and xbuf.dtype == obuf.dtype == dtype
        io = UNetIO(x=xbuf, timestep=torch.empty(B, device=dev, dtype=torch.float32), out=obuf)
        if got["pooled"] is not None:
            io.pooled = torch.empty(B, got["pooled"].shape[1], device=dev, dtype=dtype)
            io.time_ids = torch.empty(B, got["time_ids"].shape[1], device=dev, dtype=torch.float32)
        if got.get("timesteps_all") is not None:  # CompiledSDXL: the solver's whole timestep list -> the timestep-embedding chain becomes a prologue table
            S = int(got["timesteps_all"].numel())
            io.timesteps = torch.empty(B * S, device=dev, dtype=torch.float32)
            io.step_rows = torch.zeros(B, device=dev, dtype=torch.int32)
            self.rows_table = (torch.arange(B, dtype=torch.int32).unsqueeze(0) * S + torch.arange(S, dtype=torch.int32).unsqueeze(1)).to(dev)  # [S, B]: b*S + s
        for ck, v in got["tokens"].items():
            assert v.shape[0] == B, f"context {ck} has batch {v.shape[0]}, latents have {B}"
            L, width = v.shape[1], v.shape[2]
            lp = (L + 63) // 64 * 64
            io.tokens[ck] = (torch.zeros(B * lp, width, device=dev, dtype=dtype), L)
        for name, v in got["conditions"].items():
            io.conditions[name] = torch.empty(tuple(v.shape), device=dev, dtype=dtype)
        for name, feats in got.get("t2i", {}).items():
            io.t2i[name] = [torch.empty(tuple(f.shape), device=dev, dtype=dtype) for f in feats]
        low = UNetLowering(dev, dtype, self.cache, self.lora_mode)
        low.sag_capture = getattr(self, "sag_capture", True)
        low.lower(self.unet, io)
        self.cache.sweep()
        # the step program is replayed step after step: let every GEMM / conv pull the weights of the launches behind it into
        # the Infinity Cache (weights are read exactly once per step, so otherwise every kernel starts on DRAM misses)
        pf = native.link_weight_prefetch(low.step, enable=self.weight_prefetch)
        self.low, self.io, self.graph, self.prologue_key, self.prologue_refs = low, io, None, None, None
        from . import tuning

        self.stats = dict(low.stats, step_ops=launches(low.step), prologue_ops=launches(low.prologue), pool_bytes=low.step_pool.bytes() + low.prologue_pool.bytes(),
                          weight_prefetch=pf, gemm_tuning=tuning.summary(), ln_fuse=low.ln_fuse, qkv_merge=low.qkv_merge)

    def _out_channels(self) -> int:
        last = [m for m in self.unet.modules() if isa(m, "Conv2d")][-1]
        return last.out_channels

    def _stage_inputs(self, got: dict[str, Any]) -> bool:
        """Copy the user's side inputs into the static buffers; returns True when a prompt-side input changed."""
        io = self.io
        assert io is not None
        ts = got["timestep"].to(device=io.timestep.device, dtype=torch.float32).reshape(-1)
        io.timestep.copy_(ts.expand(io.timestep.shape[0]) if ts.numel() == 1 else ts)
        if io.step_rows is not None:
            io.step_rows.copy_(self.rows_table[int(got["step_index"])])
        pk = (_ident(got.get("timesteps_all")), _ident(got["pooled"]), _ident(got["time_ids"]), tuple(_ident(v) for v in got["tokens"].values()), tuple(_ident(v) for v in got["conditions"].values()),
              tuple(_ident(f) for feats in got.get("t2i", {}).values() for f in feats))
        if pk == self.prologue_key:
            return False
        if io.pooled is not None:
            io.pooled.copy_(got["pooled"])
            io.time_ids.copy_(got["time_ids"])  # type: ignore[union-attr]
        if io.timesteps is not None:
            B = io.timestep.shape[0]
            io.timesteps.view(B, -1).copy_(got["timesteps_all"].to(device=io.timesteps.device, dtype=torch.float32).reshape(1, -1).expand(B, -1))
        for ck, v in got["tokens"].items():
            buf, L = io.tokens[ck]
            B = v.shape[0]
            buf.view(B, -1, v.shape[2])[:, :L].copy_(v)
        for name, v in got["conditions"].items():
            io.conditions[name].copy_(v)
        for name, feats in got.get("t2i", {}).items():
            for buf, f in zip(io.t2i[name], feats):
                buf.copy_(f)
        self.prologue_key = pk
        self.prologue_refs = (got.get("timesteps_all"), got["pooled"], got["time_ids"], tuple(got["tokens"].values()), tuple(got["conditions"].values()),
                              tuple(f for feats in got.get("t2i", {}).values() for f in feats))
        return True

    # -- execution ---------------------------------------------------------------------------------------------
    def prepare(self, x: Tensor) -> bool:
        """Make sure a program matching the tree and the input geometry exists and its inputs are staged (inputs taken
        from the UNet's context store, as refiners' pipeline leaves them)."""
        got = self._gather()
        changed = self.prepare_explicit(tuple(x.shape), x.device, got)
        self.io.x.copy_(x)  # type: ignore[union-attr]
        return changed

    def _tree_state(self) -> Any:
        """What invalidates a lowered program.  Trees built from refiners_amd.fluxion bump a global epoch on every
        structural change and scale assignment; trees built from refiners' own classes have no such counter, so their
        state is a signature of the module identities and the live scales (a ~1 ms walk per call)."""
        from ..fluxion.tree import Chain as MirrorChain

        if isinstance(self.unet, MirrorChain):
            ep = tree_epoch()
            return (ep, _weights_version(self.unet, self.__dict__.setdefault("_params_cache", {}), ep))
        sig: list[Any] = [_weights_version(self.unet)]
        for m in self.unet.modules():
            sig.append(id(m))
            if isa(m, "Multiply", "T2IFeatures"):  # nodes whose live scale is baked into the program
                sig.append(float(m.scale))
            if isa(m, "Controlnet"):
                sig.extend([float(m.scale), float(m.scale_decay)])
        return hash(tuple(sig))

    def prepare_explicit(self, x_shape: tuple, device: torch.device, got: dict[str, Any]) -> bool:
        """Same, with the side inputs given explicitly: {"timestep", "pooled", "time_ids", "tokens": {(ctx, key): t},
        "conditions": {ctx_name: t}}.  Does not stage x (the caller fills io.x).  Returns True when the prologue must run."""
        key = (self._tree_state(), tuple(x_shape), self.unet.dtype, None if got.get("timesteps_all") is None else int(got["timesteps_all"].numel()),
               tuple((k, tuple(v.shape)) for k, v in got["tokens"].items()),
               tuple((k, tuple(v.shape)) for k, v in got["conditions"].items()), got["pooled"] is not None,
               tuple((k, tuple(tuple(f.shape) for f in feats)) for k, feats in got.get("t2i", {}).items()))
        if key != self.key:
            if key == self.bad_key:
                raise Unsupported(self.stats.get("whole_fallback", "this tree could not be lowered"))
            try:
                self._build(x_shape, device, got)
            except Unsupported as exc:
                # remembered per (tree state, geometry): the next call does not walk the tree again; any inject / eject / scale change retries
                self.bad_key, self.key, self.low, self.io, self.graph = key, None, None, None, None
                self.stats = {"whole_fallback": str(exc), "fallback_nodes": ["<whole UNet>"], "step_ops": 0, "prologue_ops": 0}
                warnings.warn(f"refiners_amd: this UNet tree is not lowered to the MI355X kernels ({exc}); running the stock Chain forward instead", RuntimeWarning, stacklevel=3)
                raise
            self.key, self.bad_key = key, None
        return self._stage_inputs(got)

    def run_prologue(self) -> None:
        assert self.low is not None
        native.replay(self.low.prologue)
        self.check_handovers()  # once per prompt: a host round trip here is nothing next to the prologue's ~290 launches

    CHECK_EVERY = 16  # replays of the step program between two looks at the hand-over error words (one stacked .any() + one host sync)

    def check_handovers(self, every: int = 1) -> None:
        """The in-launch LoRA hand-over and the stream-K collect are bounded waits: a tile that gives up after 2 s raises an error word in device
        memory and goes on with undefined operands (a trap would kill the process's HIP context).  Nothing on the device can raise into Python, so
        the engine looks at those words at its host sync points -- after the prologue, every CHECK_EVERY-th step replay (`every`), at the end of a
        sampling run -- and turns a raised word into NativeError (round-5 advisor: until round 6 only the tests looked)."""
        self._replays = getattr(self, "_replays", 0) + 1
        if self.low is None or self._replays % max(every, 1):
            return
        bad = self.low.handover_pending()
        if bad and bool((bad[0] if len(bad) == 1 else torch.stack(bad).any()).item()):
            self.low.handover_raise()

    def run_step(self) -> None:
        """Replay the per-step program (directly the first time, as a HIP graph afterwards)."""
        assert self.low is not None
        if not self.use_graph:
            native.replay(self.low.step)
            return
        if self.graph is None:
            native.replay(self.low.step)  # warm-up: first-launch work (function attributes, workspace) must not be captured
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                native.replay(self.low.step)
            self.graph = g
            return
        self.graph.replay()

    @torch.no_grad()
    def __call__(self, x: Tensor) -> Tensor:
        try:
            changed = self.prepare(x)
        except Unsupported:
            return self.unet(x)  # the stock child loop: reads the same context store, resets it itself (chain.py:245-257)
        if changed:
            self.run_prologue()
        self.check_handovers(self.CHECK_EVERY)  # (looks at what the PREVIOUS replays left: no wait for this one)
        self.run_step()
        assert self.io is not None
        self.unet._reset_context()  # what Chain.forward does after running its children (chain.py:256)
        return self.io.out.clone()


class CompiledSDXL:
    """One classifier-free-guidance denoising step per call: UNet on cat(x, x) + CFG combine + solver update
    (reference latent_diffusion/model.py:128-159, solvers/ddim.py:56-95), latents resident in HBM across steps.

    `solver`: None = DDIM over `num_inference_steps` (SDXL's default); or an `Euler` / `DPMSolver` of
    refiners_amd.latent_diffusion.solvers (anything with `linear_step`): guidance + update + the next step's model-input
    scaling then run as ONE kernel (mi355x_cfg_linear_step) and the two cat(x, x) copies disappear from the step.
    SD1.5 UNets work too: leave `pooled_text_embedding` / `time_ids` out of `set_inputs`.

    The per-step host work is: two tiny device copies (timestep, solver coefficients) and one hipGraphLaunch.  The Chain
    tree's context store is not touched per step (the reference spends ~19 000 Python calls per step on it)."""

    def __init__(self, unet: Any, num_inference_steps: int = 50, condition_scale: float = 5.0, use_graph: bool = True, lora_mode: str = "fused",
                 solver: Any = None, cfg_split: Optional[bool] = None) -> None:
        from ..latent_diffusion.sampling import DDIM

        self.unet = unet
        self.engine = CompiledUNet(unet, use_graph=False, lora_mode=lora_mode)
        # cfg_split: the two halves of the classifier-free-guidance pair never meet before the guidance kernel (model.py:128-159 runs them as one
        # batch only because that is one module call), so they are lowered as TWO batch-n programs and replayed on two streams -- two branches of
        # the one captured HIP graph.  At one image per GPU every launch of the pair is short of workgroups (M = 2048 rows: 160 tiles on 256 CUs)
        # and a third of the step is fill / epilogue / launch boundary; two independent launch sequences fill each other's gaps.
        # None = REFINERS_AMD_CFG_SPLIT (default: see _split_wanted).  Self-Attention Guidance keeps the single program (its tap reads both halves).
        self.cfg_split = cfg_split if cfg_split is not None else {"0": False, "1": True}.get(os.environ.get("REFINERS_AMD_CFG_SPLIT", ""), None)
        self.engine_c: Optional[CompiledUNet] = None  # split mode: the conditional half (self.engine then runs the negative half)
        self.pair_x: Optional[Tensor] = None  # split mode: [2n, C, H, W] model input / UNet output, halves owned by the two engines
        self.pair_out: Optional[Tensor] = None
        self.side_stream: Optional[torch.cuda.Stream] = None
        self.use_graph = use_graph
        self.coef_table: Optional[Tensor] = None
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.solver = solver if solver is not None else DDIM(num_inference_steps)
        self.hist: Optional[Tensor] = None
        self.primed, self.primed_key = False, None
        self._condition_scale = condition_scale
        self.x: Optional[Tensor] = None
        self.graph_key: Any = None
        self.inputs: dict[str, Any] = {}
        self.engine2: Optional[CompiledUNet] = None  # self-attention guidance: the second (unconditional, batch n) UNet pass

    @property
    def condition_scale(self) -> float:
        return self._condition_scale

    @condition_scale.setter
    def condition_scale(self, value: float) -> None:  # baked into the device coefficient table: rebuild it on the next set_inputs / step
        self._condition_scale = value
        self.coef_table = None
        self.graph = None  # self-attention guidance bakes sag.scale / condition_scale into the captured launches as a host scalar

    @property
    def solver(self) -> Any:
        return self._solver

    @solver.setter
    def solver(self, value: Any) -> None:
        self._solver = value
        self.linear = hasattr(value, "linear_step")
        self.coef_table, self.graph = None, None  # other coefficients, possibly another update kernel

    def _tables(self, device: torch.device) -> None:
        rows = []
        for s in range(self.solver.num_inference_steps):
            if self.linear:
                rows.append([self.condition_scale, *self.solver.linear_step(s)])
            else:
                cur, sig, prev, nf = self.solver.coefficients(s)
                rows.append([self.condition_scale, cur, sig, prev, nf, 0.0, 0.0, 0.0])
        self.coef_table = torch.tensor(rows, dtype=torch.float32, device=device)
        self.coef = torch.zeros(8, dtype=torch.float32, device=device)
        self.ts_table = self.solver.timesteps.to(device=device, dtype=torch.float32)
        # Self-Attention Guidance degrades the latents through Solver.remove_noise / add_noise: (-, scale_t, std_t) per step, for mi355x_sag_degrade.
        # A solver whose add_noise / remove_noise the reference itself cannot evaluate (Euler: float timesteps) has no table; SAG then raises.
        # The one refusal that is the reference's own: a solver whose timesteps are floats (Euler) cannot index the integer train-time tables
        # in add_noise / remove_noise -- torch raises IndexError there too.  Anything else (a solver without the method, a bug in its tables)
        # propagates instead of being turned into "no SAG" (round-3 advisor finding).
        ts = self.solver.timesteps
        if not hasattr(self.solver, "sag_coefficients") or (torch.is_tensor(ts) and ts.is_floating_point()):
            self.sag_table = None
        else:
            self.sag_table = torch.tensor([[0.0, *self.solver.sag_coefficients(s)] for s in range(self.solver.num_inference_steps)], dtype=torch.float32, device=device)
        self.sag_coef = torch.zeros(3, dtype=torch.float32, device=device)

    def set_inputs(self, x: Tensor, *, clip_text_embedding: Tensor, pooled_text_embedding: Optional[Tensor] = None, time_ids: Optional[Tensor] = None,
                   clip_image_embedding: Optional[Tensor] = None, conditions: Optional[dict[str, Tensor]] = None,
                   t2i_features: Optional[dict[str, Any]] = None, generator: Optional[torch.Generator] = None) -> None:
        """x: (N, 4, H, W) initial latents; embeddings are [negative ; conditional] stacks of 2N rows;
        `conditions` maps a ControlLora name to its control image, (2N, 3, 8H, 8W) or ONE picture (1, 3, 8H, 8W) for the whole batch; `t2i_features` maps a T2I-Adapter name to
        the tuple its `compute_condition_features` returned (batch 1 or 2N)."""
        self.generator = generator  # stochastic solvers (LCM) draw their per-step noise from it, in the reference's order
        tokens = {("cross_attention_block", "clip_text_embedding"): clip_text_embedding}
        if clip_image_embedding is not None:
            tokens[("ip_adapter", "clip_image_embedding")] = clip_image_embedding
        self.inputs = {"pooled": pooled_text_embedding, "time_ids": time_ids, "tokens": tokens,
                       "conditions": {f"control_lora_{k}": v for k, v in (conditions or {}).items()},
                       "t2i": {k: tuple(v) for k, v in (t2i_features or {}).items()}}
        if self.x is None or self.x.shape != x.shape or self.x.device != x.device or self.x.dtype != self.unet.dtype:
            self.x = torch.empty(tuple(x.shape), device=x.device, dtype=self.unet.dtype)
            # the solver history lives as long as x does: a captured graph holds both addresses, so neither may be
            # re-allocated per trajectory (a fresh zeros_like() here left the graph reading / writing a freed block)
            self.hist = torch.zeros_like(self.x)
            self.graph = None
        self.x.copy_(x)
        if self.linear:
            assert self.hist is not None
            self.hist.zero_()
            self.primed = False  # the model-input buffer must be (re)filled with s_0 * x before the next step
        if self.coef_table is None or self.coef_table.device != x.device:
            self._tables(x.device)

    # -- the CFG pair as two programs on two streams -----------------------------------------------------------------------------
    # default policy: never.  Measured at one image per GPU (profiles/r06_a_ab_cfg_split.log, r06_b_ab_cfg_split_lead.log): 26.1 ms against 25.2-25.4 ms for the
    # single program, with the tuning table's choices carried over to the halved shapes and with one half running 3 / 5 / 40 launches ahead of the other --
    # two identical launch sequences pair equal kernels, and a launch that leaves CUs idle leaves them idle in both.  Kept as an option (and as the
    # two-concurrent-programs test of the engine): `cfg_split=True` / REFINERS_AMD_CFG_SPLIT=1.
    SPLIT_MAX_IMAGES = 0

    def _split_wanted(self, n: int) -> bool:
        if self._sag_adapter() is not None or self.x is None or self.x.device.type != "cuda":
            return False
        return self.cfg_split if self.cfg_split is not None else n <= self.SPLIT_MAX_IMAGES

    def _split_got(self, got: dict[str, Any], n: int) -> tuple[dict[str, Any], dict[str, Any]]:
        """[negative ; conditional] stacks -> (negative rows, conditional rows).  The views are made once per set_inputs: a fresh view per step
        would look like a new prompt to the engines (_ident keys on id()) and re-run their prologues every step."""
        cached = getattr(self, "_pair_halves", None)
        if cached is None or cached[0] is not self.inputs:
            def cut(t: Any, h: int) -> Any:
                if t is None or t.shape[0] != 2 * n:  # ONE control picture / feature set for the whole batch broadcasts into both halves
                    return t
                return t[h * n : (h + 1) * n]

            sides = []
            for h in (0, 1):
                sides.append({"pooled": cut(got["pooled"], h), "time_ids": cut(got["time_ids"], h), "tokens": {k: cut(v, h) for k, v in got["tokens"].items()},
                              "conditions": {k: cut(v, h) for k, v in got["conditions"].items()},
                              "t2i": {k: tuple(cut(f, h) for f in feats) for k, feats in got.get("t2i", {}).items()}})
            cached = (self.inputs, sides)
            self._pair_halves = cached
        rest = {"timestep": got["timestep"], "timesteps_all": got.get("timesteps_all"), "step_index": got.get("step_index", 0)}
        return {**rest, **cached[1][0]}, {**rest, **cached[1][1]}

    def _prepare_pair(self, got: dict[str, Any], n: int) -> tuple[bool, bool]:
        """Both halves lowered / staged; (negative prologue must run, conditional prologue must run).  Raises Unsupported like prepare_explicit."""
        x = self.x
        assert x is not None
        eu = self.engine
        if self.engine_c is None:
            self.engine_c = CompiledUNet(self.unet, use_graph=False, lora_mode=eu.lora_mode)
            self.engine_c.cache = eu.cache  # one set of packed weights for both programs
        ec = self.engine_c
        shape2 = (2 * n,) + tuple(x.shape[1:])
        if self.pair_x is None or tuple(self.pair_x.shape) != shape2 or self.pair_x.device != x.device or self.pair_x.dtype != self.unet.dtype:
            self.pair_x = torch.empty(shape2, device=x.device, dtype=self.unet.dtype)
            self.pair_out = torch.empty((2 * n, eu._out_channels()) + tuple(x.shape[2:]), device=x.device, dtype=self.unet.dtype)
            eu.key = ec.key = None  # the programs hold the old buffers' addresses
        assert self.pair_out is not None
        eu.io_override, ec.io_override = (self.pair_x[:n], self.pair_out[:n]), (self.pair_x[n:], self.pair_out[n:])
        gu, gc = self._split_got(got, n)
        shape1 = (n,) + tuple(x.shape[1:])
        cu = eu.prepare_explicit(shape1, x.device, gu)
        cc = ec.prepare_explicit(shape1, x.device, gc)
        if self.side_stream is None:
            self.side_stream = torch.cuda.Stream(device=x.device)
        return cu, cc

    def _replay_pair(self) -> None:
        """The conditional half on the side stream beside the negative half on the current one (under capture: two branches of the graph)."""
        cur, side = torch.cuda.current_stream(), self.side_stream
        assert side is not None and self.engine.low is not None and self.engine_c is not None and self.engine_c.low is not None
        lead = int(os.environ.get("REFINERS_AMD_CFG_SPLIT_LEAD", "0"))  # launches the current stream's half runs ahead of the other (probing: two identical sequences in lockstep pair equal kernels)
        ops = self.engine.low.step
        if lead > 0:
            native.replay(ops[:lead])
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            native.replay(self.engine_c.low.step)
        native.replay(ops[lead:] if lead > 0 else ops)
        cur.wait_stream(side)

    def _split_step(self, step: int, got: dict[str, Any], n: int) -> Tensor:
        from types import SimpleNamespace

        try:
            cu, cc = self._prepare_pair(got, n)
        except Unsupported:
            return self._stock_step(step, got)
        eu, ec = self.engine, self.engine_c
        assert ec is not None and self.x is not None
        if cu:
            eu.run_prologue()
        if cc:
            ec.run_prologue()
        eu.check_handovers(eu.CHECK_EVERY)
        ec.check_handovers(ec.CHECK_EVERY)
        self.coef.copy_(self.coef_table[step])
        io = SimpleNamespace(x=self.pair_x, out=self.pair_out)
        gkey = ("pair", eu.key, ec.key)
        if self.linear:
            return self._linear_step(step, io, None, eu, None, gkey, run=self._replay_pair)
        if not self.use_graph:
            self._fill(io)
            self._replay_pair()
            native.cfg_ddim_step(self.x, io.out, self.coef)
            return self.x
        if self.graph is None or self.graph_key != gkey:
            keep = self.x.clone()
            self._fill(io)
            self._replay_pair()  # warm-up outside capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._fill(io)
                self._replay_pair()
                native.cfg_ddim_step(self.x, io.out, self.coef)
            self.x.copy_(keep)
            self.graph, self.graph_key = g, gkey
        self.graph.replay()
        return self.x

    def lower_now(self) -> None:
        """Build the launch programs for the staged inputs without running anything (refiners_amd.parallel.broadcast_packs lowers on every
        rank but lets only the source compute the packed weights: the prologue must not run before they have arrived)."""
        assert self.x is not None, "call set_inputs first"
        if self.coef_table is None or self.coef_table.device != self.x.device:
            self._tables(self.x.device)
        n = self.x.shape[0]
        if self._split_wanted(n):
            try:
                self._prepare_pair(dict(self.inputs, timestep=self.ts_table[0:1], timesteps_all=self.ts_table, step_index=0), n)
            except Unsupported:
                return
            self.engine.prologue_key = None
            self.engine_c.prologue_key = None  # type: ignore[union-attr]
            return
        try:
            self.engine.prepare_explicit((2 * n,) + tuple(self.x.shape[1:]), self.x.device, dict(self.inputs, timestep=self.ts_table[0:1], timesteps_all=self.ts_table, step_index=0))
        except Unsupported:
            return  # a tree this lowering does not know: nothing to stage, step() takes the stock Chain forward with its warning (the fallback contract)
        self.engine.prologue_key = None  # staged, not yet run: the first step() re-stages and runs the prologue

    def _fill(self, io: Any = None) -> None:
        io = io if io is not None else self.engine.io
        n = self.x.shape[0]  # type: ignore[union-attr]
        io.x[:n].copy_(self.x)  # type: ignore[union-attr]
        io.x[n:].copy_(self.x)  # type: ignore[union-attr]

    @torch.no_grad()
    def step(self, step: int) -> Tensor:
        assert self.x is not None, "call set_inputs first"
        if self.coef_table is None or self.coef_table.device != self.x.device:
            self._tables(self.x.device)  # condition_scale / solver changed since the last call
        eng = self.engine
        got = dict(self.inputs, timestep=self.ts_table[step : step + 1], timesteps_all=self.ts_table, step_index=step)
        n = self.x.shape[0]
        if self._split_wanted(n):
            return self._split_step(step, got, n)
        eng.io_override = None
        shape2 = (2 * n,) + tuple(self.x.shape[1:])
        try:
            changed = eng.prepare_explicit(shape2, self.x.device, got)
        except Unsupported:
            return self._stock_step(step, got)
        if changed:
            eng.run_prologue()
        eng.check_handovers(eng.CHECK_EVERY)
        io, low = eng.io, eng.low
        assert io is not None and low is not None
        self.coef.copy_(self.coef_table[step])
        sag = self._sag_adapter()
        tail = None if sag is None else self._prepare_sag(sag, got, n, io, low, step)
        # everything a captured graph holds by value or by address: both programs, and for self-attention guidance the host-side ratio
        # sag.scale / condition_scale and the blur weights' tensor (kernel_size, sigma)
        gkey = (eng.key, None if self.engine2 is None or sag is None else (self.engine2.key, float(sag.scale) / float(self.condition_scale), int(sag.kernel_size), float(sag.sigma)))
        if self.linear:
            return self._linear_step(step, io, low, eng, tail, gkey)
        if tail is None:
            tail = lambda: None  # noqa: E731
        if not self.use_graph:
            self._fill()
            native.replay(low.step)
            tail()
            native.cfg_ddim_step(self.x, io.out, self.coef)
            return self.x
        if self.graph is None or self.graph_key != gkey:
            keep = self.x.clone()
            self._fill()
            native.replay(low.step)  # warm-up outside capture (first-launch attribute calls, workspaces)
            tail()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._fill()
                native.replay(low.step)
                tail()
                native.cfg_ddim_step(self.x, io.out, self.coef)
            self.x.copy_(keep)
            self.graph, self.graph_key = g, gkey
        self.graph.replay()
        return self.x

    # -- whole-tree fallback ------------------------------------------------------------------------------------------------------
    def _stock_step(self, step: int, got: dict[str, Any]) -> Tensor:
        """A tree the lowering refused (CompiledUNet.prepare_explicit warned and recorded why): the step runs like the reference's
        `LatentDiffusionModel.forward` (latent_diffusion/model.py:128-159) -- contexts set on the tree, `unet(cat(x, x))` through the stock
        child loop -- and only the guidance + solver update stays on the native kernel."""
        assert self._sag_adapter() is None, "Self-Attention Guidance on a tree that is not lowered: use refiners' own pipeline"
        unet, x = self.unet, self.x
        assert x is not None
        unet.set_timestep(got["timestep"])
        for (ctx, key), v in got["tokens"].items():
            unet.set_context(ctx, {key: v.to(x.dtype)})
        if got.get("pooled") is not None:
            unet.set_pooled_text_embedding(got["pooled"].to(x.dtype))
            unet.set_time_ids(got["time_ids"])
        for name, v in got.get("conditions", {}).items():
            ctx, _, key = name.partition(".")
            unet.set_context(ctx, {key or "condition": v.to(x.dtype)})
        for name, feats in got.get("t2i", {}).items():
            unet.set_context("t2iadapter", {f"condition_features_{name}": tuple(f.to(x.dtype) for f in feats)})
        self.coef.copy_(self.coef_table[step])
        xin = torch.cat((x, x))
        if not self.linear:
            native.cfg_ddim_step(x, unet(xin).contiguous(), self.coef)
            return x
        assert self.hist is not None
        if getattr(self.solver, "needs_noise", None) is not None and self.solver.needs_noise(step):
            noise = torch.randn(tuple(x.shape), generator=getattr(self, "generator", None), device=getattr(self.solver, "device", x.device), dtype=getattr(self.solver, "dtype", x.dtype))
            self.hist.copy_(noise)
        s0 = float(self.solver.input_scale(step))
        y = unet(xin * s0 if s0 != 1.0 else xin).contiguous()
        native.cfg_linear_step(x, y, self.hist, None, self.coef)  # (no model-input buffer: the next step scales cat(x, x) itself)
        return x

    # -- self-attention guidance (xl/model.py:164-250) ---------------------------------------------------------------------------
    def _sag_adapter(self) -> Any:
        p = getattr(self.unet, "parent", None)
        while p is not None:
            if isa(p, "SAGAdapter"):
                return p
            p = getattr(p, "parent", None)
        return None

    def _prepare_sag(self, sag: Any, got: dict[str, Any], n: int, io: Any, low: Any, step: int) -> Any:
        """Stage the second UNet pass of Self-Attention Guidance and return the closure that runs between the CFG pass and the
        guidance + DDIM kernel: degraded latents (mask from the tapped attention's column mass, Gaussian blur, re-noising: one
        kernel) -> unconditional UNet pass on them (a second lowered program, batch n) -> cond += (sag / cfg) * (uncond - degraded),
        which makes the unchanged CFG kernel produce eps_cfg + sag_scale * (eps_uncond - eps_degraded)   (model.py:147-155)."""
        if self.sag_table is None:  # the reference raises the same way (Solver.remove_noise indexes integer tables with the solver's timesteps)
            raise IndexError(f"{type(self.solver).__name__}: add_noise / remove_noise are not defined for this solver's timesteps, so Self-Attention Guidance "
                             "cannot be evaluated (refiners raises here as well)")
        assert self.condition_scale != 0.0, "SAG with a zero guidance scale is not lowered"
        # ControlLora pictures / T2I-Adapter features stay in their contexts for the second pass (xl/model.py:186-246 swaps only the text, pooled,
        # time-id and image embeddings), which has n rows after the CFG pass's 2n: batch-1 conditions broadcast into both, anything else cannot be
        # added to the n-row stem -- torch raises RuntimeError in the reference (oracle/make_golden_sag.py --conditions), so does this
        for name, rows in [(k, v.shape[0]) for k, v in got["conditions"].items()] + [(k, f.shape[0]) for k, feats in got.get("t2i", {}).items() for f in feats]:
            if rows not in (1, n):
                raise RuntimeError(f"Self-Attention Guidance runs its second UNet pass on {n} row(s): condition '{name}' has {rows} rows and cannot be "
                                   "broadcast to it (give one picture / one set of features for the whole batch; refiners fails on this shape as well)")
        self.sag_coef.copy_(self.sag_table[step])
        assert getattr(low, "sag", None) is not None and getattr(low, "sag_shape", None) is not None, "SAG adapter present but its taps were not found in the lowered tree"
        if self.engine2 is None:
            self.engine2 = CompiledUNet(self.unet, use_graph=False, lora_mode=self.engine.lora_mode)
            self.engine2.sag_capture = False  # the reference recomputes (and discards) the attention map in this pass
        e2 = self.engine2
        # [negative ; conditional] stacks -> negative half.  The views are made once per set_inputs: a fresh view per step would look like a
        # new prompt to engine2 (_ident keys on id()) and re-run its whole prologue (text / image K, V projections) every step.
        halves = getattr(self, "_sag_halves", None)
        if halves is None or halves[0] is not self.inputs:
            half = lambda t: None if t is None else t[: t.shape[0] // 2]  # noqa: E731
            halves = (self.inputs, {"pooled": half(got["pooled"]), "time_ids": half(got["time_ids"]), "tokens": {k: half(v) for k, v in got["tokens"].items()}})
            self._sag_halves = halves
        got2 = {"timestep": got["timestep"], "timesteps_all": got.get("timesteps_all"), "step_index": got.get("step_index", 0), **halves[1],
                "conditions": got["conditions"], "t2i": got.get("t2i", {})}
        x = self.x
        if e2.prepare_explicit((n,) + tuple(x.shape[1:]), x.device, got2):
            e2.run_prologue()
        io2, low2 = e2.io, e2.low
        ks, sigma = int(sag.kernel_size), float(sag.sigma)
        key = (ks, sigma, str(x.device))
        if getattr(self, "_sag_w1_key", None) != key:
            t = torch.linspace(-(ks - 1) * 0.5, (ks - 1) * 0.5, steps=ks, dtype=torch.float32)
            pdf = torch.exp(-0.5 * (t / sigma).pow(2))
            self._sag_w1, self._sag_w1_key = (pdf / pdf.sum()).to(x.device), key
        mass, (ah, aw) = low.sag["mass"], low.sag_shape
        assert ah * aw == low.sag["tokens"]
        ratio = float(sag.scale) / float(self.condition_scale)
        u, c = io.out[:n], io.out[n:]

        def tail() -> None:
            # x is the UNSCALED latent (model.py:146: the guidance sees x, not scale_model_input(x)), and so is what the second pass is fed
            native.sag_degrade(x, u, mass, (ah, aw), self.sag_coef, self._sag_w1, io2.x)
            native.replay(low2.step)
            native.axpby(u, ratio, c, 1.0, c)
            native.axpby(io2.out, -ratio, c, 1.0, c)

        return tail

    def _linear_step(self, step: int, io: Any, low: Any, eng: Any, tail: Any = None, gkey: Any = None, run: Any = None) -> Tensor:
        """Euler / DPM-Solver++ / LCM: the UNet program (then the Self-Attention Guidance pass, `tail`) then ONE kernel (guidance, update,
        history, next model input).  `run`: what replays the UNet (default: `low.step` on the current stream; the split CFG pair passes its own)."""
        gkey = gkey if gkey is not None else eng.key
        if run is None:
            run = lambda: native.replay(low.step)  # noqa: E731
        if tail is None:
            tail = lambda: None  # noqa: E731
        assert self.x is not None and self.hist is not None
        if getattr(self.solver, "needs_noise", None) is not None and self.solver.needs_noise(step):
            # LCMSolver re-noises the consistency estimate (solvers/lcm.py:143-150): same draw as the reference (shape, device,
            # dtype, generator), placed in the kernel's history slot whose coefficient is the next timestep's noise std
            # drawn where and as what the reference draws it (solvers/lcm.py:143-150: the SOLVER's device and dtype), so that the same
            # generator and seed give the same stream; then cast / moved into the history slot
            sdev = getattr(self.solver, "device", self.x.device)
            sdt = getattr(self.solver, "dtype", self.x.dtype)
            noise = torch.randn(tuple(self.x.shape), generator=getattr(self, "generator", None), device=sdev, dtype=sdt)
            self.hist.copy_(noise)
        if not self.primed or self.primed_key != gkey:  # first step of a trajectory, or the engine re-lowered into new buffers
            self._fill(io)  # cat(x, x) ...
            s0 = float(self.solver.input_scale(step))
            if s0 != 1.0:
                io.x.mul_(s0)  # ... through Solver.scale_model_input for THIS step; later steps get it from the kernel
            self.primed, self.primed_key = True, gkey
        if not self.use_graph:
            run()
            tail()
            native.cfg_linear_step(self.x, io.out, self.hist, io.x, self.coef)
            return self.x
        if self.graph is None or self.graph_key != gkey:
            run()  # warm-up outside capture; reads io.x only
            tail()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run()
                tail()
                native.cfg_linear_step(self.x, io.out, self.hist, io.x, self.coef)
            self.graph, self.graph_key = g, gkey
        self.graph.replay()
        return self.x

    @torch.no_grad()
    def sample(self, first_step: Optional[int] = None) -> Tensor:
        """Run the trajectory from the solver's `first_inference_step` (img2img starts later than 0).  A multistep solver
        decides between its first- and second-order update by comparing the step with ITS first_inference_step
        (solvers/dpm.py:136-152), so starting anywhere else would run a second-order update on an all-zero history."""
        first = int(getattr(self.solver, "first_inference_step", 0))
        if first_step is None:
            first_step = first
        assert first_step == first or not self.linear, (
            f"sample(first_step={first_step}) but solver.first_inference_step={first}: build the solver with first_inference_step={first_step}")
        for s in range(first_step, self.solver.num_inference_steps):
            self.step(s)
        for eng in (self.engine, self.engine_c, self.engine2):  # the trajectory's last replays (the in-loop look runs every CHECK_EVERY-th step)
            if eng is not None:
                eng.check_handovers()
        return self.x  # type: ignore[return-value]
