"""ctypes binding of libmi355x_refiners.so (the C ABI declared in include/mi355x_refiners.h).

This module is the ONLY place where Python touches the native library.  It deliberately passes nothing but raw
device pointers, sizes, strides and scalars; torch is used as the owner of device memory and of the current HIP
stream.  There is no CPU or PyTorch fallback here: if the library is missing or a kernel refuses a shape, the call
raises `NativeError` (the fused fluxion nodes decide, *before* calling, whether a sub-tree is eligible).
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from pathlib import Path
from typing import Any, Optional, Sequence

import torch
from torch import Tensor

MI355X_F32 = 0
MI355X_BF16 = 1
MAX_SEG = 3

_ERR = {0: "OK", -1: "EDTYPE", -2: "ESHAPE", -3: "ELAUNCH", -4: "EARG"}

LIB_PATH = Path(__file__).resolve().parent / "csrc" / "libmi355x_refiners.so"


class NativeError(RuntimeError):
    """Raised when the native library is unavailable or a native call returns a negative status."""


class GemmSeg(C.Structure):
    _fields_ = [
        ("x", C.c_void_p),
        ("ldx", C.c_int64),
        ("w", C.c_void_p),
        ("ldw", C.c_int64),
        ("k", C.c_int32),
        ("ksize", C.c_int32),
        ("stride", C.c_int32),
        ("ups", C.c_int32),
        ("H", C.c_int32),
        ("W", C.c_int32),
        ("asym", C.c_int32),
        ("kblocked", C.c_int32),
    ]


class KBlocked:
    """A weight matrix [N, K] re-laid K-BLOCKED for mi355x_gemm: `t` is [K * itemsize / 128][N][128 / itemsize], so the
    N x 128-byte slab one K step of the kernel streams is contiguous (mi355x_gemm_seg.kblocked).  global -> LDS streaming
    runs at 10 TB/s on 2.5 KB rows but 5 TB/s on 10 KB rows and 4.5 TB/s on the 23-46 KB rows of a 3x3 convolution's
    weights (profiles/r01_aa_probe_glds.log); weights are static, so the host pays for the re-layout once."""

    def __init__(self, w: Tensor, _adopt: Optional[tuple[int, int]] = None) -> None:
        if _adopt is not None:  # `w` already IS the blocked buffer of an [n, k] matrix (written that way by a kernel)
            self.N, self.K = _adopt
            self.t = w
            return
        assert w.dim() == 2 and w.stride(1) == 1
        n, k = w.shape
        blk = 128 // w.element_size()
        assert k % blk == 0, f"K = {k} is not a multiple of {blk}"
        self.N, self.K = n, k
        self.t = w.reshape(n, k // blk, blk).permute(1, 0, 2).contiguous()

    @classmethod
    def adopt(cls, buf: Tensor, rows: int, k: int) -> "KBlocked":
        """Wrap a contiguous buffer of rows * k elements that a kernel fills in blocked order (mi355x_gemm_args.out_kblocked)."""
        assert buf.is_contiguous() and buf.numel() == rows * k
        return cls(buf, _adopt=(rows, k))

    def dense(self) -> Tensor:
        """Back to a row-major [N, K] tensor (tests / debugging)."""
        blk = 128 // self.t.element_size()
        return self.t.view(self.K // blk, self.N, blk).permute(1, 0, 2).reshape(self.N, self.K)

    @property
    def shape(self) -> tuple[int, int]:
        return (self.N, self.K)

    def data_ptr(self) -> int:
        return self.t.data_ptr()

    @property
    def dtype(self) -> torch.dtype:
        return self.t.dtype

    @property
    def device(self) -> torch.device:
        return self.t.device


MAX_PREFETCH = 2  # == MI355X_MAX_PREFETCH
LORA_R = 32  # granularity of the stacked LoRA rank mi355x_gemm handles inside the parent launch (gemm_kernel.cuh: LORA_RC)
LORA_RMAX = 128  # largest stacked rank (LORA_RMAX there)


def lora_rank(rt: int) -> int:
    """Stacked rank -> the padded rank the kernel handles (32, 64 or 128); 0 = too large for the in-launch path."""
    return next((r for r in (32, 64, 128) if rt <= r), 0)


class GemmArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32),
        ("M", C.c_int32),
        ("N", C.c_int32),
        ("nseg", C.c_int32),
        ("conv", C.c_int32),
        ("B", C.c_int32),
        ("OH", C.c_int32),
        ("OW", C.c_int32),
        ("seg", GemmSeg * MAX_SEG),
        ("out", C.c_void_p),
        ("ldo", C.c_int64),
        ("bias", C.c_void_p),
        ("rowbias", C.c_void_p),
        ("ld_rowbias", C.c_int64),
        ("rows_per_group", C.c_int32),
        ("geglu", C.c_int32),
        ("res", C.c_void_p),
        ("ldres", C.c_int64),
        ("zeros", C.c_void_p),
        ("tile", C.c_int32),
        ("ksplit", C.c_int32),
        ("ws", C.c_void_p),
        ("ws_bytes", C.c_int64),
        ("prefetch", C.c_void_p * MAX_PREFETCH),
        ("prefetch_bytes", C.c_int64 * MAX_PREFETCH),
        ("prefetch_blocks", C.c_int32),
        ("out_kblocked", C.c_int32),
        ("stages", C.c_int32),
        ("nt_begin", C.c_int32),
        ("out_t", C.c_void_p),
        ("ldt", C.c_int64),
        ("ln_stats", C.c_void_p),
        ("ln_parts", C.c_int32),
        ("ln_eps", C.c_float),
        ("ln_s", C.c_void_p),
        ("ln_c", C.c_void_p),
        ("stats_out", C.c_void_p),
        ("out_f32", C.c_int32),
        ("lora_a", C.c_void_p * 3),
        ("lora_nb", C.c_int32 * 3),
        ("lora_groups", C.c_int32),
        ("lora_r", C.c_int32),
        ("lora_b", C.c_void_p),
        ("lora_ls", C.c_void_p),
        ("lora_lc", C.c_void_p),
        ("lora_t", C.c_void_p),
        ("lora_flags", C.c_void_p),
        ("lora_epoch", C.c_void_p),
        ("colstats_out", C.c_void_p),
        ("sk_ws", C.c_void_p),
        ("sk_flags", C.c_void_p),
        ("sk_slots", C.c_int32),
    ]


class KvStream(C.Structure):
    _fields_ = [
        ("k", C.c_void_p),
        ("ldk", C.c_int64),
        ("k_batch_stride", C.c_int64),
        ("vt", C.c_void_p),
        ("ldvt", C.c_int64),
        ("vt_batch_stride", C.c_int64),
        ("Lk", C.c_int32),
        ("out_scale", C.c_float),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32),
        ("B", C.c_int32),
        ("H", C.c_int32),
        ("D", C.c_int32),
        ("Lq", C.c_int32),
        ("nstream", C.c_int32),
        ("q", C.c_void_p),
        ("ldq", C.c_int64),
        ("q_batch_stride", C.c_int64),
        ("out", C.c_void_p),
        ("ldo", C.c_int64),
        ("o_batch_stride", C.c_int64),
        ("scale", C.c_float),
        ("kv", KvStream * 2),
    ]


class AttnGeneralArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32),
        ("B", C.c_int32),
        ("H", C.c_int32),
        ("Lq", C.c_int32),
        ("Lk", C.c_int32),
        ("Dqk", C.c_int32),
        ("Dv", C.c_int32),
        ("causal", C.c_int32),
        ("q", C.c_void_p),
        ("ldq", C.c_int64),
        ("q_batch_stride", C.c_int64),
        ("k", C.c_void_p),
        ("ldk", C.c_int64),
        ("k_batch_stride", C.c_int64),
        ("vt", C.c_void_p),
        ("ldvt", C.c_int64),
        ("vt_batch_stride", C.c_int64),
        ("out", C.c_void_p),
        ("ldo", C.c_int64),
        ("o_batch_stride", C.c_int64),
        ("scale", C.c_float),
        ("out_scale", C.c_float),
    ]


class LayerNormArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32),
        ("M", C.c_int32),
        ("C", C.c_int32),
        ("x", C.c_void_p),
        ("ldx", C.c_int64),
        ("gamma", C.c_void_p),
        ("beta", C.c_void_p),
        ("eps", C.c_float),
        ("out", C.c_void_p),
        ("ldo", C.c_int64),
    ]


class GroupNormArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32),
        ("B", C.c_int32),
        ("HW", C.c_int32),
        ("C", C.c_int32),
        ("G", C.c_int32),
        ("x", C.c_void_p),
        ("ldx", C.c_int64),
        ("gamma", C.c_void_p),
        ("beta", C.c_void_p),
        ("eps", C.c_float),
        ("silu", C.c_int32),
        ("out", C.c_void_p),
        ("ldo", C.c_int64),
        ("ws", C.c_void_p),
        ("colstats", C.c_void_p),
        ("x2", C.c_void_p),
        ("ldx2", C.c_int64),
        ("C1", C.c_int32),
        ("colstats2", C.c_void_p),
    ]


#: every symbol include/mi355x_refiners.h declares (tests check that the library exports all of them)
EXPORTS = [
    "mi355x_abi_version",
    "mi355x_device_info",
    "mi355x_gemm",
    "mi355x_epoch_bump",
    "mi355x_attention",
    "mi355x_attention_general",
    "mi355x_layernorm",
    "mi355x_groupnorm_ws_floats",
    "mi355x_groupnorm",
    "mi355x_nchw_to_nhwc",
    "mi355x_nhwc_to_nchw",
    "mi355x_im2col3x3_nchw",
    "mi355x_concat2",
    "mi355x_axpby",
    "mi355x_silu",
    "mi355x_softmax_rows",
    "mi355x_colsum_rows",
    "mi355x_sag_degrade",
    "mi355x_cfg_ddim_step",
    "mi355x_cfg_linear_step",
    "mi355x_sinusoidal",
    "mi355x_patchify_nchw",
    "mi355x_gather_rows",
    "mi355x_pointwise_nchw",
    "mi355x_relpos_pack",
]

_lib: Optional[C.CDLL] = None
_lib_path: Optional[str] = None


def load(path: Optional[Path] = None) -> C.CDLL:
    """Load the shared library (once). Raises NativeError if it has not been built."""
    global _lib, _lib_path
    if _lib is not None:
        return _lib
    p = Path(path) if path else Path(os.environ.get("REFINERS_AMD_LIB") or LIB_PATH)  # (REFINERS_AMD_LIB: an experiment build, tools only)
    if not p.exists():
        raise NativeError(
            f"{p} is missing: build it with `python -m refiners_amd.build_native` (or __graft_entry__.build()). "
            "There is no fallback path for the MI355X kernels."
        )
    try:
        lib = C.CDLL(str(p))
    except OSError as e:  # e.g. no HIP runtime on this machine
        raise NativeError(f"cannot load {p}: {e}") from e
    lib.mi355x_abi_version.restype = C.c_int
    if os.environ.get("REFINERS_AMD_FORCE_TILE"):  # probing: every GEMM / conv launch that can run on this tile configuration does (tools, A/B runs)
        lib.mi355x_set_option(b"tile", int(os.environ["REFINERS_AMD_FORCE_TILE"]))
    lib.mi355x_device_info.argtypes = [C.c_char_p, C.c_int32]
    lib.mi355x_gemm.argtypes = [C.POINTER(GemmArgs), C.c_void_p]
    lib.mi355x_epoch_bump.argtypes = [C.c_void_p, C.c_void_p]
    lib.mi355x_attention.argtypes = [C.POINTER(AttnArgs), C.c_void_p]
    lib.mi355x_attention_general.argtypes = [C.POINTER(AttnGeneralArgs), C.c_void_p]
    lib.mi355x_layernorm.argtypes = [C.POINTER(LayerNormArgs), C.c_void_p]
    lib.mi355x_groupnorm.argtypes = [C.POINTER(GroupNormArgs), C.c_void_p]
    lib.mi355x_groupnorm_ws_floats.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.mi355x_groupnorm_ws_floats.restype = C.c_int64
    lib.mi355x_nchw_to_nhwc.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]
    lib.mi355x_nhwc_to_nchw.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]
    lib.mi355x_im2col3x3_nchw.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]
    lib.mi355x_concat2.argtypes = [C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    lib.mi355x_axpby.argtypes = [C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]
    lib.mi355x_silu.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.mi355x_colsum_rows.argtypes = [C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_float, C.c_void_p]
    lib.mi355x_sag_degrade.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_void_p]
    lib.mi355x_softmax_rows.argtypes = [C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_void_p]
    lib.mi355x_cfg_ddim_step.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.mi355x_cfg_linear_step.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.mi355x_sinusoidal.argtypes = [C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
    lib.mi355x_patchify_nchw.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]
    lib.mi355x_gather_rows.argtypes = [C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]
    lib.mi355x_pointwise_nchw.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]
    lib.mi355x_relpos_pack.argtypes = [C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.mi355x_set_option.argtypes = [C.c_char_p, C.c_int]
    lib.mi355x_attention_set_glds.argtypes = [C.c_int]
    lib.mi355x_attention_general_set_fast.argtypes = [C.c_int]
    lib.mi355x_attention_set_pipeline.argtypes = [C.c_int, C.c_int]
    if lib.mi355x_abi_version() != 7:
        raise NativeError("libmi355x_refiners.so ABI version mismatch")
    _lib = lib
    _lib_path = str(p)
    attention_pipeline_from_env()
    return lib


def attention_pipeline_from_env() -> None:
    """A/B: REFINERS_AMD_ATTN_PIPE="<K/V tiles in flight 1|2>,<XCD-aware block order 0|1>,<OPT bits of attn_kernel>,<short grids: 0 auto = 16-query waves where the grid is short|1 never|2 key-split workgroups always|3 16-query waves always>,<software-pipelined loop 0 off|1 register-staged|2 = 1 without the pinned order|3 LDS-DMA>,<short-K/V kernel: 0 default|1 128-query workgroups|2 64-query workgroups, +4 = the register-staged form>"
    (default 1,1,13,0,3,0; "/" separates as well); read at launch / capture time."""
    import os

    d, x, o, k, sp, sh = (os.environ.get("REFINERS_AMD_ATTN_PIPE", "").replace("/", ",").split(",") + ["", "", "", "", "", ""])[:6]
    _lib.mi355x_attention_set_pipeline(int(d or 1) | (int(o or 13) << 4) | (int(k or 0) << 16) | (int(sp or 3) << 19) | (int(sh or 0) << 23), int(x or 1))


def available() -> bool:
    try:
        load()
        return True
    except NativeError:
        return False


def switch_library(path: Optional[Path]) -> C.CDLL:
    """Tools only (tools/ab_step.py): make another build of the library the one new launches / recordings bind to (None = the product library).
    Programs recorded earlier keep the entry points they were recorded with."""
    global _lib
    _lib = None
    return load(path)


def loaded_library_path() -> Optional[str]:
    """The shared library the process is bound to (the product library unless a tool switched to an experiment build), None before load()."""
    return _lib_path if _lib is not None else None


def check(status: int, what: str) -> None:
    if status != 0:
        raise NativeError(f"{what} failed with MI355X_{_ERR.get(status, status)}")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return MI355X_F32
    if dt == torch.bfloat16:
        return MI355X_BF16
    raise NativeError(f"unsupported dtype {dt} (the MI355X path computes in float32 or bfloat16)")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def device_info() -> str:
    buf = C.create_string_buffer(256)
    check(load().mi355x_device_info(buf, 256), "mi355x_device_info")
    return buf.value.decode()



# ------------------------------------------------------------------------------------------------ launch / record
_recorder: Optional[list] = None


class recording:
    """Context manager: native calls made inside are appended to `ops` instead of being launched.

    The engine lowers a Chain tree once into such a list (all argument structs prebuilt, all buffers static) and then
    replays it, directly or under HIP-graph capture.  An entry is (cfunc, args, name, keepalive)."""

    def __init__(self, ops: list) -> None:
        self.ops = ops

    def __enter__(self) -> list:
        global _recorder
        self._saved = _recorder
        _recorder = self.ops
        return self.ops

    def __exit__(self, *exc: object) -> None:
        global _recorder
        _recorder = self._saved


class not_recording:
    """Context manager: native calls made inside are LAUNCHED even while a program is being recorded (weight preparation at lowering time)."""

    def __enter__(self) -> None:
        global _recorder
        self._saved = _recorder
        _recorder = None

    def __exit__(self, *exc: object) -> None:
        global _recorder
        _recorder = self._saved


def matmul_f32(x: Tensor, w: Tensor, res: Optional[Tensor] = None) -> Tensor:
    """x [M, K] @ w [N, K]^T (+ res) in float32 on the library's own f32 MFMA path (v_mfma_f32_16x16x4_f32), launched at once.  For the
    set-up arithmetic of the engine (LoRA merges, LayerNorm folding): no vendor BLAS inside the product package.  K is zero-padded to the
    kernel's 32-float granularity."""
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and x.dtype == torch.float32 and w.dtype == torch.float32
    Kp = (K + 31) // 32 * 32

    def prep(t: Tensor) -> Tensor:
        if Kp == K and t.is_contiguous() and t.data_ptr() % 16 == 0:
            return t
        p = torch.zeros(t.shape[0], Kp, dtype=torch.float32, device=t.device)
        p[:, :K] = t
        return p

    out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    with not_recording():
        gemm([(prep(x), prep(w))], out, res=res)
    return out


def _launch(sym: str, args: tuple, what: str, keep: tuple = ()) -> None:
    fn = getattr(load(), sym)
    if _recorder is not None:
        _recorder.append((fn, args, what, keep))
        return
    check(fn(*args, stream_ptr()), what)


def record_python(fn, what: str = "python") -> bool:
    """Record a Python callable (torch glue or an unfused fallback sub-tree) as a step of the program being recorded.
    Returns False when nothing is recording (the caller then runs `fn` itself)."""
    if _recorder is None:
        return False
    _recorder.append((None, fn, what, ()))
    return True


def replay(ops: list) -> None:
    """Launch a recorded program on the current stream."""
    s = stream_ptr()
    for fn, args, what, _ in ops:
        if fn is None:
            args()
            continue
        st = fn(*args, s)
        if st:
            check(st, what)

_zeros: dict[int, Tensor] = {}


def zero_page(device: torch.device) -> Tensor:
    idx = (device.index if device.index is not None else torch.cuda.current_device()) if device.type == "cuda" else -1
    z = _zeros.get(idx)
    if z is None:
        z = torch.zeros(256, dtype=torch.uint8, device=device)
        _zeros[idx] = z
    return z


# ------------------------------------------------------------------------------------------------ weight packing
def geglu_pack_index(n_out: int, device: torch.device | str = "cpu") -> Tensor:
    """Row order of a GEGLU-fused Linear(C -> 2*n_out): packed row p holds reference row index[p].

    Packed rows come in blocks of 64 = 32 value rows + 32 gate rows of the same 32 output columns, arranged so that the
    lane that owns packed columns 16g+4j+r (j = 0..3, r = 0..3) of a block holds value columns 8g+4j+r (j < 2) and
    the matching gate columns (j >= 2): see gemm.hip's epilogue.
    """
    assert n_out % 32 == 0, "GEGLU fusion needs the output width to be a multiple of 32"
    p = torch.arange(2 * n_out, device=device)
    blk, q = p // 64, p % 64
    g, j, r = q // 16, (q // 4) % 4, q % 4
    u = 8 * g + 4 * (j % 2) + r
    return torch.where(j >= 2, n_out + 32 * blk + u, 32 * blk + u)


def pack_conv_weight(w: Tensor) -> Tensor:
    """OIHW conv weight -> [O][kh*kw*I] rows with K ordered (ky, kx, channel), as conv mode of mi355x_gemm expects."""
    o, i, kh, kw = w.shape
    return w.permute(0, 2, 3, 1).reshape(o, kh * kw * i).contiguous()


# ------------------------------------------------------------------------------------------------ in-launch LoRA hand-off state
class LoraSync:
    """Device-side state of mi355x_gemm's in-launch LoRA (include/mi355x_refiners.h: lora_t / lora_flags / lora_epoch): the epoch word that
    a recorded program bumps once per replay (`bump_op()` goes to the head of the program), per-site flag arrays (zeroed, never shared
    between launches of one replay) and scratch for t = x A^T, which consecutive launches may share (stream order).

    The flag arrays of all sites are cut from a few large int32 chunks, so that the sites' error words (the int32 behind each site's last
    flag, raised by a tile whose hand-over never arrived) can be consulted with ONE gather per chunk: `pending()` is what the engines poll
    at their host sync points (CompiledUNet.check_handovers)."""

    CHUNK = 1 << 20  # int32 words per chunk (4 MB): the SDXL step's 722 sites need ~0.1 M

    def __init__(self, device: torch.device) -> None:
        self.device = device
        self.epoch = torch.zeros(1, dtype=torch.int32, device=device)
        self.chunks: list[Tensor] = []
        self.err_pos: list[list[int]] = []   # per chunk: positions of the sites' error words
        self._err_idx: list[Optional[Tensor]] = []  # the same as device index tensors (built on first use, dropped when a site is added)
        self.cursor = 0

    def flags(self, groups: int, M: int) -> Tensor:
        n = groups * ((M + 31) // 32) + 1  # (+ the launch's error word)
        if not self.chunks or self.cursor + n > self.chunks[-1].numel():
            self.chunks.append(torch.zeros(max(self.CHUNK, n), dtype=torch.int32, device=self.device))
            self.err_pos.append([])
            self._err_idx.append(None)
            self.cursor = 0
        f = self.chunks[-1][self.cursor : self.cursor + n]
        self.err_pos[-1].append(self.cursor + n - 1)
        self._err_idx[-1] = None
        self.cursor += (n + 3) // 4 * 4  # sites start on 16-byte boundaries
        return f

    def reset(self) -> None:
        """Forget every site (eager calls: one launch at a time, fresh flags and a fresh error word per call)."""
        if self.chunks:
            self.chunks, self.err_pos, self._err_idx = self.chunks[:1], [[]], [None]
            self.chunks[0].zero_()
        self.cursor = 0

    def pending(self) -> Optional[Tensor]:
        """A 0-d bool tensor on the device: some site's error word is raised (None: no site yet).  No host synchronisation."""
        parts = []
        for i, (chunk, pos) in enumerate(zip(self.chunks, self.err_pos)):
            if not pos:
                continue
            if self._err_idx[i] is None:
                self._err_idx[i] = torch.tensor(pos, dtype=torch.int64, device=self.device)
            parts.append(chunk[self._err_idx[i]].any())
        if not parts:
            return None
        return parts[0] if len(parts) == 1 else torch.stack(parts).any()

    def clear_errors(self) -> None:
        for i, (chunk, pos) in enumerate(zip(self.chunks, self.err_pos)):
            if pos:
                if self._err_idx[i] is None:
                    self._err_idx[i] = torch.tensor(pos, dtype=torch.int64, device=self.device)
                chunk[self._err_idx[i]] = 0

    def check(self) -> None:
        """Raise if a launch of this state's sites reported a hand-over that never arrived (mi355x_gemm_args.lora_flags: the last word)."""
        bad = self.pending()
        if bad is not None and bool(bad.item()):
            self.clear_errors()
            raise NativeError("mi355x_gemm (in-launch LoRA): a tile waited 2 s for t = x A^T that never came (MI355X_ELAUNCH)")

    def scratch(self, groups: int, M: int, r: int, dtype: torch.dtype) -> Tensor:
        return torch.empty(lora_scratch_rows(groups, M, r, dtype) * r, dtype=dtype, device=self.device)

    def bump_op(self) -> tuple:
        return (getattr(load(), "mi355x_epoch_bump"), (self.epoch.data_ptr(),), "mi355x_epoch_bump", (self,))

    def bump(self) -> None:
        check(load().mi355x_epoch_bump(self.epoch.data_ptr(), stream_ptr()), "mi355x_epoch_bump")


_eager_sync: dict[int, "LoraSync"] = {}


def lora_scratch_rows(groups: int, M: int, R: int, dtype: torch.dtype) -> int:
    """Rows of R elements the in-launch LoRA hand-off scratch needs: every group's block starts on a 128-byte line (mi355x_gemm_args.lora_t)."""
    rb = R * (4 if dtype == torch.float32 else 2)
    return groups * (((M * rb + 127) // 128 * 128) // rb)


def _lora_fill(a: GemmArgs, lora: tuple, ln_given: bool, dtype: torch.dtype, K: int, keep: list, sync: Optional[tuple]) -> None:
    """lora = ([(first column of the group, K-blocked stacked down rows [R, K])], pre-scaled up rows [N, R] (, sA, cA float32 [groups, R])).
    sync = (t scratch, flags, LoraSync) from the engine; None (eager calls: tests, probes) = fresh buffers and an epoch bump per call."""
    groups, lb = lora[0], lora[1]
    R = lb.shape[1]
    assert 1 <= len(groups) <= 3 and lb.dim() == 2 and lb.shape[0] == a.N and R in (32, 64, 128) and lb.is_contiguous() and lb.dtype == dtype
    for g, (nb, la) in enumerate(groups):
        assert isinstance(la, KBlocked) and la.shape == (R, K) and la.dtype == dtype, (la.shape, R, K)
        a.lora_a[g], a.lora_nb[g] = la.data_ptr(), nb
    a.lora_groups, a.lora_r, a.lora_b = len(groups), R, lb.data_ptr()
    if len(lora) > 2:  # LayerNorm folded into this launch as well
        ls_, lc_ = lora[2], lora[3]
        assert ln_given and ls_.dtype == torch.float32 and lc_.dtype == torch.float32 and ls_.is_contiguous() and lc_.is_contiguous()
        assert ls_.numel() == len(groups) * R and lc_.numel() == len(groups) * R
        a.lora_ls, a.lora_lc = ls_.data_ptr(), lc_.data_ptr()
    if sync is None:
        assert _recorder is None, "a recorded program passes its own LoRA hand-off buffers (Lowering.lora_sync)"
        dev = lb.device
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        ls = _eager_sync.setdefault(idx, LoraSync(dev))
        ls.reset()
        ls.bump()
        sync = (ls.scratch(len(groups), a.M, R, dtype), ls.flags(len(groups), a.M), ls)
    t, flags, ls = sync
    assert t.numel() >= lora_scratch_rows(len(groups), a.M, R, dtype) * R and t.dtype == dtype and t.data_ptr() % 128 == 0
    assert flags.numel() >= len(groups) * ((a.M + 31) // 32) + 1 and flags.dtype == torch.int32  # (+ the error word)
    a.lora_t, a.lora_flags, a.lora_epoch = t.data_ptr(), flags.data_ptr(), ls.epoch.data_ptr()
    keep.append((lora, t, flags, ls))


# ------------------------------------------------------------------------------------------------ call wrappers
def _seg_plain(x: Any, w: Any) -> tuple:
    """(x, ldx, w, ldw, k, kblocked flags) of one plain K segment; either operand may be a KBlocked weight."""
    assert tuple(x.shape)[1] == tuple(w.shape)[1], (x.shape, w.shape)
    flags = (1 if isinstance(w, KBlocked) else 0) | (2 if isinstance(x, KBlocked) else 0)
    ldx = x.K if isinstance(x, KBlocked) else x.stride(0)
    ldw = w.K if isinstance(w, KBlocked) else w.stride(0)
    for t in (x, w):
        assert isinstance(t, KBlocked) or (t.dim() == 2 and t.stride(1) == 1)
    return (x, ldx, w, ldw, tuple(x.shape)[1], flags)


def gemm(
    segs: Sequence[tuple[Tensor, Tensor]],
    out: Optional[Tensor],
    *,
    bias: Optional[Tensor] = None,
    rowbias: Optional[Tensor] = None,
    rows_per_group: int = 1,
    res: Optional[Tensor] = None,
    geglu: bool = False,
    gelu: bool = False,
    M: Optional[int] = None,
    N: Optional[int] = None,
    tile: int = 0,
    stages: int = 0,
    ksplit: int = 1,
    ws: Optional[Tensor] = None,
    prefetch: Optional[Tensor] = None,
    weight_operand: str = "w",
    out_kblocked: bool = False,
    out_t: Optional[Tensor] = None,
    nt_begin: int = 0,
    ln: Optional[tuple[Tensor, Tensor, Tensor, float]] = None,
    stats_out: Optional[Tensor] = None,
    out_f32: bool = False,
    lora: Optional[tuple[Sequence[tuple[int, "KBlocked"]], Tensor]] = None,
    lora_sync: Optional[tuple] = None,
    colstats_out: Optional[Tensor] = None,
) -> Optional[Tensor]:
    """out[M,N] = epi(sum_s x_s @ w_s^T) for plain row-major 2-D segments (x_s: [M,K_s], w_s: [N,K_s]).
    weight_operand="x" marks launches whose PARAMETERS sit in the x slot (transposed projections) for link_weight_prefetch.
    out_t / nt_begin: columns >= nt_begin are written transposed, out_t[n - nt_begin][m] (`out` then has nt_begin columns).
    ln = (stats [parts, M, 2] float32, s [N] float32, c [N] float32, eps): LayerNorm of x folded into this launch (see the header);
    stats_out [N / 32, M, 2] float32: per-row (mean, M2) of every 32-column chunk of the stored output."""
    a = GemmArgs()
    a.weight_is_x = weight_operand == "x"
    a.out_kblocked = int(out_kblocked)
    if prefetch is not None:
        a.prefetch[0], a.prefetch_bytes[0] = prefetch.data_ptr(), prefetch.numel() * prefetch.element_size()
    x0, w0 = segs[0]
    a.dtype = dtype_code(x0.dtype)
    a.M = M if M is not None else x0.shape[0]
    a.N = N if N is not None else w0.shape[0]
    a.nseg = len(segs)
    a.conv = 0
    keep = []
    for s, (x, w) in enumerate(segs):
        t = _seg_plain(x, w)
        sg = a.seg[s]
        sg.x, sg.ldx, sg.w, sg.ldw, sg.k = t[0].data_ptr(), t[1], t[2].data_ptr(), t[3], t[4]
        sg.ksize, sg.stride, sg.ups, sg.H, sg.W, sg.kblocked = 1, 1, 1, 0, 0, t[5]
        keep.append((x, w))
    assert not (geglu and gelu)
    if out_t is not None:
        assert out_t.dim() == 2 and out_t.stride(1) == 1 and out_t.shape[0] >= a.N - nt_begin and out_t.shape[1] >= a.M
        a.out_t, a.ldt, a.nt_begin = out_t.data_ptr(), out_t.stride(0), nt_begin
    if out is None:
        assert out_t is not None and nt_begin == 0
        a.out, a.ldo = None, 0
        a.bias = bias.data_ptr() if bias is not None else None
        a.rowbias, a.ld_rowbias, a.rows_per_group, a.res, a.ldres, a.geglu = None, 0, 1, None, 0, 0
    else:
        _fill_epilogue(a, out, bias, rowbias, rows_per_group, res, (3 if gelu == "quick" else 2) if gelu else geglu)
    if out_f32:
        assert out is not None and out.dtype == torch.float32
        a.out_f32 = 1
    if lora is not None:  # LoRA inside this launch (see _lora_fill)
        _lora_fill(a, lora, ln is not None, x0.dtype, tuple(x0.shape)[1], keep, lora_sync)
    if ln is not None:
        stats, ls, lc, eps = ln
        assert stats.dtype == torch.float32 and stats.is_contiguous() and stats.dim() == 3 and stats.shape[1] == a.M and stats.shape[2] == 2
        assert ls.dtype == torch.float32 and lc.dtype == torch.float32 and ls.numel() == a.N and lc.numel() == a.N and bias is None
        a.ln_stats, a.ln_parts, a.ln_eps, a.ln_s, a.ln_c = stats.data_ptr(), stats.shape[0], eps, ls.data_ptr(), lc.data_ptr()
        keep.append(ln)
    if stats_out is not None:
        assert stats_out.dtype == torch.float32 and stats_out.is_contiguous() and stats_out.numel() >= (a.N // 32) * a.M * 2 and a.N % 64 == 0
        a.stats_out = stats_out.data_ptr()
        keep.append(stats_out)
    _fill_colstats(a, colstats_out, keep)
    _fill_split(a, tile, ksplit, ws, stages, operand=out if out is not None else out_t)
    _launch("mi355x_gemm", (C.byref(a),), "mi355x_gemm", keep=tuple(keep))
    return out


def conv_gemm(
    segs: Sequence[tuple],
    out: Tensor,
    B: int,
    OH: int,
    OW: int,
    *,
    bias: Optional[Tensor] = None,
    rowbias: Optional[Tensor] = None,
    rows_per_group: int = 1,
    res: Optional[Tensor] = None,
    tile: int = 0,
    ksplit: int = 1,
    ws: Optional[Tensor] = None,
    stages: int = 0,
    lora: Optional[tuple] = None,
    lora_sync: Optional[tuple] = None,
    colstats_out: Optional[Tensor] = None,
    table_may_replace_split: bool = False,
) -> Tensor:
    """Implicit-GEMM convolution over NHWC images.

    segs: (image [B,H,W,C] NHWC-contiguous or channel-sliced view, packed weight [N, ksize*ksize*C], ksize, stride, ups).
    out: [B*OH*OW, N] rows (i.e. NHWC output).
    lora = ([(0, K-blocked packed down-conv weights [R, ksize*ksize*C])], pre-scaled 1x1 up weights [N, R]): Conv2dLora on segment 0 inside this launch.
    """
    a = GemmArgs()
    img0, w0 = segs[0][0], segs[0][1]
    a.dtype = dtype_code(img0.dtype)
    a.M = B * OH * OW
    a.N = w0.shape[0]
    a.nseg = len(segs)
    a.conv = 1
    a.B, a.OH, a.OW = B, OH, OW
    for s, seg in enumerate(segs):
        img, w, ksize, stride, ups = seg[:5]
        asym = seg[5] if len(seg) > 5 else 0
        assert img.dim() == 4 and img.stride(3) == 1, "conv segment must be an NHWC tensor [B,H,W,C]"
        b, h, wd, c = img.shape
        assert img.stride(1) == wd * img.stride(2) and img.stride(0) == h * img.stride(1), "pixels must be uniformly strided"
        assert tuple(w.shape)[1] == ksize * ksize * c and (isinstance(w, KBlocked) or w.stride(1) == 1)
        sg = a.seg[s]
        sg.x, sg.ldx, sg.w, sg.ldw, sg.k = img.data_ptr(), img.stride(2), w.data_ptr(), (w.K if isinstance(w, KBlocked) else w.stride(0)), c
        sg.ksize, sg.stride, sg.ups, sg.H, sg.W, sg.asym = ksize, stride, ups, h, wd, asym
        sg.kblocked = 1 if isinstance(w, KBlocked) else 0
    a.zeros = zero_page(img0.device).data_ptr()
    _fill_epilogue(a, out, bias, rowbias, rows_per_group, res, False)
    keep: list = []
    if lora is not None:
        _lora_fill(a, lora, False, img0.dtype, tuple(w0.shape)[1], keep, lora_sync)
    _fill_colstats(a, colstats_out, keep)
    _fill_split(a, tile, ksplit, ws, stages, table_may_replace_split=table_may_replace_split, operand=out)
    _launch("mi355x_gemm", (C.byref(a),), "mi355x_gemm(conv)", keep=tuple(keep))
    return out


def weight_spans(a: GemmArgs) -> list[tuple[int, int]]:
    """(address, bytes) of the weight operands of a recorded GEMM / conv launch, one per K segment: N rows of ldw elements,
    the last row counted to its K-th element only (the operand may be a column slice of a wider tensor); for a launch
    recorded with weight_operand="x" -- the transposed projections -- M rows of ldx elements."""
    es = 4 if a.dtype == 0 else 2
    out = []
    for s in range(a.nseg):
        sg = a.seg[s]
        k = int(sg.k) * (int(sg.ksize) ** 2 if a.conv else 1)
        if getattr(a, "weight_is_x", False):
            out.append((int(sg.x or 0), (int(a.M) * k if sg.kblocked & 2 else (int(a.M) - 1) * int(sg.ldx) + k) * es))
        else:
            out.append((int(sg.w or 0), (int(a.N) * k if sg.kblocked & 1 else (int(a.N) - 1) * int(sg.ldw) + k) * es))
    return out


def link_weight_prefetch(ops: list, enable: bool = True, min_bytes: int = 1 << 19, bytes_per_block: int = 128 << 10, min_blocks: Optional[int] = None,
                         max_blocks: Optional[int] = None) -> dict:
    """Post-pass over a recorded program that is replayed again and again (a denoising step, an encoder): every GEMM / conv
    launch gets the weight operands of the NEXT GEMM / conv launch as its `prefetch` spans (the last one wraps around to the
    first).  Weights are read exactly once per replay from HBM (5.1 GB per SDXL step), so without this every kernel starts on
    cold lines and its short K loop cannot hide DRAM latency: in place, launches run 25-40 % slower than the same launch on
    hot weights; with the weights pulled into the Infinity Cache one launch ahead the step is 7-8 % shorter
    (tools/probe_prefetch.py, tools/probe_step.py, profiles/r01_t*).  A burst of 32-128 prefetch workgroups at the head of
    the grid beats a thin continuous stream: one workgroup sustains only ~20 GB/s of misses.  Prefetch workgroups per launch: 32 ... 128 by the
    bytes to pull; REFINERS_AMD_PF_BLOCKS="min-max" moves the bounds (an A/B lever: these workgroups sit in front of the launch's own tiles)."""
    import os

    lo, hi = (os.environ.get("REFINERS_AMD_PF_BLOCKS", "").replace("-", ",") + ",").split(",")[:2]  # "min,max" or "min-max" (tools/ab_step.py splits its variants on commas)
    min_blocks = min_blocks if min_blocks is not None else int(lo or 32)
    max_blocks = max(min_blocks, max_blocks if max_blocks is not None else int(hi or 128))
    gemms = [e[1][0]._obj for e in ops if e[0] is not None and e[2].startswith("mi355x_gemm")]
    linked = nbytes_total = 0
    for i, a in enumerate(gemms):
        for s in range(MAX_PREFETCH):
            a.prefetch[s], a.prefetch_bytes[s] = None, 0
        a.prefetch_blocks = 0
        if not enable or len(gemms) < 2:
            continue
        # (round 6 also tried the next launch's LoRA up rows [N, R] in the free second slot -- they sit on the path between a tile's last MFMA and its epilogue --:
        #  25.54 against 25.53 ms, profiles/r06_v_ab_pf_lora.log: all adapters' rows together are ~100 MB and stay in the Infinity Cache from step to step)
        spans = [(p, b) for p, b in weight_spans(gemms[(i + 1) % len(gemms)]) if p and b >= min_bytes][:MAX_PREFETCH]
        if not spans:
            continue
        for s, (p, b) in enumerate(spans):
            a.prefetch[s], a.prefetch_bytes[s] = p, b
        total = sum(b for _, b in spans)
        a.prefetch_blocks = max(min_blocks, min(max_blocks, -(-total // bytes_per_block)))
        linked += 1
        nbytes_total += total
    return {"linked": linked, "bytes": nbytes_total, "launches": len(gemms)}


def gemm_signature(a: GemmArgs) -> str:
    """Shape class of a GEMM / conv launch: the key of the measured tile table (refiners_amd/engine/tuning.py)."""
    k = sum(int(a.seg[s].k) * (int(a.seg[s].ksize) ** 2 if a.conv else 1) for s in range(a.nseg))
    flags = ("geglu" if a.geglu == 1 else "") + ("T%d" % a.nt_begin if a.out_t else "") + ("ln" if a.ln_stats else "") + ("st" if a.stats_out else "") + ("lora" if a.lora_b else "")
    return f"{'conv' if a.conv else 'gemm'}:{'f32' if a.dtype == 0 else 'bf16'}:{a.M}x{a.N}x{k}:s{a.nseg}:{flags}"


def colstats_shape(M: int, N: int) -> tuple[int, int, int]:
    """Shape of the float32 buffer mi355x_gemm_args.colstats_out fills: (sum, sum of squares) per (32-row block, column)."""
    return ((M + 31) // 32, N, 2)


def _fill_colstats(a: GemmArgs, cs: Optional[Tensor], keep: list) -> None:
    if cs is None:
        return
    assert cs.dtype == torch.float32 and cs.is_contiguous() and cs.numel() >= (a.M + 31) // 32 * a.N * 2 and a.N % 16 == 0
    a.colstats_out = cs.data_ptr()
    keep.append(cs)


class StreamK:
    """Scratch of the stream-K launches (tile 8; mi355x_gemm_args.sk_ws / sk_flags): 256 slots of 256 KB + the hand-off flags.  Launches on ONE stream
    share it (a recorded program owns one: Lowering._sk, installed by Lowering._Section while the program is recorded; eager calls use one per
    (operand device, stream): _fill_split); allocated on first use."""

    SLOTS = 256

    def __init__(self, device: torch.device) -> None:
        self.device = device
        self.ws: Optional[Tensor] = None
        self.flags: Optional[Tensor] = None

    def tensors(self) -> tuple[Tensor, Tensor]:
        if self.ws is None:
            self.ws = torch.empty(self.SLOTS * 65536, dtype=torch.float32, device=self.device)
            self.flags = torch.zeros(self.SLOTS + 1, dtype=torch.int32, device=self.device)
        return self.ws, self.flags

    def pending(self) -> Optional[Tensor]:
        """A 0-d bool tensor on the device: a launch reported a lost deposit (None: the scratch was never used).  No host synchronisation."""
        return None if self.flags is None else self.flags[-1] != 0

    def check(self) -> None:
        """Raise if a launch reported a lost deposit.  ALL flags are reset then: an owner that gave up has cleared its range, but a late depositor may
        still have set its flag afterwards, and the next launch on this scratch would find it set at once and add a stale slot (round-5 advisor)."""
        bad = self.pending()
        if bad is not None and bool(bad.item()):
            self.flags.zero_()  # type: ignore[union-attr]
            raise NativeError("mi355x_gemm (stream-K): a workgroup's partial tile never arrived (MI355X_ELAUNCH)")


_streamk_current: Optional[StreamK] = None
_streamk_eager: dict[tuple[int, int], StreamK] = {}


def set_streamk(sk: Optional[StreamK]) -> Optional[StreamK]:
    """Make `sk` the scratch that tile-8 launches made from now on carry (a Lowering installs its own while it records); returns the previous one."""
    global _streamk_current
    old, _streamk_current = _streamk_current, sk
    return old


def attach_streamk(a: GemmArgs, sk: StreamK) -> None:
    w, f = sk.tensors()
    a.sk_ws, a.sk_flags, a.sk_slots = w.data_ptr(), f.data_ptr(), sk.SLOTS
    a._sk_keep = sk


def _fill_split(a: GemmArgs, tile: int, ksplit: int, ws: Optional[Tensor], stages: int = 0, table_may_replace_split: bool = False, operand: Optional[Tensor] = None) -> None:
    """tile / stages / split-K of a launch.  tile == 0 and no split: the measured table of this GPU decides, if it knows the shape.  An EXPLICIT choice
    (tile != 0, or a split) is the caller's and stays (tests and probes of split-K on tabled shapes must run split-K) -- unless the caller says its
    split is only a heuristic (`table_may_replace_split`: Lowering.conv's three-way split for want of tiles): where the table prefers the 8-wave loop
    (tiles 7 / 8 / 9: whole tiles or stream-K) the launch asks for that tile and KEEPS ksplit / ws: mi355x_gemm drops the split when the 8-wave loop
    takes the launch and runs the split on the 128 x 128 tile when it cannot (unaligned operands, 2 GB and more)."""
    if tile == 0 and ksplit <= 1:
        from .engine import tuning

        tile, stages = tuning.lookup(gemm_signature(a), stages)
    elif ksplit > 1 and table_may_replace_split:
        from .engine import tuning

        t8, _ = tuning.lookup(gemm_signature(a), 0)
        if t8 in (7, 8, 9, 10):
            tile = t8
    a.tile, a.ksplit, a.stages = tile, ksplit, stages
    if tile == 8 or os.environ.get("REFINERS_AMD_FORCE_TILE") == "8":
        sk = _streamk_current
        if sk is None:
            # eager call (tests, probes): one scratch per (device of the operands, current stream) -- launches on two streams of one device must not
            # share it, and the current device need not be the operands' (round-5 advisor)
            dev = operand.device if operand is not None and operand.device.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
            if dev.index is None:
                dev = torch.device("cuda", torch.cuda.current_device())
            key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
            sk = _streamk_eager.get(key)
            if sk is None:
                sk = _streamk_eager[key] = StreamK(dev)
        attach_streamk(a, sk)
    if ksplit > 1:
        assert ws is not None and ws.is_contiguous(), "split-K needs a scratch tensor of ksplit * M * N float32"
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel() * ws.element_size()
    else:
        a.ws, a.ws_bytes = None, 0


def _fill_epilogue(a: GemmArgs, out: Tensor, bias, rowbias, rows_per_group, res, geglu) -> None:
    assert out.dim() == 2 and out.stride(1) == 1
    a.out, a.ldo = out.data_ptr(), out.stride(0)
    a.bias = bias.data_ptr() if bias is not None else None
    if rowbias is not None:
        assert rowbias.dim() == 2 and rowbias.stride(1) == 1
        a.rowbias, a.ld_rowbias, a.rows_per_group = rowbias.data_ptr(), rowbias.stride(0), rows_per_group
    else:
        a.rowbias, a.ld_rowbias, a.rows_per_group = None, 0, 1
    if res is not None:
        assert res.dim() == 2 and res.stride(1) == 1
        a.res, a.ldres = res.data_ptr(), res.stride(0)
    else:
        a.res, a.ldres = None, 0
    a.geglu = int(geglu)  # 0 none, 1 GEGLU, 2 GELU (erf), 3 quick GELU


def attention(
    q: Tensor,
    out: Tensor,
    num_heads: int,
    streams: Sequence[tuple[Tensor, Tensor, int, float]],
    scale: Optional[float] = None,
) -> Tensor:
    """q, out: [B, Lq, H*D] views (last dim contiguous). streams: (k [B, Lk(+), H*D] view, vt [H*D, B, Lkp] view, Lk, out_scale)."""
    a = AttnArgs()
    B, Lq, HD = q.shape
    D = HD // num_heads
    a.dtype = dtype_code(q.dtype)
    a.B, a.H, a.D, a.Lq, a.nstream = B, num_heads, D, Lq, len(streams)
    assert q.stride(2) == 1 and out.stride(2) == 1
    a.q, a.ldq, a.q_batch_stride = q.data_ptr(), q.stride(1), q.stride(0)
    a.out, a.ldo, a.o_batch_stride = out.data_ptr(), out.stride(1), out.stride(0)
    a.scale = scale if scale is not None else D ** -0.5
    for s, (k, vt, Lk, osc) in enumerate(streams):
        assert k.dim() == 3 and vt.dim() == 3 and k.stride(2) == 1 and vt.stride(2) == 1
        kv = a.kv[s]
        kv.k, kv.ldk, kv.k_batch_stride = k.data_ptr(), k.stride(1), k.stride(0)
        kv.vt, kv.ldvt, kv.vt_batch_stride = vt.data_ptr(), vt.stride(0), vt.stride(1)
        kv.Lk, kv.out_scale = Lk, osc
    _launch("mi355x_attention", (C.byref(a),), "mi355x_attention")
    return out


def attention_general(
    q: Tensor,
    k: Tensor,
    vt: Tensor,
    out: Tensor,
    num_heads: int,
    Lk: int,
    scale: Optional[float] = None,
    causal: bool = False,
    out_scale: float = 1.0,
) -> Tensor:
    """Head shapes other than 64.  q [B, Lq, H*Dqk], k [B, Lk(+), H*Dqk], vt [H*Dv, B, Lkp] (Lkp = Lk rounded up to 64,
    finite padding), out [B, Lq, H*Dv] -- all views with a contiguous last dimension."""
    a = AttnGeneralArgs()
    B, Lq, HD = q.shape
    assert HD % num_heads == 0 and vt.shape[0] % num_heads == 0 and out.shape[2] == vt.shape[0]
    assert q.stride(2) == 1 and k.stride(2) == 1 and vt.stride(2) == 1 and out.stride(2) == 1
    a.dtype = dtype_code(q.dtype)
    a.B, a.H, a.Lq, a.Lk = B, num_heads, Lq, Lk
    a.Dqk, a.Dv, a.causal = HD // num_heads, vt.shape[0] // num_heads, int(causal)
    a.q, a.ldq, a.q_batch_stride = q.data_ptr(), q.stride(1), q.stride(0)
    a.k, a.ldk, a.k_batch_stride = k.data_ptr(), k.stride(1), k.stride(0)
    a.vt, a.ldvt, a.vt_batch_stride = vt.data_ptr(), vt.stride(0), vt.stride(1)
    a.out, a.ldo, a.o_batch_stride = out.data_ptr(), out.stride(1), out.stride(0)
    a.scale = scale if scale is not None else a.Dqk ** -0.5
    a.out_scale = out_scale
    _launch("mi355x_attention_general", (C.byref(a),), "mi355x_attention_general")
    return out


def layernorm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, out: Tensor) -> Tensor:
    a = LayerNormArgs()
    assert x.dim() == 2 and out.dim() == 2 and x.stride(1) == 1 and out.stride(1) == 1
    a.dtype = dtype_code(x.dtype)
    a.M, a.C = x.shape
    a.x, a.ldx, a.gamma, a.beta, a.eps = x.data_ptr(), x.stride(0), gamma.data_ptr(), beta.data_ptr(), eps
    a.out, a.ldo = out.data_ptr(), out.stride(0)
    _launch("mi355x_layernorm", (C.byref(a),), "mi355x_layernorm")
    return out


_gn_ws: "weakref.WeakValueDictionary[tuple[int, int, int], Tensor]" = weakref.WeakValueDictionary()


def groupnorm_nhwc(x: Tensor, gamma: Tensor, beta: Tensor, groups: int, eps: float, silu: bool, out: Tensor, colstats: Optional[Tensor] = None,
                   x2: Optional[Tensor] = None, colstats2: Optional[Tensor] = None) -> Tensor:
    """x, out: [B, HW, C] views with contiguous channels.  `colstats`: what the launch that produced x wrote through `colstats_out`
    ([B * HW / 32, C, 2] float32): the statistics pass over x is skipped.  `x2` [B, HW, C2]: the normalised tensor is the channel concatenation
    (x | x2), which never exists (out has C + C2 channels; `colstats2` then belongs to x2)."""
    a = GroupNormArgs()
    B, HW, C1 = x.shape
    Cc = C1 + (x2.shape[2] if x2 is not None else 0)
    assert x.stride(2) == 1 and out.stride(2) == 1 and x.stride(0) == HW * x.stride(1) and out.stride(0) == HW * out.stride(1) and out.shape[2] == Cc
    if x2 is not None:
        assert x2.shape[:2] == x.shape[:2] and x2.stride(2) == 1 and x2.stride(0) == HW * x2.stride(1) and x2.dtype == x.dtype and (colstats is None) == (colstats2 is None)
        a.x2, a.ldx2, a.C1 = x2.data_ptr(), x2.stride(1), C1
    need = load().mi355x_groupnorm_ws_floats(B, HW, Cc)
    # one scratch per (device, stream) for eager calls, one per PROGRAM while recording: two recorded programs may replay concurrently on two
    # streams (CompiledSDXL's split CFG pair, two trajectories in one process) whatever stream they were lowered on.  The table holds weak
    # references: a program's scratch lives in the keep-alive tuples of its launches and goes with them.
    if x.device.type == "cuda":
        dev = x.device.index if x.device.index is not None else torch.cuda.current_device()
        key = (dev, torch.cuda.current_stream().cuda_stream, id(_recorder) if _recorder is not None else 0)
    else:  # dry-run lowering on the meta / cpu device (tests): nothing is launched
        key = (-1, 0, 0)
    ws = _gn_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 20), dtype=torch.float32, device=x.device)
        _gn_ws[key] = ws
    a.dtype = dtype_code(x.dtype)
    a.B, a.HW, a.C, a.G = B, HW, Cc, groups
    a.x, a.ldx, a.gamma, a.beta, a.eps, a.silu = x.data_ptr(), x.stride(1), gamma.data_ptr(), beta.data_ptr(), eps, int(silu)
    a.out, a.ldo, a.ws = out.data_ptr(), out.stride(1), ws.data_ptr()
    if colstats is not None:
        assert HW % 32 == 0 and colstats.dtype == torch.float32 and colstats.is_contiguous() and colstats.numel() >= B * HW // 32 * C1 * 2
        a.colstats = colstats.data_ptr()
        if colstats2 is not None:
            assert colstats2.dtype == torch.float32 and colstats2.is_contiguous() and colstats2.numel() >= B * HW // 32 * (Cc - C1) * 2
            a.colstats2 = colstats2.data_ptr()
    _launch("mi355x_groupnorm", (C.byref(a),), "mi355x_groupnorm", keep=(ws, colstats, x2, colstats2))
    return out


def nchw_to_nhwc(x: Tensor, out: Tensor) -> Tensor:
    B, Cc, H, W = x.shape
    assert x.is_contiguous() and out.stride(-1) == 1
    _launch("mi355x_nchw_to_nhwc", (dtype_code(x.dtype), x.data_ptr(), out.data_ptr(), B, Cc, H * W, out.stride(-2),), "mi355x_nchw_to_nhwc")
    return out


def nhwc_to_nchw(x: Tensor, out: Tensor, C_: int) -> Tensor:
    """x: [B, HW, ld] rows; out: [B, C, H, W] contiguous."""
    B, HW = x.shape[0], x.shape[1]
    assert out.is_contiguous() and x.stride(-1) == 1
    _launch("mi355x_nhwc_to_nchw", (dtype_code(x.dtype), x.data_ptr(), out.data_ptr(), B, C_, HW, x.stride(1),), "mi355x_nhwc_to_nchw")
    return out


def im2col3x3_nchw(x: Tensor, out: Tensor) -> Tensor:
    B, Cc, H, W = x.shape
    assert x.is_contiguous() and out.dim() == 2 and out.is_contiguous()
    _launch("mi355x_im2col3x3_nchw", (dtype_code(x.dtype), x.data_ptr(), out.data_ptr(), B, Cc, H, W, out.shape[1],), "mi355x_im2col3x3_nchw")
    return out


def concat2(a_: Tensor, b_: Tensor, out: Tensor) -> Tensor:
    """Channel concat of two row-major 2-D tensors [M, C1] ++ [M, C2] -> out [M, C1+C2]."""
    M = a_.shape[0]
    _launch("mi355x_concat2", (dtype_code(a_.dtype), a_.data_ptr(), a_.stride(0), a_.shape[1], b_.data_ptr(), b_.stride(0), b_.shape[1],
                              out.data_ptr(), out.stride(0), M,), "mi355x_concat2")
    return out


def axpby(a_: Tensor, alpha: float, b_: Tensor, beta: float, out: Tensor) -> Tensor:
    assert a_.is_contiguous() and b_.is_contiguous() and out.is_contiguous() and a_.numel() == b_.numel() == out.numel()
    _launch("mi355x_axpby", (dtype_code(a_.dtype), a_.data_ptr(), alpha, b_.data_ptr(), beta, out.data_ptr(), a_.numel(),), "mi355x_axpby")
    return out


def softmax_rows(s: Tensor, out: Tensor, L: int, scale: float) -> Tensor:
    """out[m, :L] = softmax(scale * s[m, :L]), out[m, L:] = 0.  s: float32 [M, >= L] rows; out: [M, Lp] rows of the compute dtype."""
    assert s.dtype == torch.float32 and s.dim() == 2 and out.dim() == 2 and s.stride(1) == 1 and out.stride(1) == 1 and s.shape[0] == out.shape[0]
    assert s.shape[1] >= L and out.shape[1] >= L
    _launch("mi355x_softmax_rows", (dtype_code(out.dtype), s.data_ptr(), s.stride(0), out.data_ptr(), out.stride(0), s.shape[0], L, out.shape[1], scale), "mi355x_softmax_rows")
    return out


def colsum_rows(p: Tensor, acc: Tensor, accumulate: bool, scale: float) -> Tensor:
    """acc[j] (+)= scale * sum_i p[i, j]: p [M, L] rows of the compute dtype, acc float32 [L] (contiguous)."""
    assert p.dim() == 2 and p.stride(1) == 1 and acc.dtype == torch.float32 and acc.is_contiguous() and acc.numel() >= p.shape[1]
    _launch("mi355x_colsum_rows", (dtype_code(p.dtype), p.data_ptr(), p.stride(0), p.shape[0], p.shape[1], acc.data_ptr(), int(accumulate), scale), "mi355x_colsum_rows")
    return acc


def sag_degrade(x: Tensor, eps: Tensor, mass: Tensor, attn_hw: tuple[int, int], coef: Tensor, w1: Tensor, out: Tensor) -> Tensor:
    """Self-attention-guidance degraded latents (see the header): x, eps, out [n, C, h, w]; mass float32 [n, ah*aw]; coef device row; w1 float32 [k]."""
    n, Cc, h, w = x.shape
    assert x.is_contiguous() and eps.is_contiguous() and out.is_contiguous() and eps.shape == x.shape == out.shape and x.dtype == eps.dtype == out.dtype
    assert mass.dtype == torch.float32 and mass.is_contiguous() and mass.numel() == n * attn_hw[0] * attn_hw[1] and w1.dtype == torch.float32 and w1.is_contiguous()
    _launch("mi355x_sag_degrade", (dtype_code(x.dtype), x.data_ptr(), eps.data_ptr(), mass.data_ptr(), attn_hw[0], attn_hw[1], coef.data_ptr(), w1.data_ptr(), w1.numel(),
                                   out.data_ptr(), n, Cc, h, w), "mi355x_sag_degrade", keep=(mass, w1, coef))
    return out


def silu(x: Tensor, out: Tensor) -> Tensor:
    assert x.is_contiguous() and out.is_contiguous()
    _launch("mi355x_silu", (dtype_code(x.dtype), x.data_ptr(), out.data_ptr(), x.numel(),), "mi355x_silu")
    return out


def cfg_ddim_step(x: Tensor, unet_out: Tensor, coef: Tensor) -> Tensor:
    """In-place: x <- ddim(x, cfg(unet_out)). unet_out = [uncond; cond] (2*x.numel() elements); coef: f32[5] on device."""
    assert x.is_contiguous() and unet_out.is_contiguous() and unet_out.numel() == 2 * x.numel()
    assert coef.dtype == torch.float32 and coef.numel() >= 5 and coef.is_cuda
    _launch("mi355x_cfg_ddim_step", (dtype_code(x.dtype), x.data_ptr(), unet_out.data_ptr(), coef.data_ptr(), x.numel(),), "mi355x_cfg_ddim_step")
    return x


def cfg_linear_step(x: Tensor, unet_out: Tensor, hist: Tensor, model_in: Optional[Tensor], coef: Tensor) -> Tensor:
    """x, hist: [N, ...] latents (in place); unet_out: [2N, ...] = (unconditional, conditional); model_in: [2N, ...] or None;
    coef: 8 float32 on the device = (cfg, hx, he, kx, ke, kd, kp, s_next) -- see mi355x_cfg_linear_step in the header."""
    assert x.is_contiguous() and unet_out.is_contiguous() and hist.is_contiguous() and unet_out.numel() == 2 * x.numel() == 2 * hist.numel()
    assert coef.dtype == torch.float32 and coef.numel() >= 8 and (model_in is None or (model_in.is_contiguous() and model_in.numel() == 2 * x.numel()))
    _launch("mi355x_cfg_linear_step", (dtype_code(x.dtype), x.data_ptr(), unet_out.data_ptr(), hist.data_ptr(), model_in.data_ptr() if model_in is not None else None,
                                      coef.data_ptr(), x.numel()), "mi355x_cfg_linear_step")
    return x


def sinusoidal(x: Tensor, dim: int, out: Tensor, group: int = 1, col0: int = 0) -> Tensor:
    """x: float32 values (any shape, contiguous); out: 2-D [x.numel() / group, >= col0 + group * dim] rows of the compute dtype."""
    assert x.dtype == torch.float32 and x.is_contiguous() and out.dim() == 2 and out.stride(1) == 1
    assert x.numel() % group == 0 and out.shape[0] >= x.numel() // group and out.shape[1] >= col0 + group * dim
    _launch("mi355x_sinusoidal", (dtype_code(out.dtype), x.data_ptr(), x.numel(), dim, group, out.data_ptr(), out.stride(0), col0), "mi355x_sinusoidal")
    return out


def patchify_nchw(x: Tensor, patch: int, out: Tensor) -> Tensor:
    """x: [B, C, H, W] contiguous; out: [B * (H/P) * (W/P), >= C*P*P] rows."""
    B, Cc, H, W = x.shape
    assert x.is_contiguous() and out.dim() == 2 and out.stride(1) == 1 and out.shape[1] >= Cc * patch * patch
    _launch("mi355x_patchify_nchw", (dtype_code(x.dtype), x.data_ptr(), out.data_ptr(), B, Cc, H, W, patch, out.stride(0)), "mi355x_patchify_nchw")
    return out


def gather_rows(x: Tensor, idx: Tensor, out: Tensor) -> Tensor:
    """out[i] = x[idx[i]] (zero row where idx[i] < 0); x, out: 2-D with unit column stride; idx: int32 on the device."""
    assert x.dim() == 2 and out.dim() == 2 and x.stride(1) == 1 and out.stride(1) == 1 and x.shape[1] == out.shape[1]
    assert idx.dtype == torch.int32 and idx.is_contiguous() and idx.numel() == out.shape[0]
    _launch("mi355x_gather_rows", (dtype_code(x.dtype), x.data_ptr(), x.stride(0), idx.data_ptr(), out.data_ptr(), out.stride(0), out.shape[0], x.shape[1]),
            "mi355x_gather_rows", keep=(idx,))
    return out


def relpos_pack(src: Tensor, out: Tensor, heads: int, d: int, S1: int, S2: int, Lp: int, Dq: int) -> Tensor:
    """src [M, heads*Lp] -> out [M, heads*Dq]: the per-token windows of the folded relative-position products (see the header)."""
    assert src.dim() == 2 and out.dim() == 2 and src.stride(1) == 1 and out.stride(1) == 1 and src.shape[0] == out.shape[0]
    assert src.shape[1] >= heads * Lp and out.shape[1] >= heads * Dq
    _launch("mi355x_relpos_pack", (dtype_code(src.dtype), src.data_ptr(), src.stride(0), out.data_ptr(), out.stride(0), src.shape[0], heads, d, S1, S2, Lp, Dq),
            "mi355x_relpos_pack")
    return out


def pointwise_nchw(x: Tensor, w: Tensor, bias: Optional[Tensor], out: Tensor) -> Tensor:
    """1x1 conv of a few-channel NCHW image: x [B, Ci, H, W], w [Co, Ci], out [B, Co, H, W] (all contiguous)."""
    B, Ci, H, W = x.shape
    assert x.is_contiguous() and out.is_contiguous() and w.is_contiguous() and tuple(w.shape) == (out.shape[1], Ci)
    _launch("mi355x_pointwise_nchw", (dtype_code(x.dtype), x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(),
                                      B, Ci, out.shape[1], H * W), "mi355x_pointwise_nchw")
    return out


def set_glds(enabled: bool) -> None:
    """A/B switch: global_load_lds staging (default) vs register staging, for the GEMM and attention tile loaders."""
    lib = load()
    lib.mi355x_set_option(b"glds", int(enabled))
    lib.mi355x_attention_set_glds(int(enabled))
