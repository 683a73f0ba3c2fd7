"""Build libmi355x_refiners.so (hand-written gfx950 kernels + C ABI) in-tree with hipcc.

The library is built next to its sources (refiners_amd/csrc/) so that it travels with the repository snapshot to
the GPU box; nothing is installed into site-packages and nothing is JIT-cached under ~/.cache.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB = CSRC / "libmi355x_refiners.so"
SOURCES = ["gemm.hip", "gemm_conv.hip", "gemm8.hip", "attention.hip", "attention_general.hip", "norm.hip", "elementwise.hip"]
HEADERS = ["common.cuh", "gemm_params.cuh", "gemm_epilogue.cuh", "gemm_kernel.cuh", "gemm8_kernel.cuh", "gemm_lora_producer.cuh", "../../include/mi355x_refiners.h"]
ARCH = "gfx950"
#: headers each translation unit includes (an incremental build -- `python -m refiners_amd.build_native` without --force -- recompiles a source only when it or one of these changed)
DEPS = {
    "gemm.hip": ["common.cuh", "gemm_params.cuh", "gemm_epilogue.cuh", "gemm_kernel.cuh", "gemm8_kernel.cuh", "gemm_lora_producer.cuh"],
    "gemm_conv.hip": ["common.cuh", "gemm_params.cuh", "gemm_epilogue.cuh", "gemm_kernel.cuh", "gemm_lora_producer.cuh"],
    "gemm8.hip": ["common.cuh", "gemm_params.cuh", "gemm_epilogue.cuh", "gemm8_kernel.cuh", "gemm_lora_producer.cuh"],
    "attention.hip": ["common.cuh"], "attention_general.hip": ["common.cuh"], "norm.hip": ["common.cuh"], "elementwise.hip": ["common.cuh"],
}


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or add /opt/rocm/bin to PATH)")


def is_stale() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + [(CSRC / h).resolve() for h in HEADERS] + [Path(__file__)]
    return any(d.stat().st_mtime > t for d in deps)


def build_variant(tag: str, defines: list[str]) -> Path:
    """An experiment build of the same sources with extra -D switches -> csrc/variants/libmi355x_refiners_<tag>.so (git-ignored, travels with
    the snapshot).  tools/ab_step.py runs it beside the product library IN ONE PROCESS (variant `name%<tag>=...`): a recorded program keeps the
    entry points of the library it was lowered with."""
    out = CSRC / "variants" / f"libmi355x_refiners_{tag}.so"
    return build_native(force=True, defines=defines, lib=out, objdir=CSRC / "variants" / tag)


def build_native(force: bool = False, verbose: bool = False, defines: list[str] | None = None, lib: Path = LIB, objdir: Path = CSRC) -> Path:
    """Compile every HIP source for gfx950 and link the shared library. Returns the library path."""
    if not force and not is_stale():
        return lib
    LIB = lib  # noqa: N806 -- (shadows the module constant for an experiment build)
    hipcc = hipcc_path()
    objdir.mkdir(parents=True, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = objdir / (src.replace(".hip", ".o"))
        if not force and not defines and obj.exists():  # incremental: the object is newer than its source and every header it includes
            deps = [CSRC / src, (CSRC / "../../include/mi355x_refiners.h").resolve(), Path(__file__)] + [CSRC / h for h in DEPS[src]]
            if all(d.stat().st_mtime <= obj.stat().st_mtime for d in deps):
                objs.append(obj)
                continue
        # -amdgpu-mfma-vgpr-form: MFMA accumulators live in VGPRs (gfx950's register file is unified), which removes the
        # v_accvgpr_read/write traffic between the softmax / epilogue VALU code and the matrix cores (attention inner
        # loop: 1062 -> 876 instructions) and lowers the total register count of every kernel.
        # -pragma-unroll-threshold (gemm8.hip only): the shared tile epilogue's row loop must unroll fully over the 8 x 4 accumulator blocks of the
        # 256 x 256 tile -- a partly unrolled loop indexes the accumulators at run time, i.e. puts all 128 of them into scratch memory
        extra = ["-mllvm", "-pragma-unroll-threshold=100000"] if src == "gemm8.hip" else []
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1", *extra, *[f"-D{d}" for d in (defines or [])],
               "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out:
            print(out, file=sys.stderr)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB)] + [str(o) for o in objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:  # python -m refiners_amd.build_native --variant prio MI355X_GEMM_PRIO=1
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2 :]))
    else:
        print(build_native(force="--force" in sys.argv, verbose=True))
