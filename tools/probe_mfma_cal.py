"""Calibration launch set for the MFMA-busy counter: ten 4096^3 bf16 GEMMs on the 128 x 128 tile (8 388 608 v_mfma_f32_16x16x32_bf16
per launch, known exactly), run under `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` by tools/profile_round.py to find what
one counted unit means on this profiler build (which SIMDs / XCDs the value aggregates)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402

native.load()
x = torch.randn(4096, 4096, device="cuda").bfloat16()
w = native.KBlocked((torch.randn(4096, 4096, device="cuda") / 64).bfloat16())
o = torch.empty(4096, 4096, device="cuda", dtype=torch.bfloat16)
for _ in range(10):
    native.gemm([(x, w)], o, tile=1, stages=2)
torch.cuda.synchronize()
