"""Exit 0 iff the 8-wave loop (tile 7) runs a hot 4096^3 bf16 GEMM above 1100 TFLOP/s: guards GPU minutes of tuning runs against a slow build."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402

x = torch.randn(4096, 4096, device="cuda").bfloat16()
w = native.KBlocked((torch.randn(4096, 4096, device="cuda") / 64).bfloat16())
o = torch.empty(4096, 4096, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    native.gemm([(x, w)], o, tile=7)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    native.gemm([(x, w)], o, tile=7)
b.record()
torch.cuda.synchronize()
tf = 2 * 4096 ** 3 / (a.elapsed_time(b) / 20 * 1e-3) / 1e12
print(f"tile 7, 4096^3: {tf:.0f} TFLOP/s")
sys.exit(0 if tf > 1100 else 1)
