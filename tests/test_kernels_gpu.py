"""pytest -m gpu: every C-ABI kernel vs a plain PyTorch fp32 reference of the same op (see kernel_cases.py)."""
import pytest
import torch

from tests import kernel_cases

pytestmark = pytest.mark.gpu

CASES = kernel_cases.all_cases()


@pytest.fixture(scope="module", autouse=True)
def _require_native(gpu_device):
    from refiners_amd import native

    native.load()  # fails loudly if the HIP library is missing: there is no fallback to test


@pytest.mark.parametrize("name", [n for n, _ in CASES])
def test_kernel_parity(name):
    thunk = dict(CASES)[name]
    err, scale, tol = thunk()
    torch.cuda.synchronize()
    assert err <= tol * scale + 1e-7, f"{name}: max|err|={err:.3e} vs ref max {scale:.3e} (tol {tol:g} relative)"


def test_bit_reproducible():
    """Two runs of the same launch are bit-identical (reference: test_sd15_unet.py:21-37 torch.equal contract)."""
    e1 = kernel_cases.attention_case(2, 4, 1024, 1024, torch.bfloat16)
    e2 = kernel_cases.attention_case(2, 4, 1024, 1024, torch.bfloat16)
    assert e1[0] == e2[0]
    g1 = kernel_cases.groupnorm_case(2, 320, 4096, torch.float32)
    g2 = kernel_cases.groupnorm_case(2, 320, 4096, torch.float32)
    assert g1[0] == g2[0]
