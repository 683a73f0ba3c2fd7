"""Only in the build container (skipped where /root/reference is absent): run the reference's OWN fluxion / adapter unit
tests, unmodified, against the refiners_amd mirror (module aliases in tests/refcompat/alias_conftest.py).  This is the API
regression suite SURVEY.md section 8(c) lists: 50 tests."""
import os
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

REF_TESTS = Path("/root/reference/tests")
ROOT = Path(__file__).resolve().parent.parent
FILES = ["fluxion/layers/test_chain.py", "fluxion/layers/test_basics.py", "fluxion/test_module.py", "adapters/test_adapter.py",
         "adapters/test_adapter_context.py", "adapters/test_lora.py", "adapters/test_range_adapter.py"]
pytestmark = pytest.mark.skipif(not REF_TESTS.exists(), reason="the reference checkout is only present in the build container")


def test_reference_unit_tests_pass_on_the_mirror(tmp_path):
    shutil.copy(ROOT / "tests" / "refcompat" / "alias_conftest.py", tmp_path / "conftest.py")
    for f in FILES:
        shutil.copy(REF_TESTS / f, tmp_path / Path(f).name.replace("test_", "test_ref_", 1))
    env = dict(os.environ, REFINERS_AMD_ROOT=str(ROOT))
    r = subprocess.run([sys.executable, "-m", "pytest", "-p", "no:cacheprovider", "-q", str(tmp_path)], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:]
    assert r.returncode == 0, r.stdout[-3000:]
    assert "50 passed" in tail, tail
