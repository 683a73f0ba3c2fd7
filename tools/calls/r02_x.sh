#!/bin/bash
# round 2, GPU call X: HEAD sanity after the ABI-4 build: every kernel parity case, then smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 100 python -m pytest tests/test_kernels_gpu.py -q 2>&1 | tail -3 ) > $O/r02_x_kernel_tests.log
tail -1 $O/r02_x_kernel_tests.log
( timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 ) > $O/r02_x_smoke.log
cat $O/r02_x_smoke.log
