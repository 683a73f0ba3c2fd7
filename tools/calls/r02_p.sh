#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "groupnorm or gn_ or reproducible" 2>&1 | tail -6 ) > $O/r02_p_kernel_tests.log
tail -3 $O/r02_p_kernel_tests.log
( timeout 300 python tools/probe_gn.py 2>&1 | grep -v amdgpu.ids ) > $O/r02_p_probe_gn.log
cat $O/r02_p_probe_gn.log
