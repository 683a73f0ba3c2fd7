#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 120 python -m pytest tests/test_kernels_gpu.py -q -x -k "xattn" 2>&1 | tail -12 ) > $O/r02_w_kernel_tests.log
tail -6 $O/r02_w_kernel_tests.log
( timeout 100 python tools/probe_xattn.py 2>&1 | grep -v amdgpu.ids ) > $O/r02_w_probe_xattn.log
cat $O/r02_w_probe_xattn.log
