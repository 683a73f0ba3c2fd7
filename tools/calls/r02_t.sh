#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "attn" 2>&1 | tail -8 ) > $O/r02_t_kernel_tests.log
tail -4 $O/r02_t_kernel_tests.log
( timeout 300 python tools/probe_attn_pipe.py 2>&1 | grep -v amdgpu.ids ) > $O/r02_t_probe_attn.log
cat $O/r02_t_probe_attn.log
