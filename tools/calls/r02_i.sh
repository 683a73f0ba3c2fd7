#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "tile7 or tile8 or tiles_7" 2>&1 | tail -6 ) > $O/r02_i_kernel_tests.log
tail -2 $O/r02_i_kernel_tests.log
( timeout 200 python tools/probe_tile7.py 2>&1 | grep -v amdgpu.ids ) > $O/r02_i_probe_tiles.log
cat $O/r02_i_probe_tiles.log
