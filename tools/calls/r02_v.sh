#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 120 python -m pytest tests/test_kernels_gpu.py -q -k "qkv" 2>&1 | tail -3 ) > $O/r02_v_kernel_tests.log
tail -1 $O/r02_v_kernel_tests.log
( timeout 300 python tools/ab_step.py --workload lora_ip --steps 20 --rounds 3 \
    base= pad64=REFINERS_AMD_VT_PAD:64 pad192=REFINERS_AMD_VT_PAD:192 2>&1 | grep -v amdgpu.ids ) > $O/r02_v_ab.log
grep "ms/step" $O/r02_v_ab.log
