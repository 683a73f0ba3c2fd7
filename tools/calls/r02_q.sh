#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 300 python -m pytest tests/test_kernels_gpu.py -q 2>&1 | tail -6 ) > $O/r02_q_kernel_tests.log
tail -3 $O/r02_q_kernel_tests.log
( timeout 500 python tools/ab_step.py --workload lora_ip --steps 20 --rounds 3 \
    gnfused=REFINERS_AMD_GN_FUSED:1 \
    all= 2>&1 | grep -v amdgpu.ids ) > $O/r02_q_ab.log
grep "ms/step\|launches" $O/r02_q_ab.log
tail -1 $O/r02_q_ab.log | cut -c1-1200
