#!/bin/bash
# round 2, GPU call A: new GEMM core (rotated loop, K groups, transposed groups, LN fusion) -- parity, tuning, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x 2>&1 | tail -25 ) > $O/r02_a_kernel_tests.log
tail -3 $O/r02_a_kernel_tests.log
( timeout 900 python -m pytest tests/test_engine_gpu.py -q -k "float32_matches_reference or bfloat16_close or merged_lora or two_trajectories or recycled or in_place or merged_weights or sd1_float32" -s 2>&1 | grep -v "^$" | tail -40 ) > $O/r02_a_engine_tests.log
tail -5 $O/r02_a_engine_tests.log
( timeout 600 python tools/autotune.py --budget-s 330 2>&1 | grep -v amdgpu.ids ) > $O/r02_a_autotune.log
tail -3 $O/r02_a_autotune.log
( timeout 600 python tools/ab_step.py --workload lora_ip base=REFINERS_AMD_LN_FUSE:0,REFINERS_AMD_QKV_MERGE:0,REFINERS_AMD_TUNING:0 qkv=REFINERS_AMD_LN_FUSE:0,REFINERS_AMD_TUNING:0 ln_qkv=REFINERS_AMD_TUNING:0 all= 2>&1 | grep -v amdgpu.ids ) > $O/r02_a_ab.log
tail -8 $O/r02_a_ab.log
