#!/bin/bash
# round 2, GPU call B: LN stats fix, fused GroupNorm, wide-head attention, LCM; parity + A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 600 python -m pytest tests/test_kernels_gpu.py -q 2>&1 | tail -15 ) > $O/r02_b_kernel_tests.log
tail -3 $O/r02_b_kernel_tests.log
( timeout 900 python -m pytest tests/test_engine_gpu.py -q -k "float32_matches_reference or bfloat16_close or vae_decoder or lcm or other_solvers or sd1_float32" -s 2>&1 | grep -v "^$" | tail -30 ) > $O/r02_b_engine_tests.log
tail -4 $O/r02_b_engine_tests.log
( timeout 600 python tools/ab_step.py --workload lora_ip base=REFINERS_AMD_LN_FUSE:0,REFINERS_AMD_GN_FUSED:0 gn=REFINERS_AMD_LN_FUSE:0 ln=REFINERS_AMD_GN_FUSED:0 all= 2>&1 | grep -v amdgpu.ids ) > $O/r02_b_ab.log
tail -8 $O/r02_b_ab.log
