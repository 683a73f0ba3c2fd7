#!/bin/bash
# round 2, GPU call D: new kernel cases (LN+LoRA), re-tune (merged, fused), profile the tuned build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 600 python -m pytest tests/test_kernels_gpu.py -q 2>&1 | tail -8 ) > $O/r02_d_kernel_tests.log
tail -3 $O/r02_d_kernel_tests.log
( timeout 600 python -m pytest tests/test_engine_gpu.py -q -k "float32_matches_reference and lora_ip" -s 2>&1 | grep -v "^$" | tail -6 ) > $O/r02_d_engine_tests.log
tail -3 $O/r02_d_engine_tests.log
rm -f refiners_amd/engine/tuning_gfx950.json
( timeout 500 python tools/autotune.py --budget-s 300 2>&1 | grep -v amdgpu.ids ) > $O/r02_d_autotune_merged.log
tail -2 $O/r02_d_autotune_merged.log
( timeout 500 python tools/autotune.py --lora-mode fused --merge --budget-s 300 2>&1 | grep -v amdgpu.ids ) > $O/r02_d_autotune_fused.log
tail -2 $O/r02_d_autotune_fused.log
( timeout 300 python tools/ab_step.py --workload lora_ip --lora-mode fused inlaunch= 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-600 ) > $O/r02_d_fused_ab.log
cat $O/r02_d_fused_ab.log
( timeout 900 python tools/profile_round.py --tag r02_d 2>&1 | grep -v amdgpu.ids ) > $O/r02_d_profile.log
tail -40 $O/r02_d_profile.log
