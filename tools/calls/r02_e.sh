#!/bin/bash
# round 2, GPU call E: full GPU suite, re-tune (merged, fused), profile the tuned build, default bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $O/r02_e_gpu_tests.log
tail -4 $O/r02_e_gpu_tests.log
rm -f refiners_amd/engine/tuning_gfx950.json
( timeout 500 python tools/autotune.py --budget-s 300 2>&1 | grep -v amdgpu.ids ) > $O/r02_e_autotune_merged.log
tail -1 $O/r02_e_autotune_merged.log
( timeout 500 python tools/autotune.py --lora-mode fused --merge --budget-s 300 2>&1 | grep -v amdgpu.ids ) > $O/r02_e_autotune_fused.log
tail -1 $O/r02_e_autotune_fused.log
cp refiners_amd/engine/tuning_gfx950.json $O/r02_e_tuning_gfx950.json
( timeout 900 python tools/profile_round.py --tag r02_e 2>&1 | grep -v amdgpu.ids ) > $O/r02_e_profile.log
tail -12 $O/r02_e_profile.log | cut -c1-400
( time timeout 1200 python bench.py ) > $O/r02_e_bench_default.json 2> $O/r02_e_bench_default.err
tail -3 $O/r02_e_bench_default.err
cut -c1-600 $O/r02_e_bench_default.json
du -sh $O
