#!/bin/bash
# round 2, GPU call G: staggered 8-wave tile (tile 7): parity, re-tune with it as a candidate, engine parity on the new table, bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 600 python -m pytest tests/test_kernels_gpu.py -q 2>&1 | tail -12 ) > $O/r02_g_kernel_tests.log
tail -3 $O/r02_g_kernel_tests.log
( timeout 200 python tools/probe_tile7.py 2>&1 | grep -v amdgpu.ids ) > $O/r02_g_probe_tile7.log
cat $O/r02_g_probe_tile7.log
cp refiners_amd/engine/tuning_gfx950.json $O/r02_g_table_before.json
( timeout 500 python tools/autotune.py --merge --budget-s 320 2>&1 | grep -v amdgpu.ids ) > $O/r02_g_autotune_merged.log
tail -1 $O/r02_g_autotune_merged.log
grep -c '"kept": \[7' $O/r02_g_autotune_merged.log
cp refiners_amd/engine/tuning_gfx950.json $O/r02_g_tuning_gfx950.json
( timeout 600 python -m pytest tests/test_engine_gpu.py -q -k "float32_matches_reference or bfloat16_close or full_size_step" -s 2>&1 | grep -v "^$" | tail -14 ) > $O/r02_g_engine_tests.log
tail -3 $O/r02_g_engine_tests.log
( timeout 300 python tools/ab_step.py --workload lora_ip tuned= 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-900 ) > $O/r02_g_ab.log
cat $O/r02_g_ab.log
