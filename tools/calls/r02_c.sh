#!/bin/bash
# round 2, GPU call C: in-launch LoRA, ControlNet lowering, GN probe, full default bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 600 python -m pytest tests/test_kernels_gpu.py -q 2>&1 | tail -15 ) > $O/r02_c_kernel_tests.log
tail -3 $O/r02_c_kernel_tests.log
( timeout 900 python -m pytest tests/test_engine_gpu.py -q -k "float32_matches_reference or bfloat16_close or merged_lora_mode_bfloat16 or sd1_controlnet or adapters_stay_live" -s 2>&1 | grep -v "^$" | tail -30 ) > $O/r02_c_engine_tests.log
tail -4 $O/r02_c_engine_tests.log
( timeout 300 python tools/probe_gn.py 2>&1 | grep -v amdgpu.ids ) > $O/r02_c_probe_gn.log
cat $O/r02_c_probe_gn.log
( time timeout 1200 python bench.py ) > $O/r02_c_bench_default.json 2> $O/r02_c_bench_default.err
tail -4 $O/r02_c_bench_default.err
cut -c1-1500 $O/r02_c_bench_default.json
