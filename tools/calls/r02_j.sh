#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "attn" 2>&1 | tail -6 ) > $O/r02_j_kernel_tests.log
tail -2 $O/r02_j_kernel_tests.log
( timeout 200 python tools/probe_attn_pipe.py 2>&1 | grep -v amdgpu.ids ) > $O/r02_j_probe_attn.log
cat $O/r02_j_probe_attn.log
( timeout 400 python tools/ab_step.py --workload lora_ip --steps 20 --rounds 3 \
    base=REFINERS_AMD_TIME_BATCH:0,REFINERS_AMD_ATTN_PIPE:1/0 \
    attn=REFINERS_AMD_TIME_BATCH:0 \
    xcd0=REFINERS_AMD_ATTN_PIPE:2/0 \
    all= 2>&1 | grep -v amdgpu.ids ) > $O/r02_j_ab.log
grep "ms/step\|launches" $O/r02_j_ab.log
