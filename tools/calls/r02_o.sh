#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 500 python tools/ab_step.py --workload lora_ip --steps 20 --rounds 3 \
    base=REFINERS_AMD_TIME_BATCH:0,REFINERS_AMD_ATTN_PIPE:1/0/0/1 \
    tb=REFINERS_AMD_ATTN_PIPE:1/0/0/1 \
    xcd=REFINERS_AMD_ATTN_PIPE:1/1/0/1 \
    pl=REFINERS_AMD_ATTN_PIPE:1/1/1/1 \
    all= 2>&1 | grep -v amdgpu.ids ) > $O/r02_o_ab.log
grep "ms/step\|launches" $O/r02_o_ab.log
tail -1 $O/r02_o_ab.log | cut -c1-1500
