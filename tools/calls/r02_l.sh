#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 300 python tools/probe_attn_pipe.py 2>&1 | grep -v amdgpu.ids ) > $O/r02_l_probe_attn.log
cat $O/r02_l_probe_attn.log
