#!/bin/bash
# round 2, GPU call S: rocprofv3 kernel trace + PMC passes of the final build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 420 python tools/profile_round.py --tag r02_s 2>&1 | grep -v amdgpu.ids ) > $O/r02_s_profile.log
tail -8 $O/r02_s_profile.log | cut -c1-600
du -sh $O
