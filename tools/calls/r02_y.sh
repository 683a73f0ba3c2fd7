#!/bin/bash
# round 2, GPU call Y: the experimental q-projection + cross-attention fusion through one golden engine test (float32, LoRA + IP-Adapter)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( REFINERS_AMD_XATTN_FUSE=1 timeout 40 python -m pytest tests/test_engine_gpu.py -q -x -s -k "test_merged_lora_mode_matches_reference and sdxl_lora_ip" 2>&1 | grep -v amdgpu.ids | tail -4 ) > $O/r02_y_xattn_engine.log
cat $O/r02_y_xattn_engine.log
