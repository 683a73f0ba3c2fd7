#!/bin/bash
# An experiment build of attention.hip alone, linked with the product objects of the other sources (seconds instead of the minutes of a whole variant build):
#   tools/build_attn_variant.sh <tag> [MACRO=value ...]  ->  refiners_amd/csrc/variants/libmi355x_refiners_<tag>.so   (use with REFINERS_AMD_LIB=...)
set -e
cd "$(dirname "$0")/../refiners_amd/csrc"
tag=$1; shift
defs=""; for d in "$@"; do defs="$defs -D$d"; done
mkdir -p variants/$tag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 $defs -c attention.hip -o variants/$tag/attention.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libmi355x_refiners_$tag.so gemm.o gemm_conv.o gemm8.o variants/$tag/attention.o attention_general.o norm.o elementwise.o
echo variants/libmi355x_refiners_$tag.so
