#!/bin/bash
# One experimental build of ONE translation unit with extra compiler flags, linked with the product objects of the others:
#   tools/variant_build.sh <name> <unit, e.g. egv_norm> "<flags>"  ->  tools/exp_libs/libegovlp_hip_<name>.so
# A/B against the product library with EGV_LIB_PATH.
set -e
cd "$(dirname "$0")/../egovlpv2_amd/csrc"
name=$1; unit=$2; shift; shift
mkdir -p ../../tools/exp_libs
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-value"
case $unit in egv_attn_mfma|egv_attn_time|egv_attn_space|egv_attn_cross) FLAGS="$FLAGS -mllvm -amdgpu-mfma-vgpr-form";; esac
hipcc $FLAGS "$@" -c $unit.hip -o build/${unit}_$name.o
objs=""
for o in egv_gemm egv_gemm2 egv_gemm3 egv_gemm4 egv_gemm5 egv_mx egv_norm egv_attn egv_attn_mfma egv_attn_time egv_attn_space egv_attn_cross egv_misc egv_optim egv_api egv_block; do
  if [ $o = $unit ]; then objs="$objs build/${unit}_$name.o"; else objs="$objs build/$o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp_libs/libegovlp_hip_$name.so $objs
echo built $name
