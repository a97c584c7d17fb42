#!/bin/bash
# Experimental builds of the persistent GEMM (EGV_PP_EXP=1: no epilogue stores, 2: stores trickled over the plain K-tiles):
# gpurun_exp/libegovlp_hip_exp{1,2}.so next to the product library (same objects, only egv_gemm3.o differs).
set -e
cd "$(dirname "$0")/../egovlpv2_amd/csrc"
bash build.sh
mkdir -p ../../tools/exp_libs
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-value"
for v in "$@"; do
  hipcc $FLAGS -DEGV_PP_EXP=$v -c egv_gemm3.hip -o build/egv_gemm3_exp$v.o &
done
wait
for v in "$@"; do
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp_libs/libegovlp_hip_exp$v.so build/egv_gemm.o build/egv_gemm2.o build/egv_gemm3_exp$v.o build/egv_gemm4.o build/egv_gemm5.o build/egv_mx.o build/egv_norm.o build/egv_attn.o build/egv_attn_mfma.o build/egv_attn_time.o build/egv_attn_space.o build/egv_misc.o build/egv_optim.o build/egv_api.o build/egv_block.o
done
echo done
