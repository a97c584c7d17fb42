#!/bin/bash
# instrumentation build (per-K-tile stamps: -DEGV_INSTRUMENT on egv_gemm2 / egv_gemm3) of a variant of egv_gemm3:
#   tools/instr_variant.sh <name> "<flags>"  ->  tools/exp_libs/libegovlp_hip_instr_<name>.so   (tools/gemm_pp_stamps.py with EGV_LIB_PATH)
set -e
cd "$(dirname "$0")/../egovlpv2_amd/csrc"
name=$1; shift
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-value -DEGV_INSTRUMENT"
[ -f build/egv_gemm2_instr.o ] || hipcc $F -c egv_gemm2.hip -o build/egv_gemm2_instr.o
hipcc $F "$@" -c egv_gemm3.hip -o build/egv_gemm3_instr_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp_libs/libegovlp_hip_instr_$name.so build/egv_gemm.o build/egv_gemm2_instr.o build/egv_gemm3_instr_$name.o build/egv_gemm4.o build/egv_gemm5.o build/egv_mx.o build/egv_norm.o build/egv_attn.o build/egv_attn_mfma.o build/egv_attn_time.o build/egv_attn_space.o build/egv_attn_cross.o build/egv_misc.o build/egv_optim.o build/egv_api.o build/egv_block.o
echo built instr_$name
