#!/bin/bash
# Build libegovlp_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../libegovlp_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-value"
# EGV_INSTRUMENT=1 bash build.sh: instrumentation build (per-K-tile cycle stamps of the persistent GEMM, egv_debug_timing); rebuilds everything
if [ -n "$EGV_INSTRUMENT" ]; then FLAGS="$FLAGS -DEGV_INSTRUMENT"; rm -f build/*.o; fi
mkdir -p build
pids=()
for f in egv_gemm.hip egv_gemm2.hip egv_gemm3.hip egv_gemm4.hip egv_gemm5.hip egv_mx.hip egv_norm.hip egv_attn.hip egv_attn_mfma.hip egv_attn_time.hip egv_attn_space.hip egv_attn_cross.hip egv_misc.hip egv_optim.hip; do
  o=build/${f%.hip}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ egv_common.h -nt "$o" ] || [ egv_attn.h -nt "$o" ] || [ egv_gemm.h -nt "$o" ] || [ egv_wgrad_core.h -nt "$o" ] || [ ../../include/egovlp_hip.h -nt "$o" ]; then
    # attention: keep MFMA accumulators in VGPRs (they feed the softmax VALU code directly; the default AGPR form costs an
    # accvgpr copy per accumulator register, ~15-20 % of the loop's instructions)
    EXTRA=""
    if [ "$f" = "egv_attn_mfma.hip" ] || [ "$f" = "egv_attn_time.hip" ] || [ "$f" = "egv_attn_space.hip" ] || [ "$f" = "egv_attn_cross.hip" ]; then EXTRA="-mllvm -amdgpu-mfma-vgpr-form"; fi
    hipcc $FLAGS $EXTRA -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for f in egv_api.cpp egv_block.cpp; do
  o=build/${f%.cpp}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ ../../include/egovlp_hip.h -nt "$o" ]; then
    hipcc $FLAGS -x hip -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT build/egv_gemm.o build/egv_gemm2.o build/egv_gemm3.o build/egv_gemm4.o build/egv_gemm5.o build/egv_mx.o build/egv_norm.o build/egv_attn.o build/egv_attn_mfma.o build/egv_attn_time.o build/egv_attn_space.o build/egv_attn_cross.o build/egv_misc.o build/egv_optim.o build/egv_api.o build/egv_block.o
echo "built $OUT"
