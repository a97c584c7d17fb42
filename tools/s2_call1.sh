#!/bin/bash
O=gpurun_out/s2c1; mkdir -p $O
python bench.py --steps 10 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
timeout 600 python -m pytest tests/test_hip_ops.py -q -x -m gpu -k "persistent_gemm or mlp or gelu or linear" > $O/tests.log 2>&1; tail -2 $O/tests.log
G0=tools/exp_libs/libegovlp_hip_gelu0.so
for i in 1 2; do
  echo "product:"; python tools/gemm_gelu_bench.py 2>&1 | tail -3
  echo "gelu0:"; EGV_LIB_PATH=$G0 python tools/gemm_gelu_bench.py 2>&1 | tail -3
done 2>&1 | tee $O/gelu_bench.txt
bash tools/ab_multi.sh 3 "EGV_NOP=0" "EGV_LIB_PATH=$G0" 2>&1 | tee $O/ab_gelu.txt
for r in 1 2 3; do for v in 2 3 4; do
  echo "== smd$v round $r"; EGV_LIB_PATH=tools/exp_libs/libegovlp_hip_smd$v.so RUNS=6 REC=0 timeout 300 python tools/repro_check.py 2>&1 | grep -E "vs 0|differ" | cut -c1-300
done; done 2>&1 | tee $O/stale_diag.txt
