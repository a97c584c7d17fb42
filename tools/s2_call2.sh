#!/bin/bash
O=gpurun_out/s2c2; mkdir -p $O
python bench.py --steps 10 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
for r in 1 2 3 4; do for v in 5 2; do
  echo "== smd$v round $r"; EGV_LIB_PATH=tools/exp_libs/libegovlp_hip_smd$v.so RUNS=6 REC=0 timeout 300 python tools/repro_check.py 2>&1 | grep -E "vs 0|differ" | grep -v "0 of 213" | cut -c1-300
done; done 2>&1 | tee $O/stale_diag2.txt
