#!/bin/bash
# round 4, session 3, GPU call 1: grouped weight gradients -- matrix-pipe bias sums + contiguous units vs HEAD, ablations, step A/B
mkdir -p gpurun_out/r4b1; O=gpurun_out/r4b1
timeout 600 python -m pytest tests/test_hip_ops.py -q -x -m gpu -k "grouped_weight or wgrad or linear_forms or linear_large" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
for v in product head w4nocontig w4exp1 w4exp2 w4exp3; do
  if [ $v = product ]; then timeout 300 python tools/wgrad_exp2.py product; else EGV_LIB_PATH=$PWD/tools/exp_libs/libegovlp_hip_$v.so timeout 300 python tools/wgrad_exp2.py $v; fi
done 2>&1 | grep -v Warning | tee $O/wgrad_exp2.log
bash tools/ab_multi.sh 3 "EGV_LIB_PATH=$PWD/tools/exp_libs/libegovlp_hip_head.so" "EGV_DUMMY=1" "EGV_LIB_PATH=$PWD/tools/exp_libs/libegovlp_hip_w4nocontig.so" 2>&1 | tee $O/ab.log
