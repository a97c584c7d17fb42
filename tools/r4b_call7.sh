#!/bin/bash
mkdir -p gpurun_out/r4b7; O=gpurun_out/r4b7
timeout 900 python -m pytest tests/test_model_parity.py tests/test_hip_ops.py -q -x -m gpu -k "bf16 or linear" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
bash tools/ab_multi.sh 2 "EGV_LIB_PATH=$PWD/tools/exp_libs/libegovlp_hip_head2.so" "EGV_DUMMY=1" "EGV_WGRAD_CUS_FUSED=96" "EGV_WGRAD_CUS_FUSED=81" "EGV_PP_LIMIT_SLACK=0" "EGV_PP_LIMIT_SLACK_FUSED=16" 2>&1 | tee $O/ab.log
