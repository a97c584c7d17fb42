#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5s3; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "gemm or linear or mlp" > $O/test_gemm.txt 2>&1; echo "gemm tests rc $?"; tail -4 $O/test_gemm.txt
for v in prod bulk xb4 xb1 prod bulk; do
  if [ $v = prod ]; then timeout 300 python tools/pp_exp.py prod 2>/dev/null | tail -1; else EGV_LIB_PATH=$R/tools/exp_libs/libegovlp_hip_$v.so timeout 300 python tools/pp_exp.py $v 2>/dev/null | tail -1; fi
done
bash tools/ab_multi.sh 2 "EGV_X=0" "EGV_LIB_PATH=$R/tools/exp_libs/libegovlp_hip_bulk.so"
