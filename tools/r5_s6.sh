#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5s6; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_hip_ops.py -x -q -k "gemm or linear or mlp or block" > $O/test_gemm.txt 2>&1; echo "gemm tests rc $?"; tail -5 $O/test_gemm.txt
for v in 1 0 1 0; do EGV_PP_MIXED=$v timeout 300 python tools/pp_exp.py mixed$v 2>/dev/null | tail -1; done
bash tools/ab_multi.sh 3 "EGV_PP_MIXED=1" "EGV_PP_MIXED=0"
