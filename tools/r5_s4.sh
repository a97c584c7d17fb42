#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5s4; mkdir -p $O; cd $R
PB=$R/tools/exp_libs/libegovlp_hip_pbulk.so
EGV_LIB_PATH=$PB timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "gemm or linear or mlp" > $O/test_gemm_pbulk.txt 2>&1; echo "pbulk gemm tests rc $?"; tail -3 $O/test_gemm_pbulk.txt
for v in prod pbulk prod pbulk; do
  if [ $v = prod ]; then timeout 300 python tools/pp_exp.py prod 2>/dev/null | tail -1; else EGV_LIB_PATH=$R/tools/exp_libs/libegovlp_hip_$v.so timeout 300 python tools/pp_exp.py $v 2>/dev/null | tail -1; fi
done
bash tools/ab_multi.sh 3 "EGV_X=0" "EGV_LIB_PATH=$PB"
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_model_parity.py -x -q -k "attention or attn or tiny or text_layer or roberta" > $O/test_attn.txt 2>&1; echo "attn tests rc $?"; tail -3 $O/test_attn.txt
timeout 1200 python tools/bf16_grad_error.py base_f4 > $O/bf16_grad.json 2> $O/bf16_grad.txt; echo "grad err rc $?"; head -14 $O/bf16_grad.txt | cut -c1-170
