#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5s8; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_model_parity.py -x -q -k "fold or tiny or two_layer or dropout or reproducible" > $O/test_fold.txt 2>&1; echo "fold tests rc $?"; tail -5 $O/test_fold.txt
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "block" > $O/test_block.txt 2>&1; echo "block tests rc $?"; tail -3 $O/test_block.txt
bash tools/ab_multi.sh 3 "EGV_LN_FOLD=1" "EGV_LN_FOLD=0" "EGV_PP_MIXED=0"
