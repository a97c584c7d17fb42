#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5s20; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_model_parity.py -x -q -k "tiny or reproducible or two_layer or cls_only or fold or dropout or odd_batch" > $O/t2.txt 2>&1; echo "parity tests rc $?"; tail -4 $O/t2.txt
bash tools/ab_multi.sh 3 "EGV_ITM_HEAD_FORK=1 EGV_ITM_FIRST=1" "EGV_ITM_HEAD_FORK=1 EGV_ITM_FIRST=0" "EGV_ITM_HEAD_FORK=0 EGV_ITM_FIRST=0"
