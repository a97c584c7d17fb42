#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5s12; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_hip_ops.py -x -q -k "cls_only" > $O/t1.txt 2>&1; echo "cls attn test rc $?"; tail -15 $O/t1.txt
timeout 900 python -m pytest tests/test_model_parity.py -x -q -k "cls_only or fold or tiny" > $O/t2.txt 2>&1; echo "tail model tests rc $?"; tail -25 $O/t2.txt
bash tools/ab_multi.sh 2 "EGV_CLS_TAIL=1" "EGV_CLS_TAIL=0"
