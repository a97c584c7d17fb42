#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5s16; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "patch" > $O/t1.txt 2>&1; echo "patch tests rc $?"; tail -3 $O/t1.txt
timeout 1200 python -m pytest tests/test_model_parity.py -x -q -k "tiny or reproducible or two_layer or dual or cls_only or fold" > $O/t2.txt 2>&1; echo "parity tests rc $?"; tail -4 $O/t2.txt
timeout 900 python -m pytest tests/test_multirank_gpu.py -x -q > $O/t3.txt 2>&1; echo "multirank tests rc $?"; tail -4 $O/t3.txt
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-gemm-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['value'])"; done
