#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5s18; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_model_parity.py -x -q -k "tiny or reproducible or two_layer or cls_only or fold or dropout or odd_batch" > $O/t2.txt 2>&1; echo "parity tests rc $?"; tail -4 $O/t2.txt
timeout 900 python -m pytest tests/test_multirank_gpu.py -x -q -k "world2" > $O/t3.txt 2>&1; echo "world2 tests rc $?"; tail -4 $O/t3.txt
bash tools/ab_multi.sh 3 "EGV_TAIL_LATE=1" "EGV_TAIL_LATE=0"
for c in 80 88 104 112; do EGV_WGRAD_CUS=$c python bench.py --no-cpu-baseline --no-gemm-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wgrad cus $c', d['ms_per_step'])"; done
python bench.py --no-cpu-baseline --no-gemm-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['ms_per_step'])"
