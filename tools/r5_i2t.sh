#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5i2t; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_hip_ops.py -x -q -k "few_keys or plain_attention or block_entry or fp32_residual" > $O/t.txt 2>&1; echo "tests rc $?"; tail -5 $O/t.txt
for it in 1 2 4 8; do EGV_ATTN_FEWKEYS_ITERS=$it python tools/attn_bench.py 2>&1 | grep i2t; done
EGV_ATTN_FEWKEYS=0 python tools/attn_bench.py 2>&1 | grep i2t
bash tools/ab_multi.sh 3 "EGV_ATTN_FEWKEYS=1" "EGV_ATTN_FEWKEYS=0"
