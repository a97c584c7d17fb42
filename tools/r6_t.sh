#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6t; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_ops.py tests/test_model_parity.py -m gpu -x -q -k "gemm or linear or grouped or accumulation or base_f16 or base_f4 or reproducible or block_entry or tiny_emb or fold" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -25 $O/tests.log
timeout 600 bash tools/ab_multi.sh 3 "EGV_WGRAD_ACC=0" "EGV_WGRAD_ACC=1" > $O/ab.log 2>&1
cat $O/ab.log
