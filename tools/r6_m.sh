#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6m; mkdir -p $O
timeout 900 python -m pytest tests/test_model_parity.py tests/test_multirank_gpu.py -m gpu -x -q -k "base_f4 or tiny_emb or reproducible or world2_full_step_vs_oracle\[True-flat or odd_batch or dropout or egonce_only" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -6 $O/tests.log
timeout 900 bash tools/ab_stats.sh 3 20 "EGV_TAIL_REST_AUX=0" "EGV_TAIL_REST_AUX=1" > $O/ab.log 2>&1
cat $O/ab.log
timeout 300 python tools/step_timeline.py --dump 2>&1 | grep -v amdgpu.ids > $O/timeline.log
grep -A12 "largest single gaps" $O/timeline.log
