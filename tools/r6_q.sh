#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6q2; mkdir -p $O
timeout 1200 bash tools/ab_stats.sh 2 30 "EGV_X=1" "EGV_ITM_FIRST=1" "EGV_ITM_FIRST=1 GPU_MAX_HW_QUEUES=8" "EGV_EGONCE_TAIL_LATE=0" > $O/ab.log 2>&1
cat $O/ab.log
