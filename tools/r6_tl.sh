#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6tl; mkdir -p $O
timeout 600 python tools/step_timeline.py --dump 2>&1 | grep -v amdgpu.ids > $O/timeline.log
head -60 $O/timeline.log
