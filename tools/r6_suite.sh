#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6suite; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=45 > $O/suite.log 2>&1; echo "rc=$?" >> $O/suite.log
tail -70 $O/suite.log
