#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c; mkdir -p $O
timeout 900 python -m pytest tests/test_model_parity.py tests/test_multirank_gpu.py -m gpu -x -q -k "checkpointing or world2" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -8 $O/tests.log
