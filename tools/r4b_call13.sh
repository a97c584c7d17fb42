#!/bin/bash
mkdir -p gpurun_out/r4b13; O=gpurun_out/r4b13
timeout 900 python -m pytest tests/test_model_parity.py tests/test_multirank_gpu.py -q -x -m gpu -k "not large and not long_clip" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
bash tools/ab_multi.sh 3 "EGV_ITM_PREFIX_INTERLEAVE=0" "EGV_ITM_PREFIX_INTERLEAVE=1" 2>&1 | tee $O/ab.log
python tools/step_timeline.py 2>&1 | grep -v Warning > $O/timeline.log; head -3 $O/timeline.log; grep -A10 "largest single gaps" $O/timeline.log
