#!/bin/bash
# Round-5 third session, third gpurun call: EGV_BLOCK_INFER (no pre-activation store in calls made under no_grad) -- bitwise test, the
# inference tests, configs[3] A/B.   gpurun --timeout 600 -- 'bash tools/experiments/infer_lean_call.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s3c
mkdir -p $O
( timeout 400 python -m pytest tests/test_model_parity.py -x -q -k "inference_calls_skip or long_clip or egomcq or tiny_embeddings or batch_independence_full" 2>&1 | tail -40 ) > $O/tests_model.log 2>&1
tail -n 4 $O/tests_model.log
for i in 1 2; do
  echo -n "lean  "; python tools/infer_bench.py 2>&1 | tail -n 1
  echo -n "full  "; EGV_INFER_LEAN=0 python tools/infer_bench.py 2>&1 | tail -n 1
done | tee $O/infer_ab.log
