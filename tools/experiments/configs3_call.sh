#!/bin/bash
# Round-5 third session, second gpurun call: 33-row time attention (32-frame clips, BASELINE.json configs[3]) on its own space-kernel
# instance -- fp64 tests, kernel trace of tools/infer_bench.py, A/B against the five-tile form; plus a sweep of the persistent GEMM's CU cap.
#   gpurun --timeout 840 -- 'bash tools/experiments/configs3_call.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s3b
mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_hip_ops.py -x -q -k "divided_attention" 2>&1 | tail -30 ) > $O/tests_ops.log 2>&1
( timeout 300 python -m pytest tests/test_model_parity.py -x -q -k "long_clip or tiny_embeddings" 2>&1 | tail -40 ) > $O/tests_model.log 2>&1
tail -n 3 $O/tests_ops.log $O/tests_model.log
for i in 1 2; do
  echo -n "nt3   "; python tools/infer_bench.py 2>&1 | tail -n 1
  echo -n "nont3 "; EGV_LIB_PATH=$R/tools/exp_libs/libegovlp_hip_nont3.so python tools/infer_bench.py 2>&1 | tail -n 1
done | tee $O/infer_ab.log
( cd /tmp; timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/c3 -o p -- python $R/tools/infer_bench.py > $O/c3.log 2>&1 || echo "rocprofv3 failed" )
python tools/prof_summary.py $(find $O/c3 -name "*.db" | head -1) 7 40 > $O/configs3_kernel_stats.md 2> $O/prof_summary.err
head -n 30 $O/configs3_kernel_stats.md
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
bash tools/ab_multi.sh 2 "EGV_PP_CUS=0" "EGV_PP_CUS=248" "EGV_PP_CUS=240" 2>&1 | tee $O/ab_ppcus.log
