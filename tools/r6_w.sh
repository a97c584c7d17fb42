#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6w; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_model_parity.py -m gpu -x -q -k "grouped or wgrad or weight_grad or accumulation or reproducible or base_f4 or linear_forms" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -6 $O/tests.log
for r in 1 2; do
EGV_LIB_PATH=$PWD/tools/exp_libs/libegovlp_hip_w4p4.so timeout 300 python tools/wgrad_group_bench.py 2>&1 | grep -v amdgpu.ids | sed 's/^/[4ph] /' >> $O/wg.log
timeout 300 python tools/wgrad_group_bench.py 2>&1 | grep -v amdgpu.ids | sed 's/^/[2ph] /' >> $O/wg.log
done
grep "grouped:\|cus= *96\|cus= *256\|cus= *144\|rel diff" $O/wg.log
timeout 900 bash tools/ab_multi.sh 3 "EGV_LIB_PATH=$PWD/tools/exp_libs/libegovlp_hip_w4p4.so" "EGV_X=1" "EGV_WGRAD_CUS=88" "EGV_WGRAD_CUS=80" > $O/ab.log 2>&1
cat $O/ab.log
