#!/bin/bash
# round 5, first GPU session: probes + new tests + baseline
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5s1; mkdir -p $O
cd $R
timeout 180 tools/probe/probe_cumask > $O/probe_cumask.txt 2>&1; echo "probe rc $?"
EGV_LIB_PATH=$R/tools/exp_libs/libegovlp_hip_instr.so timeout 300 python tools/gemm_pp_stamps.py > $O/stamps.txt 2>&1; echo "stamps rc $?"
timeout 900 python -m pytest tests/test_multirank_gpu.py -x -q -k "rccl" > $O/test_rccl.txt 2>&1; echo "rccl tests rc $?"; tail -5 $O/test_rccl.txt
timeout 1200 python -m pytest tests/test_model_parity.py -x -q -k "base_f4 or base_f16" > $O/test_base.txt 2>&1; echo "base tests rc $?"; tail -5 $O/test_base.txt
timeout 1200 python tools/bf16_grad_error.py base_f4 base_f16 > $O/bf16_grad.json 2> $O/bf16_grad.txt; echo "grad err rc $?"; head -12 $O/bf16_grad.txt
for i in 1 2; do python bench.py --no-cpu-baseline --no-gemm-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['value'])"; done
cat $O/probe_cumask.txt
