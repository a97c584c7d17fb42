#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6s; mkdir -p $O
python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-gemm-events > /dev/null 2>&1
timeout 1500 bash tools/ab_multi.sh 2 "EGV_X=1" "EGV_PP_LIMIT_SLACK=8" "EGV_PP_LIMIT_SLACK=24" "EGV_PP_LIMIT_SLACK=32" "EGV_WGRAD_CUS=104" "EGV_PP_MIXED=1" "EGV_PP_MIXED=2" "EGV_PP_TILE_C0=40" "EGV_PP_TILE_C0=72" > $O/ab.log 2>&1
cat $O/ab.log
RUNS=6 timeout 600 python tools/repro_check.py 2>&1 | grep -v amdgpu.ids | tail -8
