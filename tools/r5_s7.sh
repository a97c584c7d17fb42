#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5s7; mkdir -p $O; cd $R
bash tools/ab_multi.sh 2 "EGV_PP_MIXED=1" "EGV_PP_MIXED=0" "EGV_PP_MIXED=2"
for m in 0 1 2; do EGV_PP_MIXED=$m EGV_BENCH_SHAPES=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2> $O/shapes$m.txt > /dev/null; python tools/shapes_md.py $O/shapes$m.txt > $O/shapes$m.md; done
timeout 1200 python tools/bf16_grad_error.py base_f16 > $O/bf16_grad.json 2> $O/bf16_grad.txt; echo "grad err rc $?"; head -12 $O/bf16_grad.txt | cut -c1-170
EGV_ITM_RES32=0 timeout 1200 python tools/bf16_grad_error.py base_f16 > $O/bf16_grad_noitm32.json 2> $O/bf16_grad_noitm32.txt; head -12 $O/bf16_grad_noitm32.txt | cut -c1-170
