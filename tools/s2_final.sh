#!/bin/bash
# the round's final measurement set on the final tree (wraps tools/round_final.sh) + a 200-step run + the side configurations
TAG=${1:-s2final}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
bash tools/round_final.sh $TAG 2>&1 | tee $O/round_final.out
python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-gemm-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('200 steps:', d['ms_per_step'], d['step_ms_min_median_max'])" | tee $O/steps200.txt
python bench.py --workload dual --steps 20 --warmup 5 --no-cpu-baseline --no-gemm-events 2>/dev/null | cut -c1-260 | tee $O/dual.txt
python bench.py --arch large14 --batch 4 --steps 10 --warmup 8 --no-cpu-baseline --no-gemm-events 2>/dev/null | cut -c1-260 | tee $O/large14_bf16.txt
python bench.py --arch large14 --batch 4 --steps 10 --warmup 8 --fp8 --no-cpu-baseline --no-gemm-events 2>/dev/null | cut -c1-260 | tee $O/large14_fp8.txt
python tools/infer_bench.py 2>&1 | tail -3 | tee $O/infer.txt
