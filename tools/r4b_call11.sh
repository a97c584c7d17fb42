#!/bin/bash
mkdir -p gpurun_out/r4b11; O=gpurun_out/r4b11
timeout 900 python -m pytest tests/test_model_parity.py -q -x -m gpu -k "not large and not long_clip" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
bash tools/ab_multi.sh 3 "EGV_MLM_TOP_LATE=0" "EGV_MLM_TOP_LATE=1" 2>&1 | tee $O/ab.log
python tools/step_timeline.py 2>&1 | grep -v Warning > $O/timeline.log; grep -A12 "largest single gaps" $O/timeline.log
EGV_BENCH_SHAPES=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2> $O/shapes.txt > $O/bench_shapes.json; python tools/shapes_md.py $O/shapes.txt > $O/gemm_shapes_instep.md; head -30 $O/gemm_shapes_instep.md
