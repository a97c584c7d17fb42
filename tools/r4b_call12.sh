#!/bin/bash
mkdir -p gpurun_out/r4b12; O=gpurun_out/r4b12
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_model_parity.py -q -x -m gpu -k "patch or train_mode_dropout or base_f4" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
bash tools/ab_multi.sh 2 "EGV_DUMMY=1" "EGV_TEXT_PRIORITY=0" "EGV_TEXT_PRIORITY=-1" 2>&1 | tee $O/ab.log
EGV_BENCH_SHAPES=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2> $O/shapes.txt > $O/bench_shapes.json; python tools/shapes_md.py $O/shapes.txt > $O/gemm_shapes_instep.md; wc -l $O/gemm_shapes_instep.md
