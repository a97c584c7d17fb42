python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "linear or mlp" 2>&1 | tail -2
EGV_BENCH_SHAPES=1 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | grep -E "^shape kind=(8|10) gflop= *(118|88|29)|ms_per_step" | cut -c1-200
EGV_GEMM_NO_TAIL_SPLIT=1 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-gemm-events 2>&1 | tail -1 | cut -c1-200
