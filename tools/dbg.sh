python tools/attn_bench.py 2>&1 | grep -E "space|time"
python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "attention" 2>&1 | tail -2
for i in 1 2; do timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gemm-events 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'; done
