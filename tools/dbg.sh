for v in "" "EGV_AB_VOCAB=1" "EGV_AB_LN=1" "EGV_AB_VOCAB=1 EGV_AB_LN=1" ""; do
  echo "== $v"; env $v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-gemm-events 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
