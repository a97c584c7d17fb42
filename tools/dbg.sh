timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --optimizer 2>&1 | tail -8 | cut -c1-600
echo rc=$?
