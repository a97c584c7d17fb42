# usage: bash tools/sweep_large14.sh "ENV=.. [FP8=1]" ...   -- ms/step of bench.py --arch large14 --batch 4 under each environment
for e in "$@"; do
  f=""; case "$e" in *FP8=1*) f="--fp8";; esac
  r=$(env ${e/FP8=1/} python bench.py --arch large14 --batch 4 --steps 5 --warmup 2 --no-cpu-baseline --no-gemm-events $f 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "$e : $r ms/step"
done
