#!/bin/bash
# alternating comparison with per-step statistics: bash tools/ab_stats.sh <rounds> <steps> "<env 1>" "<env 2>" ...
N=${1:-2}; S=${2:-30}; shift; shift
for i in $(seq 1 $N); do
  for e in "$@"; do
    printf "%-70s " "[$e]"
    env $e python bench.py --steps $S --no-cpu-baseline --no-gemm-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms_min_median_max'])"
  done
done
