#!/bin/bash
# alternating A/B of the default bench under two environments: bash tools/ab_env.sh <rounds> "<env A>" "<env B>"
N=${1:-2}; A="$2"; B="$3"
for i in $(seq 1 $N); do
  for e in "$A" "$B"; do
    printf "%-50s " "[$e]"
    env $e python bench.py --no-cpu-baseline --no-gemm-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  done
done
