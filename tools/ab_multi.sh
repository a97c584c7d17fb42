#!/bin/bash
# alternating comparison of the default bench under several environments: bash tools/ab_multi.sh <rounds> "<env 1>" "<env 2>" ...
N=${1:-2}; shift
for i in $(seq 1 $N); do
  for e in "$@"; do
    printf "%-60s " "[$e]"
    env $e python bench.py --no-cpu-baseline --no-gemm-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  done
done
