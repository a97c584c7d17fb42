#!/bin/bash
# usage: sweep_env.sh "VAR1=a VAR2=b" "VAR1=c" ...   -> one bench line (ms/step) per environment setting
R=${GRAFT_REPO_ROOT:-$(pwd)}
for e in "$@"; do
  ms=$(env $e python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-gemm-events 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "$e : $ms ms/step"
done
