#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s2c7; mkdir -p $O
python bench.py --steps 10 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $O/tr -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-gemm-events > $O/tr.log 2>&1
cd $R
python tools/trace_dump.py $(find $O/tr -name "*.db" | head -1) $O/step.csv 1 | tail -1
find $O -name "*.db" -delete; rm -rf $O/tr
wc -l $O/step.csv
