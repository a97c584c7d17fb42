#!/bin/bash
# kernel-trace of the step and windows around the fp32 generic GEMM launches (what are they, who waits for them)
TAG=${1:-f32w}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
env "$@" timeout -s KILL 400 rocprofv3 --kernel-trace -d $O/two -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-gemm-events > $O/two.log 2>&1
DB=$(find $O/two -name "*.db" | head -1)
for n in 12 13 14 15 16 17; do echo "=== occurrence $n"; python $R/tools/trace_window.py $DB "gemm_kernel<float, 0, 0, float>" $n 150 700 | cut -c1-170; done > $O/window.txt
find $O -name "*.db" -delete
