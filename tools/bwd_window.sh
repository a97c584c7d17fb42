#!/bin/bash
# kernel-trace of the step and a window of the backward pass around one fc2 data-gradient launch (who runs beside whom)
TAG=${1:-bw}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
env "$@" timeout -s KILL 400 rocprofv3 --kernel-trace -d $O/two -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-gemm-events > $O/two.log 2>&1
DB=$(find $O/two -name "*.db" | head -1)
python $R/tools/trace_window.py $DB "gemm_pp_kernel<2" 100 2500 2500 > $O/window.txt
find $O -name "*.db" -delete
cat $O/window.txt
