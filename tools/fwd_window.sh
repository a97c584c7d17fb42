#!/bin/bash
# kernel-trace of the step and a window of the forward pass around one fc1 launch
TAG=${1:-fw}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
env "$@" timeout -s KILL 400 rocprofv3 --kernel-trace -d $O/two -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-gemm-events > $O/two.log 2>&1
DB=$(find $O/two -name "*.db" | head -1)
python $R/tools/trace_window.py $DB "gemm_pp_kernel<0, true, true" 120 1800 1800 > $O/window.txt
find $O -name "*.db" -delete
grep -v "^s1" $O/window.txt | cut -c1-130
