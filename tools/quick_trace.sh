#!/bin/bash
# one kernel-trace pass over bench.py (two-stream step as timed) + per-kernel summary.  bash tools/quick_trace.sh <tag> [env assignments...]
TAG=${1:-qt}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
env "$@" timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $O/two -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-gemm-events > $O/two.log 2>&1
python $R/tools/prof_summary.py $(find $O/two -name "*.db" | head -1) 6 45 > $O/two_stream_kernel_stats.md
grep -h ms_per_step $O/two.log | cut -c1-200
sed -n '/^| kernel/,$p' $O/two_stream_kernel_stats.md | cut -c1-150 | head -40
find $O -name "*.db" -delete
