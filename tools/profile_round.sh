#!/bin/bash
# One gpurun call that refreshes the judged profiles of a round: two-stream and single-stream kernel traces, the three in-step
# PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy, separate runs, --kernel-trace only) and their summaries.  Every rocprofv3
# run is under `timeout -s KILL`: a counter set the hardware rejects leaves rocprofv3 hanging in its signal handler.
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>        -> gpurun_out/<tag>/
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-gemm-events"
run() { name=$1; shift; timeout -s KILL 400 rocprofv3 "$@" -d $O/$name -o p -- $B > $O/$name.log 2>&1 || echo "$name: rocprofv3 failed or timed out"; }
run two --kernel-trace --stats
EGV_NO_OVERLAP=1 run single --kernel-trace --stats
run fetch --pmc FETCH_SIZE --kernel-trace
run write --pmc WRITE_SIZE --kernel-trace
run sq --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace
db() { find $O/$1 -name "*.db" | head -1; }
python $R/tools/prof_summary.py $(db two) 6 60 > $O/two_stream_kernel_stats.md
python $R/tools/prof_summary.py $(db single) 6 60 > $O/single_stream_kernel_stats.md
python $R/tools/pmc_instep.py $(db fetch) $(db write) $(db sq) $(db two) 6 $O/pmc_instep > /dev/null
grep -h ms_per_step $O/two.log $O/single.log | cut -c1-200
head -3 $O/two_stream_kernel_stats.md; head -3 $O/single_stream_kernel_stats.md
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete; du -sh $O
