#!/bin/bash
# kernel-trace of the step and a window of the backward pass around one FUSED video block (its image-to-text attention backward)
TAG=${1:-fw}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
env "$@" timeout -s KILL 400 rocprofv3 --kernel-trace -d $O/two -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-gemm-events > $O/two.log 2>&1
DB=$(find $O/two -name "*.db" | head -1)
python $R/tools/trace_window.py $DB "attn_dkv_mfma_kernel<2, 4" 20 2600 2600 > $O/window_bwd.txt
python $R/tools/trace_window.py $DB "attn_fwd_mfma_kernel<2, 4" 20 1500 1500 > $O/window_fwd.txt
find $O -name "*.db" -delete
