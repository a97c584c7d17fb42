#!/bin/bash
# One gpurun call that produces everything a round's judged numbers come from, on the final tree:
#   full GPU suite (timed), smoke(), default bench.py line, the profile set of tools/profile_round.sh, the in-step GEMM shape table.
# usage (GPU box, repo root): bash tools/round_final.sh <tag>  ->  gpurun_out/<tag>/   (copy the summaries into profiles/ afterwards)
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
python bench.py --steps 10 --warmup 5 --no-cpu-baseline > /dev/null 2>&1     # warms the box: the first processes on a cold box read 10-30 % slow
[ "$SKIP_TESTS" = 1 ] || { timeout 1300 python -m pytest tests -q -x -m gpu --durations=12 > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log | tail -3; }
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json
EGV_BENCH_SHAPES=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/shapes.json 2> $O/shapes.txt; python tools/shapes_md.py $O/shapes.txt > $O/gemm_shapes_instep.md; head -5 $O/gemm_shapes_instep.md
bash tools/profile_round.sh $TAG
cd $R; python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-200
