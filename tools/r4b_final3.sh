#!/bin/bash
# last call of the round: full GPU suite + smoke + default bench on the final tree
mkdir -p gpurun_out/r4final3; O=gpurun_out/r4final3
timeout 1300 python -m pytest tests -q -x -m gpu --durations=8 > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log | tail -3
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
