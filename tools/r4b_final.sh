#!/bin/bash
# full GPU test suite, smoke, judged profiles and the default bench line of the round in one call
mkdir -p gpurun_out/r4final; O=gpurun_out/r4final
timeout 1500 python -m pytest tests -q -x -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash tools/profile_round.sh r4final/prof > $O/profile.log 2>&1; tail -8 $O/profile.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json
