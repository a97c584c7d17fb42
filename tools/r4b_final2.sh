#!/bin/bash
# final call of the round: full GPU suite with durations, smoke, kernel-trace + PMC profiles, default bench, side configurations
mkdir -p gpurun_out/r4final2; O=gpurun_out/r4final2
timeout 1500 python -m pytest tests -q -x -m gpu --durations=25 > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log | tail -3
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash tools/profile_round.sh r4final2/prof > $O/profile.log 2>&1; tail -4 $O/profile.log | cut -c1-200
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
python bench.py --workload dual --no-cpu-baseline --no-gemm-events 2>/dev/null | cut -c1-260 | tee $O/bench_dual.json
python bench.py --arch large14 --batch 4 --warmup 8 --no-cpu-baseline --no-gemm-events 2>/dev/null | cut -c1-260 | tee $O/bench_large14.json
python bench.py --arch large14 --batch 4 --fp8 --warmup 8 --no-cpu-baseline --no-gemm-events 2>/dev/null | cut -c1-260 | tee $O/bench_large14_fp8.json
timeout 300 python tools/infer_bench.py 2>/dev/null | tail -3 | tee $O/infer.log
