#!/bin/bash
O=gpurun_out/s2c4; mkdir -p $O
python bench.py --steps 10 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
timeout 900 python -m pytest tests/test_model_parity.py -q -x -m gpu -k "reproducible or base_f4 or base_f16" > $O/tests.log 2>&1; tail -2 $O/tests.log
bash tools/ab_multi.sh 3 "EGV_NOP=0" "EGV_WGRAD_TAIL_CUS=144" "EGV_WGRAD_TAIL_CUS=176" 2>&1 | tee $O/ab_tail.txt
