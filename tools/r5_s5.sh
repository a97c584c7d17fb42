#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5s5; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_hip_ops.py -x -q > $O/test_ops.txt 2>&1; echo "ops tests rc $?"; tail -3 $O/test_ops.txt
timeout 1200 python -m pytest tests/test_model_parity.py -x -q -k "tiny or base_f4 or two_layer or dropout" > $O/test_par.txt 2>&1; echo "parity tests rc $?"; tail -3 $O/test_par.txt
timeout 1200 python tools/bf16_grad_error.py base_f4 base_f16 > $O/bf16_grad.json 2> $O/bf16_grad.txt; echo "grad err rc $?"; cat $O/bf16_grad.txt | cut -c1-170 | grep -v "^   " ; grep "^   " $O/bf16_grad.txt | head -8 | cut -c1-170
