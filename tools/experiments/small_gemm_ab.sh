#!/bin/bash
# Round-5 third session, one gpurun call: tests of the small-grid text GEMM (egv_gemm6.hip) and of the reversed wave order of the
# space-attention backward, per-kind GEMM event tables with and without the new kernel, alternating A/B of the step.
#   gpurun --timeout 1500 -- 'bash tools/small_gemm_ab.sh'
mkdir -p gpurun_out/s3
O=gpurun_out/s3
export TMPDIR=/tmp
( timeout 330 python -m pytest tests/test_hip_ops.py -x -q -k "small_grid or test_mlp or linear_forms or linear_act or linear_gate or divided_attention or persistent_gemm_bitwise or fused_attention_backward" 2>&1 | tail -15 ) > $O/tests_ops.log 2>&1
( timeout 330 python -m pytest tests/test_model_parity.py -x -q -k "tiny or base_f4 or reproducible or batch_independence_full or dropout" 2>&1 | tail -15 ) > $O/tests_model.log 2>&1
tail -3 $O/tests_ops.log $O/tests_model.log
# per-kind GEMM tables (HIP events around every launch, in-step and isolated)
python bench.py > $O/bench_new.json 2> $O/bench_new.err
EGV_GEMM_SMALL_M=0 python bench.py --no-cpu-baseline > $O/bench_old.json 2> $O/bench_old.err
python - <<'PY'
import json
for tag in ('new', 'old'):
    try:
        d = json.loads(open(f'gpurun_out/s3/bench_{tag}.json').read().strip().split('\n')[-1])
    except Exception as e:
        print(tag, 'failed', e); continue
    print(tag, d['ms_per_step'], d['value'])
    for which in ('all_gemm', 'all_gemm_isolated'):
        for k, v in d['roofline'][which].items():
            if 'small' in k or '128x128' in k:
                print('   ', which, k[:40], v)
PY
bash tools/ab_multi.sh 2 "EGV_GEMM_SMALL_M=0 EGV_SPACE_BWD_REV=0" "EGV_GEMM_SMALL_M=768 EGV_SPACE_BWD_REV=0" "EGV_GEMM_SMALL_M=0 EGV_SPACE_BWD_REV=1" \
   "EGV_GEMM_SMALL_M=768 EGV_SPACE_BWD_REV=1" "EGV_GEMM_SMALL_SPLITK=512" "EGV_LIB_PATH=$PWD/tools/exp_libs/libegovlp_hip_s6plain.so" 2>&1 | tee $O/ab.log
