python tools/gemm_bench.py 2>&1 | grep fwd
EGV_GEMM_CFG=5 python tools/gemm_bench.py 2>&1 | grep fwd
EGV_GEMM_CFG=6 python tools/gemm_bench.py 2>&1 | grep fwd
