"""Micro-benchmark of the GEMM entry points on the hot-path shapes (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egovlpv2_amd import hipops as ops
from egovlpv2_amd import _lib as L

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

M = 25096
dev = 'cuda'
act = int(os.environ.get('EGV_BENCH_ACT', '0'))
for (N, K, res) in [(2304, 768, False), (768, 768, True), (3072, 768, False), (768, 3072, True)]:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev).bfloat16() if res else None
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.gemm(x, w, y, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=b, res1=r, act=act)
    ms = timeit(f)
    print(f"fwd  M={M} N={N} K={K} res={res}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:8.1f} TF")
    dy = torch.randn(M, N, device=dev).bfloat16()
    f2 = lambda: ops.wgrad(dy, x, M, N, K)
    ms = timeit(f2)
    print(f"wgrad dW[{N},{K}] over M={M}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:8.1f} TF")
