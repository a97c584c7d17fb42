import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egovlpv2_amd import hipops as ops

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

dev = 'cuda'
for M in (25096, 2048):
  for N in (768, 2304):
    for K in (64, 256, 768, 3072):
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        b = torch.randn(N, device=dev)
        r = torch.randn(M, N, device=dev).bfloat16()
        y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        out = []
        for name, kw in (('bias+res', dict(bias=b, res1=r)), ('bias', dict(bias=b)), ('plain', dict()), ('noepi', dict(act=99))):
            ms = timeit(lambda: ops.gemm(x, w, y, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, **kw))
            out.append(f"{name}={ms*1e3:7.1f}us")
        print(f"M={M} N={N} K={K}: " + "  ".join(out))
