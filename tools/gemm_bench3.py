import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egovlpv2_amd import hipops as ops
def timeit(fn, n=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
dev='cuda'
for M in (256,):
  for N,K in ((768,768),(3072,768),(768,3072)):
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev)*0.05).bfloat16()
    b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: ops.gemm(x, w, y, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=b))
    print(f"M={M} N={N} K={K}: {ms*1e3:.1f} us")
# launch-overhead reference: a trivial kernel (cast of 1k elements)
a = torch.randn(1024, device=dev); o = torch.empty(1024, device=dev, dtype=torch.bfloat16)
print("tiny cast kernel:", timeit(lambda: ops.cast(a, torch.bfloat16))*1e3, "us")
