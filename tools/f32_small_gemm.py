"""fp32-storage GEMMs of the text tower (256 token rows): time per launch of forward / dgrad / wgrad shapes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egovlpv2_amd import hipops as ops
dev = 'cuda'
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for dt in (torch.float32, torch.bfloat16):
    for (M, N, K) in [(256, 768, 768), (256, 3072, 768), (256, 768, 3072)]:
        x = torch.randn(M, K, device=dev).to(dt)
        w = torch.randn(N, K, device=dev) * 0.05
        b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev, dtype=dt)
        wc = ops.compute_weight(w, dt)
        f = lambda: ops.gemm(x, wc, y, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=b)
        dy = torch.randn(M, N, device=dev).to(dt)
        g = lambda: ops.wgrad(dy, x, M, N, K, bias=True)
        print(f"{dt} M={M} N={N} K={K}: fwd {t(f):6.1f} us  wgrad {t(g):6.1f} us", flush=True)
