"""Run one GEMM shape N times (for rocprofv3 / A-B timing).  usage: gemm_pp_one.py M N K kind [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egovlpv2_amd import hipops as ops
from egovlpv2_amd import _lib as L
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4] if len(sys.argv) > 4 else 'bias'
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = 'cuda'
torch.manual_seed(0)
x = torch.randn(M, K, device=dev).bfloat16()
w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
b = torch.randn(N, device=dev)
r1 = torch.randn(M, N, device=dev).bfloat16() if kind == 'res' else None
y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
pre = torch.empty_like(y) if kind == 'gelu_pre' else None
f = lambda: ops.gemm(x, w, y, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=b, res1=r1, pre=pre, act=L.ACT_GELU if kind == 'gelu_pre' else 0)
for _ in range(3):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    f()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"PP={os.environ.get('EGV_GEMM_PP','0')} M={M} N={N} K={K} {kind}: {ms*1e3:.1f} us {2*M*N*K/ms/1e9:.1f} TF")
