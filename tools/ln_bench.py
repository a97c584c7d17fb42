"""LayerNorm forward / backward kernel timing at the video-token shape (run on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egovlpv2_amd import hipops as ops
dev = 'cuda'
M, D = 25096, 768
x = torch.randn(M, D, device=dev).bfloat16().requires_grad_(True)
g = torch.ones(D, device=dev, requires_grad=True); b = torch.zeros(D, device=dev, requires_grad=True)
dy = torch.randn(M, D, device=dev).bfloat16()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
y = ops.layernorm(x, g, b, 1e-5)
print(f"LN_BLOCKS={os.environ.get('EGV_LN_BLOCKS','default')}: fwd {t(lambda: ops.layernorm(x, g, b, 1e-5)):.1f} us; bwd (+colsum) {t(lambda: torch.autograd.grad(y, (x, g, b), dy, retain_graph=True)):.1f} us")
