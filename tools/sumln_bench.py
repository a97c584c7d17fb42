"""egv_sum_layernorm at the video-token shape: the four call forms of a SpaceTimeBlock forward with the fp32 residual stream
(rotating buffers: nothing is served from the 256 MB last-level cache).  us per call and the HBM rate over the operand bytes."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egovlpv2_amd import _lib as L
M, D, NB = 25096, 768, 6
dev = 'cuda'
b32 = [torch.randn(M, D, device=dev) for _ in range(NB)]
o32 = [torch.empty(M, D, device=dev) for _ in range(NB)]
d = [[torch.randn(M, D, device=dev).bfloat16() for _ in range(NB)] for _ in range(3)]
o16 = [torch.empty(M, D, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
y = [torch.empty(M, D, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
st = torch.empty(M, 2, device=dev)
g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())      # noqa: E731


def run(i, form):
    k = i % NB
    if form == 'ln3':       # y = LN(base32)
        a = (p(b32[k]), None, None, None, None, None, None, None, p(y[k]))
        nb = 4 + 2
    elif form == 'ln1':     # sum16, y from base32 + d1
        a = (p(b32[k]), None, p(d[0][k]), None, None, None, None, p(o16[k]), p(y[k]))
        nb = 4 + 2 + 2 + 2
    elif form == 'final':   # out32, out16 = base32 + d1 + d2
        a = (p(b32[k]), None, p(d[0][k]), p(d[1][k]), None, None, p(o32[k]), p(o16[k]), None)
        nb = 4 + 2 + 2 + 4 + 2
    rc = L.lib.egv_sum_layernorm(*a, p(g), p(b), p(st) if a[8] else None, M, D, 1e-5, None)
    assert rc == 0
    return nb


for form in ('ln3', 'ln1', 'final'):
    for i in range(6):
        nb = run(i, form)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 60
    for i in range(n):
        run(i, form)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{sys.argv[1] if len(sys.argv) > 1 else ''} {form}: {us:.1f} us  {nb * M * D / us / 1e6:.2f} TB/s")
