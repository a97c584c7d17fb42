"""Timing of the persistent GEMM on the hot-path shapes for one build of the library (EGV_LIB_PATH selects it):
python tools/pp_exp.py [tag].  Used to A/B experimental builds (tools/pp_exp_build.sh) against the product library."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egovlpv2_amd import hipops as ops
from egovlpv2_amd import _lib as L

dev = 'cuda'
torch.manual_seed(0)


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


cases = [('qkv', 25096, 2304, 768, 'bias'), ('projd', 25096, 768, 768, 'plain'), ('proj', 25096, 768, 768, 'res'),
         ('fc1', 25096, 3072, 768, 'gelu_pre'), ('fc2d', 25096, 3072, 768, 'dact'), ('fc2', 25096, 768, 3072, 'res'),
         ('fc1d', 25096, 768, 3072, 'plain'), ('qkvd', 25096, 768, 2304, 'plain')]
out = {}
for rep in range(2):
    for name, M, N, K, kind in cases:
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        b = torch.randn(N, device=dev)
        r1 = torch.randn(M, N, device=dev).bfloat16()
        aux = torch.randn(M, N, device=dev).bfloat16()
        y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        pre = torch.empty_like(y)
        kw = dict(M=M, N=N, K=K, lda=K, ldb=K, ldc=N)
        if kind == 'bias':
            f = lambda: ops.gemm(x, w, y, bias=b, **kw)
        elif kind == 'plain':
            f = lambda: ops.gemm(x, w, y, **kw)
        elif kind == 'res':
            f = lambda: ops.gemm(x, w, y, bias=b, res1=r1, **kw)
        elif kind == 'gelu_pre':
            f = lambda: ops.gemm(x, w, y, bias=b, act=L.ACT_GELU, pre=pre, **kw)
        elif kind == 'dact':
            f = lambda: ops.gemm(x, w, y, aux=aux, dact=L.ACT_GELU, **kw)
        us = timeit(f)
        out.setdefault(name, []).append(round(us, 1))
tag = sys.argv[1] if len(sys.argv) > 1 else 'prod'
print(tag, json.dumps({k: (v, round(2.0 * dict((c[0], c[1] * c[2] * c[3]) for c in cases)[k] / min(v) / 1e6)) for k, v in out.items()}))
