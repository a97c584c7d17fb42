"""Micro-benchmark of the two GELU kinds of the persistent GEMM at the hot-path shape (run on the GPU box): fc1 forward (GELU + saved
pre-activation) and fc2's data gradient (GELU' operand), beside a plain launch of the same size.  A/B builds: EGV_LIB_PATH."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egovlpv2_amd import hipops as ops
from egovlpv2_amd import _lib as L


def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


M, D, Hd = 25096, 768, 3072
dev = 'cuda'
x = torch.randn(M, D, device=dev).bfloat16()
w1 = (torch.randn(Hd, D, device=dev) * 0.05).bfloat16()
b1 = torch.randn(Hd, device=dev)
h = torch.empty(M, Hd, device=dev, dtype=torch.bfloat16)
pre = torch.empty_like(h)
dy = torch.randn(M, D, device=dev).bfloat16()
w2t = (torch.randn(Hd, D, device=dev) * 0.05).bfloat16()     # fc2.weight^T: the NT operand of the data gradient
dpre = torch.empty_like(h)
res = []
for rep in range(int(os.environ.get('REPS', '3'))):
    t_plain = timeit(lambda: ops.gemm(x, w1, h, M=M, N=Hd, K=D, lda=D, ldb=D, ldc=Hd, bias=b1))
    t_fc1 = timeit(lambda: ops.gemm(x, w1, h, M=M, N=Hd, K=D, lda=D, ldb=D, ldc=Hd, bias=b1, act=L.ACT_GELU, pre=pre))
    t_dg = timeit(lambda: ops.gemm(dy, w2t, dpre, M=M, N=Hd, K=D, lda=D, ldb=D, ldc=Hd, aux=pre, dact=L.ACT_GELU))
    res.append((t_plain, t_fc1, t_dg))
    print(f"plain {t_plain:7.1f} us   fc1 (GELU + pre) {t_fc1:7.1f} us   fc2 dgrad (GELU') {t_dg:7.1f} us", flush=True)
