"""Separate the staging cost from the per-query-tile cost of the MFMA attention forward: same 197 keys, 1 / 4 / 13 query tiles."""
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
    return e0.elapsed_time(e1) / n * 1e3
B, H, nk = 128, 12, 197
D = H * 64
k = torch.randn(B * nk, D, device='cuda').bfloat16()
v = torch.randn(B * nk, D, device='cuda').bfloat16()
for nq in (16, 64, 112, 208):
    q = torch.randn(B * nq, D, device='cuda').bfloat16()
    t = timeit(lambda: ops.plain_attention(q, k, v, B, H, nq, nk, 0.125))
    print(f"nq={nq:4d} ({nq // 16} q-tiles per problem): fwd {t:7.1f} us", flush=True)

print("backward (dQ kernel + dK/dV kernel):")
for nq in (64, 112, 208):
    q = torch.randn(B * nq, D, device='cuda').bfloat16().requires_grad_(True)
    kk = k.clone().requires_grad_(True)
    vv = v.clone().requires_grad_(True)
    o = ops.plain_attention(q, kk, vv, B, H, nq, nk, 0.125)
    do = torch.randn_like(o)
    def fb():
        q.grad = kk.grad = vv.grad = None
        o.backward(do, retain_graph=True)
    print(f"nq={nq:4d}: bwd {timeit(fb):7.1f} us", flush=True)
