"""Grouped weight-gradient launch (egv_gemm5.hip) against six separate launches of the ping-pong kernel + slab reductions, at the
shapes of one SpaceTimeBlock backward (M = 25096 tokens, D = 768).  python tools/wgrad_group_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egovlpv2_amd import hipops as ops

M, D, Hd = 25096, 768, 3072
shapes = [(D, Hd), (Hd, D), (D, D), (3 * D, D), (D, D), (3 * D, D)]
g = torch.Generator().manual_seed(0)
probs = []
for N, K in shapes:
    dy = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    probs.append((dy, x, True, None))
flops = sum(2.0 * M * n * k for n, k in shapes)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


t_g = timeit(lambda: ops.wgrad_grouped(probs, M))
t_s = timeit(lambda: [ops.wgrad(dy, x, M, dy.shape[1], x.shape[1], bias=True) for dy, x, _, _ in probs])
print(f"grouped: {t_g * 1e3:.1f} us = {flops / t_g / 1e9:.0f} TFLOP/s;   six launches + reductions: {t_s * 1e3:.1f} us = {flops / t_s / 1e9:.0f} TFLOP/s")
a = ops.wgrad_grouped(probs, M)
b = [ops.wgrad(dy, x, M, dy.shape[1], x.shape[1], bias=True) for dy, x, _, _ in probs]
torch.cuda.synchronize()
for (dw, db), (rw, rb) in zip(a, b):
    print('rel diff dW %.2e  db %.2e' % (((dw - rw).norm() / rw.norm()).item(), ((db - rb).norm() / rb.norm()).item()))
KT = (M + 63) // 64
ntile = sum((n // 256) * (k // 256) for n, k in shapes)
for cus in (256, 248, 224, 192, 160, 144, 128, 96, 72, 48):
    t = timeit(lambda: ops.wgrad_grouped(probs, M, cus=cus), n=10)
    print(f"cus={cus:4d}: {t * 1e3:8.1f} us = {flops / t / 1e9:6.0f} TFLOP/s;  CU-time {t * cus:7.1f} ms*CU;  per K-tile unit if perfectly balanced {t * 1e3 * cus / (ntile * KT):.2f} us")
