"""Grouped weight-gradient launch of an unfused ViT-B SpaceTimeBlock (144 tiles, M = 25 096): us per launch and us per 256 x 256 x 64
K-tile unit (perfect balance assumed) at several CU grants, with and without bias gradients, for the library EGV_LIB_PATH names.
python tools/wgrad_exp2.py [label]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egovlpv2_amd import hipops as ops

D, Hd, M = 768, 3072, 25096
shapes = [(D, Hd), (Hd, D), (D, D), (3 * D, D), (D, D), (3 * D, D)]
ntile = sum((n // 256) * (k // 256) for n, k in shapes)
KT = (M + 63) // 64


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


g = torch.Generator().manual_seed(0)
ops_ = [(torch.randn(M, N, generator=g).to(torch.bfloat16).cuda(), torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()) for N, K in shapes]
label = sys.argv[1] if len(sys.argv) > 1 else os.environ.get('EGV_LIB_PATH', 'product')
for bias in (True, False):
    pr = [(dy, x, bias, None) for dy, x in ops_]
    row = []
    for cus in (72, 96, 144, 256):
        t = timeit(lambda: ops.wgrad_grouped(pr, M, cus=cus))
        row.append(f"{cus}: {t:7.1f} us / {t * cus / (ntile * KT):.2f}")
    print(f"{label:10s} bias={int(bias)} | " + " | ".join(row), flush=True)
