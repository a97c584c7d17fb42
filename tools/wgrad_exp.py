"""Where does the grouped weight-gradient launch's chip-level ceiling come from?  Per-K-tile-unit time (perfect balance assumed) of
the launch at several token counts M (operand footprint) and CU grants, and with every problem reading the SAME two operand
tensors (footprint of one pair, same strides).  python tools/wgrad_exp.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egovlpv2_amd import hipops as ops

D, Hd = 768, 3072
shapes = [(D, Hd), (Hd, D), (D, D), (3 * D, D), (D, D), (3 * D, D)]
ntile = sum((n // 256) * (k // 256) for n, k in shapes)


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


for M in (3136, 6272, 12544, 25096):
    g = torch.Generator().manual_seed(0)
    probs = [(torch.randn(M, N, generator=g).to(torch.bfloat16).cuda(), torch.randn(M, K, generator=g).to(torch.bfloat16).cuda(), True, None) for N, K in shapes]
    big_y = torch.randn(M, Hd, generator=g).to(torch.bfloat16).cuda()
    big_x = torch.randn(M, Hd, generator=g).to(torch.bfloat16).cuda()
    alias = [(big_y[:, :N], big_x[:, :K], True, None) for N, K in shapes]
    KT = (M + 63) // 64
    for name, pr in (('distinct', probs), ('aliased', alias)):
        row = []
        for cus in (72, 96, 144, 192, 256):
            t = timeit(lambda: ops.wgrad_grouped(pr, M, cus=cus))
            row.append(f"{cus}: {t:7.1f} us / {t * cus / (ntile * KT):.2f}")
        print(f"M={M:6d} {name:9s} | " + " | ".join(row), flush=True)
