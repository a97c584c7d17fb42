"""workload for rocprofv3 --pmc passes: the grouped weight-gradient launch of one SpaceTimeBlock at several CU grants"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egovlpv2_amd import hipops as ops
M, D, Hd = 25096, 768, 3072
shapes = [(D, Hd), (Hd, D), (D, D), (3 * D, D), (D, D), (3 * D, D)]
g = torch.Generator().manual_seed(0)
probs = [(torch.randn(M, N, generator=g).to(torch.bfloat16).cuda(), torch.randn(M, K, generator=g).to(torch.bfloat16).cuda(), True, None) for N, K in shapes]
for cus in (256, 224, 144, 72):
    for _ in range(3):
        ops.wgrad_grouped(probs, M, cus=cus)
    torch.cuda.synchronize()
for dy, x, _, _ in probs:
    for _ in range(3):
        ops.wgrad(dy, x, M, dy.shape[1], x.shape[1], bias=True)
torch.cuda.synchronize()
