"""Sustained-clock check of the NT GEMM: TF/s per window of launches over ~0.5 s, same buffers vs rotating buffers."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egovlpv2_amd import hipops as ops

M, dev = 25096, 'cuda'
for (N, K) in [(2304, 768), (3072, 768), (768, 3072)]:
    for nbuf in (1, 6):
        xs = [torch.randn(M, K, device=dev).bfloat16() for _ in range(nbuf)]
        ws = [(torch.randn(N, K, device=dev) * 0.05).bfloat16() for _ in range(nbuf)]
        ys = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(nbuf)]
        b = torch.randn(N, device=dev)
        def f(i):
            j = i % nbuf
            ops.gemm(xs[j], ws[j], ys[j], M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=b)
        for i in range(5): f(i)
        torch.cuda.synchronize()
        W, per = 12, 400
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(W + 1)]
        evs[0].record()
        for w in range(W):
            for i in range(per): f(i)
            evs[w + 1].record()
        torch.cuda.synchronize()
        tf = [2 * M * N * K * per / (evs[w].elapsed_time(evs[w + 1]) * 1e9) for w in range(W)]
        print(f"N={N} K={K} nbuf={nbuf}: " + " ".join(f"{t:6.0f}" for t in tf), flush=True)
