"""MX-fp8 GEMM (egv_gemm_mx) beside the bf16 persistent GEMM (egv_gemm) on the ViT-L/14 shapes of BASELINE.json configs[4]
(M = 4 x 4113 video tokens): HIP-event time per launch, TFLOP/s, and the standalone quantisation pass of the A operand.
usage: python tools/mx_gemm_bench.py [M]"""
import sys
import torch
sys.path.insert(0, '.')
from egovlpv2_amd import hipops as ops
from egovlpv2_amd._lib import lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4 * 4113
shapes = [('qkv fwd', 3072, 1024), ('proj fwd / fc1 dgrad', 1024, 1024), ('fc1 fwd', 4096, 1024), ('fc2 fwd', 1024, 4096),
          ('qkv dgrad', 1024, 3072), ('fc2 dgrad', 4096, 1024), ('ViT-B qkv fwd', 2304, 768), ('ViT-B fc2 fwd', 768, 3072)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print(f"M = {M}")
print("| GEMM | N | K | bf16 us | bf16 TF | MX-fp8 us | MX TF | x | quant A us |")
print("|---|---|---|---|---|---|---|---|---|")
for name, N, K in shapes:
    x = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    w = (torch.randn(N, K, device='cuda') * 0.03).to(torch.bfloat16)
    bias = torch.randn(N, device='cuda')
    out = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
    t_bf = timed(lambda: ops.gemm(x, w, out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias))
    xq, xs = ops.quant_mx(x, 0)
    wq, ws = ops.quant_mx(w, 1)
    t_mx = timed(lambda: lib.egv_gemm_mx(M, N, K, xq.data_ptr(), xs.data_ptr(), wq.data_ptr(), ws.data_ptr(), out.data_ptr(), N, bias.data_ptr(),
                                         0, None, None, None, 0, N, None, None, None))
    t_q = timed(lambda: lib.egv_quant_mx(x.data_ptr(), M, K, K, xq.data_ptr(), xs.data_ptr(), 0, None))
    fl = 2.0 * M * N * K
    print(f"| {name} | {N} | {K} | {t_bf:.1f} | {fl / t_bf / 1e6:.0f} | {t_mx:.1f} | {fl / t_mx / 1e6:.0f} | {t_bf / t_mx:.2f} | {t_q:.1f} |")
