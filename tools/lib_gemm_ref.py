"""Reference point only (not used by the product): what the vendor library GEMM behind torch.matmul reaches on the hot shapes."""
import torch
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
M = 25096
for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    x = torch.randn(M, K, device='cuda').bfloat16()
    w = (torch.randn(N, K, device='cuda') * 0.05).bfloat16()
    b = torch.randn(N, device='cuda').bfloat16()
    ms = timeit(lambda: torch.nn.functional.linear(x, w, b))
    print(f"torch linear  M={M} N={N} K={K}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:8.1f} TF")
    dy = torch.randn(M, N, device='cuda').bfloat16()
    ms = timeit(lambda: dy.t() @ x)
    print(f"torch wgrad   dW[{N},{K}]      : {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:8.1f} TF")
