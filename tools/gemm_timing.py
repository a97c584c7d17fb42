"""per-workgroup phase timing of gemm_ring_kernel (debug build hook egv_debug_timing)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from egovlpv2_amd import hipops as ops
from egovlpv2_amd._lib import LIB_PATH
raw = ctypes.CDLL(LIB_PATH)
raw.egv_debug_timing.argtypes = [ctypes.c_void_p]
dev='cuda'; M=25096
for (N,K,res) in [(2304,768,False),(768,768,True),(768,3072,True)]:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev)*0.05).bfloat16()
    b = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev).bfloat16() if res else None
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.gemm(x, w, y, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=b, res1=r)
    for _ in range(3): f()
    ntile = ((M+255)//256)*((N+127)//128)
    buf = torch.zeros(ntile*8, dtype=torch.int64, device=dev)
    raw.egv_debug_timing(buf.data_ptr()); f(); torch.cuda.synchronize(); raw.egv_debug_timing(None)
    s = buf.cpu().numpy().reshape(ntile, 8).astype(np.float64)
    t0 = s[:,0].min()
    d = lambda a,b: (s[:,b]-s[:,a])
    # clock64 = s_memtime at 100 MHz? report raw ticks and infer from total
    tot = (s[:,5].max()-t0)
    print(f"N={N} K={K}: tiles={ntile} total ticks={tot:.0f}; per-WG mean ticks: start->issue {d(0,1).mean():.0f}, issue->first tile {d(1,2).mean():.0f}, main loop {d(2,3).mean():.0f}, barrier {d(3,4).mean():.0f}, epilogue {d(4,5).mean():.0f}, whole {d(0,5).mean():.0f}; start spread {(s[:,0].max()-t0):.0f}")
