"""per-K-tile timestamps of the persistent GEMM (EGV_PP_STAMPS instrumentation build of the plain kind)"""
import os, sys, ctypes
os.environ['EGV_GEMM_PP'] = '1'; os.environ['EGV_PP_STAMPS'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from egovlpv2_amd import hipops as ops
from egovlpv2_amd._lib import LIB_PATH
raw = ctypes.CDLL(LIB_PATH)
raw.egv_debug_timing.argtypes = [ctypes.c_void_p]
dev = 'cuda'
np.set_printoptions(linewidth=200)
for (M, N, K) in [(100368, 2304, 768), (25096, 2304, 768), (25096, 768, 768), (25096, 768, 3072), (25096, 3072, 768)]:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.gemm(x, w, y, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=b)
    for _ in range(3): f()
    torch.cuda.synchronize()
    buf = torch.zeros(256 * 2 * 4 * 16, dtype=torch.int64, device=dev)
    raw.egv_debug_timing(buf.data_ptr()); f(); torch.cuda.synchronize(); raw.egv_debug_timing(None)
    s = buf.cpu().numpy().reshape(256, 2, 4, 16).astype(np.float64)
    KT = min(K // 64, 15)          # the kernel stamps the first 16 K-tile boundaries of a tile
    print(f"M={M} N={N} K={K}")
    for g in range(2):
        for ts in range(4):
            seg = s[:, g, ts, :KT + 1]
            ok = (seg > 0).all(axis=1)
            d = np.diff(seg[ok], axis=1)
            gap = ''
            if ts + 1 < 4:
                nxt = s[:, g, ts + 1, 0]
                okn = ok & (nxt > 0)
                gap = f" | end->next tile start {np.mean(nxt[okn] - seg[okn, KT]):.0f}"
            print(f"  wave row {g} tile {ts} (n={ok.sum()}): per-K-tile mean ticks {np.round(d.mean(axis=0)).astype(int)} total {d.sum(axis=1).mean():.0f}{gap}")
