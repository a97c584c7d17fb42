"""Micro-benchmark of the divided space/time attention core on the hot-path shape (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egovlpv2_amd import hipops as ops
from egovlpv2_amd._lib import lib

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B, Fr, N, H = 8, 16, 196, 12
S = 1 + Fr * N
qkv = torch.randn(B * S, 3 * H * 64, device='cuda').bfloat16().requires_grad_(True)
do = torch.randn(B * S, H * 64, device='cuda').bfloat16()
for mode in ('space', 'time'):
    for dbg in [int(x) for x in os.environ.get('DBG', '0').split(',')]:
        try:
            lib.egv_debug_attn(dbg)
        except Exception:
            pass
        f = lambda: ops.divided_attention(qkv.detach(), B, Fr, N, H, mode)
        t_f = timeit(f)
        o = ops.divided_attention(qkv, B, Fr, N, H, mode)
        def fb():
            qkv.grad = None
            o.backward(do, retain_graph=True)
        t_b = timeit(fb)
        print(f"{mode} dbg={dbg}: fwd {t_f:7.1f} us   bwd {t_b:7.1f} us", flush=True)
try:
    lib.egv_debug_attn(0)
except Exception:
    pass

# image -> text cross attention of a fused block (25 096 queries over 32 masked text keys): EGV_ATTN_FEWKEYS=0/1, EGV_ATTN_FEWKEYS_ITERS
L = 32
q = torch.randn(B * S, H * 64, device='cuda').bfloat16().requires_grad_(True)
kv = torch.randn(B * L, 2 * H * 64, device='cuda').bfloat16().requires_grad_(True)
mask = torch.zeros(B, L, device='cuda')
mask[:, 20:] = -10000.0
D = H * 64
f = lambda: ops.plain_attention(q.detach(), kv.detach()[:, :D], kv.detach()[:, D:], B, H, S, L, 0.125, mask=mask)
t_f = timeit(f)
o = ops.plain_attention(q, kv[:, :D], kv[:, D:], B, H, S, L, 0.125, mask=mask)
def fb():
    q.grad = None; kv.grad = None
    o.backward(do, retain_graph=True)
t_b = timeit(fb)
print(f"i2t FEWKEYS={os.environ.get('EGV_ATTN_FEWKEYS', '1')} ITERS={os.environ.get('EGV_ATTN_FEWKEYS_ITERS', '2')}: fwd {t_f:7.1f} us   bwd {t_b:7.1f} us (incl. the autograd glue of a per-op call)", flush=True)

# text -> image cross attention of a fused text layer (32 queries over 25 096 video keys, dropout 0.1): EGV_ATTN_FEWQ=0/1, EGV_ATTN_FEWQ_ITERS
qt = torch.randn(B * L, H * 64, device='cuda').bfloat16().requires_grad_(True)
kvv = torch.randn(B * S, 2 * H * 64, device='cuda').bfloat16().requires_grad_(True)
dot = torch.randn(B * L, H * 64, device='cuda').bfloat16()
f = lambda: ops.plain_attention(qt.detach(), kvv.detach()[:, :D], kvv.detach()[:, D:], B, H, L, S, 0.125, drop_p=0.1, drop_seed=7)
t_f = timeit(f)
o = ops.plain_attention(qt, kvv[:, :D], kvv[:, D:], B, H, L, S, 0.125, drop_p=0.1, drop_seed=7)
def fb():
    qt.grad = None; kvv.grad = None
    o.backward(dot, retain_graph=True)
t_b = timeit(fb)
print(f"t2i FEWQ={os.environ.get('EGV_ATTN_FEWQ', '1')} ITERS={os.environ.get('EGV_ATTN_FEWQ_ITERS', '6')}: fwd {t_f:7.1f} us   bwd {t_b:7.1f} us (incl. the autograd glue of a per-op call)", flush=True)
