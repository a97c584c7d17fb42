"""Fused divided-attention backward (csrc/egv_attn_mfma.hip: attn_bwd_fused_kernel) vs the dQ + dK/dV kernel pair and vs torch fp32:
max abs differences and time per backward at the configs[2] shape (B=8, 16 frames x 196 patches, 12 heads)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egovlpv2_amd import hipops as ops


def ref(qkv, B, Fr, N, H, mode):
    D = H * 64
    S = 1 + Fr * N
    x = qkv.float().view(B, S, 3, H, 64).permute(2, 0, 3, 1, 4)          # 3, B, H, S, 64
    q, k, v = x[0] * 0.125, x[1], x[2]
    cls_out = torch.softmax(q[:, :, :1] @ k.transpose(-1, -2), -1) @ v
    if mode == 'space':
        def grp(t): return t[:, :, 1:].reshape(B, H, Fr, N, 64)
    else:
        def grp(t): return t[:, :, 1:].reshape(B, H, Fr, N, 64).transpose(2, 3)
    qg, kg, vg = grp(q), grp(k), grp(v)
    G = qg.shape[2]
    kc = k[:, :, :1].unsqueeze(2).expand(-1, -1, G, -1, -1)
    vc = v[:, :, :1].unsqueeze(2).expand(-1, -1, G, -1, -1)
    kk, vv = torch.cat([kc, kg], 3), torch.cat([vc, vg], 3)
    og = torch.softmax(qg @ kk.transpose(-1, -2), -1) @ vv
    if mode == 'time':
        og = og.transpose(2, 3)
    og = og.reshape(B, H, Fr * N, 64)
    o = torch.cat([cls_out, og], 2)                                     # B, H, S, 64
    return o.permute(0, 2, 1, 3).reshape(B * S, D)


def main():
    dev = 'cuda'
    torch.manual_seed(0)
    for (B, Fr, N, H) in [(2, 4, 49, 2), (1, 3, 100, 2), (2, 2, 223, 1), (8, 16, 196, 12)]:
        S = 1 + Fr * N
        for mode in ('space', 'time'):
            qkv = (torch.randn(B * S, 3 * H * 64, device=dev) * 0.7).bfloat16().requires_grad_(True)
            dO = torch.randn(B * S, H * 64, device=dev).bfloat16()
            res = {}
            for fused in (False, True):
                ops.FUSED_ATTN_BWD = ops.FUSED_ATTN_CLS = fused
                out = ops.divided_attention(qkv, B, Fr, N, H, mode)
                g, = torch.autograd.grad(out, qkv, dO)
                res[fused] = g.float()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 10
                outs = [ops.divided_attention(qkv, B, Fr, N, H, mode) for _ in range(n)]
                torch.cuda.synchronize()
                e0.record()
                for o in outs:
                    torch.autograd.grad(o, qkv, dO)
                e1.record()
                torch.cuda.synchronize()
                res[('us', fused)] = e0.elapsed_time(e1) / n * 1e3
            q32 = qkv.detach().float().requires_grad_(True)
            gr, = torch.autograd.grad(ref(q32, B, Fr, N, H, mode), q32, dO.float())
            sc = gr.abs().max().item()
            D = H * 64
            def parts(x): return [(x[:, i * D:(i + 1) * D]).abs().max().item() / sc for i in range(3)]
            print(f"B={B} F={Fr} N={N} H={H} {mode:5s}: pair {res[('us', False)]:7.1f} us, fused {res[('us', True)]:7.1f} us | "
                  f"rel err vs fp32 (dq,dk,dv): pair {['%.1e' % v for v in parts(res[False] - gr)]} fused {['%.1e' % v for v in parts(res[True] - gr)]} | "
                  f"fused vs pair {['%.1e' % v for v in parts(res[True] - res[False])]}", flush=True)


if __name__ == '__main__':
    main()
