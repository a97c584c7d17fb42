"""Block-level C entry points (csrc/egv_block.cpp) against the per-op composition they replace (run on the GPU box)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egovlpv2_amd import hipops as ops

dev = 'cuda'
torch.manual_seed(0)


def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()


def text_ref(hid, mask, P, B, L, H, enc=None, S=0):
    Wq, bq, Wk, bk, Wv, bv, Wao, bao, W1, b1, W2, b2, g0, be0, g1, be1 = P[:16]
    q, k, v = ops.linear(hid, Wq, bq), ops.linear(hid, Wk, bk), ops.linear(hid, Wv, bv)
    ctx = ops.plain_attention(q, k, v, B, H, L, L, 0.125, mask=mask)
    if enc is None:
        a = ops.linear(ctx, Wao, bao, res1=hid)
    else:
        Wcq, bcq, Wck, bck, Wcv, bcv, Wco, bco, alpha = P[16:25]
        a0 = ops.linear(ctx, Wao, bao)
        cq, ck, cv = ops.linear(a0, Wcq, bcq), ops.linear(enc, Wck, bck), ops.linear(enc, Wcv, bcv)
        cctx = ops.plain_attention(cq, ck, cv, B, H, L, S, 0.125, mask=None)
        a = ops.linear(cctx, Wco, bco, gate=alpha, res1=a0, res2=hid)
    a = ops.layernorm(a, g0, be0, 1e-5)
    f = ops.mlp(a, W1, b1, W2, b2, res=a)
    return ops.layernorm(f, g1, be1, 1e-5)


def video_ref(x, P, B, Fr, N, H, y=None, y_mask=None, L=0):
    Wt, bt, Wpt, bpt, Ws, bs, Wps, bps, W1, b1, W2, b2, g3, be3, g1, be1, g2, be2 = P[:18]
    S = 1 + Fr * N
    h, xs = ops.layernorm_skip(x, g3, be3, 1e-5)
    t_ctx = ops.divided_attention(ops.linear(h, Wt, bt), B, Fr, N, H, 'time')
    tr = ops.linear(t_ctx, Wpt, bpt, res1=xs)
    s_ctx = ops.divided_attention(ops.linear(ops.layernorm(tr, g1, be1, 1e-5), Ws, bs), B, Fr, N, H, 'space')
    if y is None:
        sr = ops.linear(s_ctx, Wps, bps, res1=xs)
    else:
        Wkv, bkv, Wq, bq, Wpi, bpi, gi, bei, alpha = P[18:27]
        D = x.shape[1]
        s = ops.linear(s_ctx, Wps, bps)
        kv = ops.linear(y, Wkv, bkv)
        hs, ss = ops.layernorm_skip(s, gi, bei, 1e-5)
        q = ops.linear(hs, Wq, bq)
        o = ops.plain_attention(q, kv[:, :D], kv[:, D:], B, H, S, L, 0.125, mask=y_mask)
        sr = ops.linear(o, Wpi, bpi, gate=alpha, res1=ss, res2=xs)
    h2, srs = ops.layernorm_skip(sr, g2, be2, 1e-5)
    return ops.mlp(h2, W1, b1, W2, b2, res=srs)


def mk(shape, scale=0.05):
    return (torch.randn(*shape, device=dev) * scale).requires_grad_(True)


for dtype in (torch.float32, torch.bfloat16):
    B, L, H, D, Hd, Fr, N = 2, 16, 12, 768, 3072, 4, 49
    S = 1 + Fr * N
    tp = []
    for _ in range(4):
        tp += [mk((D, D)), mk((D,))]
    tp += [mk((Hd, D)), mk((Hd,)), mk((D, Hd)), mk((D,)), mk((D,), 1.0), mk((D,)), mk((D,), 1.0), mk((D,))]
    tpf = tp + [mk((D, D)), mk((D,)), mk((D, D)), mk((D,)), mk((D, D)), mk((D,)), mk((D, D)), mk((D,)), mk((1,), 1.0)]
    vp = [mk((3 * D, D)), mk((3 * D,)), mk((D, D)), mk((D,)), mk((3 * D, D)), mk((3 * D,)), mk((D, D)), mk((D,)),
          mk((Hd, D)), mk((Hd,)), mk((D, Hd)), mk((D,)), mk((D,), 1.0), mk((D,)), mk((D,), 1.0), mk((D,)), mk((D,), 1.0), mk((D,))]
    vpf = vp + [mk((2 * D, D)), mk((2 * D,)), mk((D, D)), mk((D,)), mk((D, D)), mk((D,)), mk((D,), 1.0), mk((D,)), mk((1,), 1.0)]
    hid0 = torch.randn(B * L, D, device=dev)
    x0 = torch.randn(B * S, D, device=dev)
    m = torch.ones(B, L, device=dev); m[0, -3:] = 0
    mask = ((1 - m) * torch.finfo(torch.float32).min).contiguous()
    for name, fused in (('text', False), ('text_fused', True), ('video', False), ('video_fused', True)):
        outs = []
        for which in ('ref', 'blk'):
            P = {'text': tp, 'text_fused': tpf, 'video': vp, 'video_fused': vpf}[name]
            for p in P:
                p.grad = None
            hid = hid0.to(dtype).detach().requires_grad_(True)
            x = x0.to(dtype).detach().requires_grad_(True)
            if name.startswith('text'):
                enc = x if fused else None
                out = (text_ref(hid, mask, P, B, L, H, enc, S) if which == 'ref'
                       else ops.text_layer(hid, mask, P, B, L, H, Hd, 1e-5, enc=enc, S=S))
            else:
                y = hid if fused else None
                out = (video_ref(x, P, B, Fr, N, H, y, mask, L) if which == 'ref'
                       else ops.video_block(x, P, B, Fr, N, H, Hd, 1e-5, y=y, y_mask=mask if fused else None, L=L))
            gout = torch.randn(out.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(5)).to(dtype)
            out.backward(gout)
            torch.cuda.synchronize()
            outs.append((out.detach().clone(), hid.grad.clone() if hid.grad is not None else None, x.grad.clone() if x.grad is not None else None,
                         [p.grad.clone() for p in P]))
        (o0, h0, xg0, g0), (o1, h1, xg1, g1) = outs
        worst = max(rel(a, b) for a, b in zip(g1, g0))
        wi = max(range(len(g0)), key=lambda i: rel(g1[i], g0[i]))
        print(f"{dtype} {name:12s}: out {rel(o1, o0):.2e}  dhid {rel(h1, h0) if h0 is not None else -1:.2e}  dx {rel(xg1, xg0) if xg0 is not None else -1:.2e}  "
              f"worst param grad {worst:.2e} (#{wi})")
