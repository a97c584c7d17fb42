"""Op-level parity of every HIP entry point (through the C ABI via egovlpv2_amd.hipops) against plain
torch fp64 math on the same (dtype-rounded) inputs.  Tolerances (relative L2):
  fp32 storage: 2e-5 forward / 1e-4 backward (exact-fp32 MFMA, fp32 accumulation order differs from torch)
  bf16 storage: 6e-3 forward / 2e-2 backward (inputs are pre-rounded to bf16, so the error budget is the
                bf16 rounding of intermediates/outputs, 2^-9 per element)
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def _tol(dtype, bwd=False):
    if dtype == torch.float32:
        return 1e-4 if bwd else 2e-5
    return 2e-2 if bwd else 6e-3


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _rnd(shape, dtype, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    t = (torch.randn(shape, generator=g) * scale).to(dtype)
    return t


@pytest.fixture(scope='module')
def ops():
    from egovlpv2_amd import hipops
    return hipops


def gelu64(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(300, 200, 768), (128, 128, 64), (25, 2, 1536), (130, 1000, 264), (8, 4096, 768)])
def test_linear_forms(ops, dtype, shape):
    M, N, K = shape
    x = _rnd((M, K), dtype, 1.0, 1).cuda().requires_grad_(True)
    w = _rnd((N, K), torch.float32, 0.05, 2).cuda().requires_grad_(True)
    b = _rnd((N,), torch.float32, 0.5, 3).cuda().requires_grad_(True)
    r = _rnd((M, N), dtype, 1.0, 4).cuda().requires_grad_(True)
    y = ops.linear(x, w, b, res1=r)
    wq = w.detach().to(dtype).double().cpu()
    x64, r64 = x.detach().double().cpu().requires_grad_(True), r.detach().double().cpu().requires_grad_(True)
    w64, b64 = wq.clone().requires_grad_(True), b.detach().double().cpu().requires_grad_(True)
    y64 = x64 @ w64.t() + b64 + r64
    assert _rel(y, y64) < _tol(dtype)
    dy = _rnd((M, N), dtype, 1.0, 5)
    y.backward(dy.cuda())
    y64.backward(dy.double())
    assert _rel(x.grad, x64.grad) < _tol(dtype, True)
    assert _rel(w.grad, w64.grad) < _tol(dtype, True)
    assert _rel(b.grad, b64.grad) < _tol(dtype, True)
    assert _rel(r.grad, r64.grad) < 1e-6


@pytest.mark.parametrize('M,N,K,act,bias', [(8, 4096, 4096, 'relu', True), (8, 4096, 768, 'relu', False), (1, 4096, 4096, 'none', True),
                                            (16, 1000, 1024, 'tanh', True), (13, 520, 2304, 'none', False),
                                            (8, 3072, 768, 'gelu', True), (8, 768, 3072, 'none', True)])
def test_skinny_linear_bf16(ops, M, N, K, act, bias):
    """Linear over <= 16 rows (the projection heads, model.py:105-115, and the CLS rows of the last block of a video pass:
    gemm_skinny_kernel, one wave per four output columns, GELU with its saved pre-activation included; the weight gradient as an outer
    product, wgrad_small_m_kernel): forward and gradients against fp64 on the bf16 operands; a row's result does not depend on the other
    rows of the call (bitwise)"""
    dtype = torch.bfloat16
    x = _rnd((M, K), dtype, 1.0, 1).cuda().requires_grad_(True)
    w = _rnd((N, K), torch.float32, 0.05, 2).cuda().requires_grad_(True)
    b = _rnd((N,), torch.float32, 0.5, 3).cuda().requires_grad_(True) if bias else None
    y = ops.linear(x, w, b, act=act)
    x64 = x.detach().double().cpu().requires_grad_(True)
    w64 = w.detach().to(dtype).double().cpu().requires_grad_(True)
    z = x64 @ w64.t() + (b.detach().double().cpu() if bias else 0.0)
    y64 = {'relu': torch.relu, 'tanh': torch.tanh, 'none': (lambda t: t), 'gelu': gelu64}[act](z)
    assert _rel(y, y64) < _tol(dtype)
    dy = _rnd((M, N), dtype, 1.0, 5)
    y.backward(dy.cuda())
    y64.backward(dy.double())
    assert _rel(x.grad, x64.grad) < _tol(dtype, True) * 1.5
    assert _rel(w.grad, w64.grad) < _tol(dtype, True) * 1.5
    with torch.no_grad():
        one = ops.linear(x[M - 1:M].detach(), w.detach(), b.detach() if bias else None, act=act)
    assert torch.equal(one[0], y[M - 1].detach())


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('act', ['relu', 'tanh', 'gelu'])
def test_linear_act(ops, dtype, act):
    M, N, K = 70, 96, 128
    x = _rnd((M, K), dtype, 1.0, 1).cuda().requires_grad_(True)
    w = _rnd((N, K), torch.float32, 0.1, 2).cuda().requires_grad_(True)
    b = _rnd((N,), torch.float32, 0.5, 3).cuda().requires_grad_(True)
    y = ops.linear(x, w, b, act=act)
    x64 = x.detach().double().cpu().requires_grad_(True)
    w64 = w.detach().to(dtype).double().cpu().requires_grad_(True)
    b64 = b.detach().double().cpu().requires_grad_(True)
    z = x64 @ w64.t() + b64
    y64 = {'relu': torch.relu, 'tanh': torch.tanh, 'gelu': gelu64}[act](z)
    assert _rel(y, y64) < _tol(dtype)
    dy = _rnd((M, N), dtype, 1.0, 5)
    y.backward(dy.cuda())
    y64.backward(dy.double())
    assert _rel(x.grad, x64.grad) < _tol(dtype, True) * 1.5
    assert _rel(w.grad, w64.grad) < _tol(dtype, True) * 1.5
    assert _rel(b.grad, b64.grad) < _tol(dtype, True) * 1.5


@pytest.mark.parametrize('dtype', DTYPES)
def test_linear_gate_two_residuals(ops, dtype):
    M, N, K = 200, 768, 768
    x = _rnd((M, K), dtype, 1.0, 1).cuda().requires_grad_(True)
    w = _rnd((N, K), torch.float32, 0.05, 2).cuda().requires_grad_(True)
    b = _rnd((N,), torch.float32, 0.5, 3).cuda().requires_grad_(True)
    r1 = _rnd((M, N), dtype, 1.0, 4).cuda().requires_grad_(True)
    r2 = _rnd((M, N), dtype, 1.0, 6).cuda().requires_grad_(True)
    alpha = torch.tensor([0.37], device='cuda', requires_grad=True)
    y = ops.linear(x, w, b, gate=alpha, res1=r1, res2=r2)
    x64 = x.detach().double().cpu().requires_grad_(True)
    w64 = w.detach().to(dtype).double().cpu().requires_grad_(True)
    b64 = b.detach().double().cpu().requires_grad_(True)
    a64 = alpha.detach().double().cpu().requires_grad_(True)
    y64 = a64 * (x64 @ w64.t() + b64) + r1.detach().double().cpu() + r2.detach().double().cpu()
    assert _rel(y, y64) < _tol(dtype)
    dy = _rnd((M, N), dtype, 1.0, 5)
    y.backward(dy.cuda())
    y64.backward(dy.double())
    assert _rel(x.grad, x64.grad) < _tol(dtype, True)
    assert _rel(w.grad, w64.grad) < _tol(dtype, True)
    assert _rel(b.grad, b64.grad) < _tol(dtype, True)
    assert _rel(alpha.grad, a64.grad) < _tol(dtype, True)
    assert _rel(r1.grad, dy) < 1e-6 and _rel(r2.grad, dy) < 1e-6


@pytest.mark.parametrize('dtype', DTYPES)
def test_mlp(ops, dtype):
    M, D, H = 333, 768, 3072
    x = _rnd((M, D), dtype, 1.0, 1).cuda().requires_grad_(True)
    w1 = _rnd((H, D), torch.float32, 0.04, 2).cuda().requires_grad_(True)
    b1 = _rnd((H,), torch.float32, 0.2, 3).cuda().requires_grad_(True)
    w2 = _rnd((D, H), torch.float32, 0.02, 4).cuda().requires_grad_(True)
    b2 = _rnd((D,), torch.float32, 0.2, 5).cuda().requires_grad_(True)
    y = ops.mlp(x, w1, b1, w2, b2, res=x)
    p64 = [t.detach().double().cpu().requires_grad_(True) for t in (x, w1.detach().to(dtype), b1, w2.detach().to(dtype), b2)]
    x64, w164, b164, w264, b264 = p64
    h = gelu64(x64 @ w164.t() + b164)
    if dtype == torch.bfloat16:
        h = h + (h.detach().to(torch.bfloat16).double() - h.detach())      # the stored activation is bf16
    y64 = h @ w264.t() + b264 + x64
    assert _rel(y, y64) < _tol(dtype)
    dy = _rnd((M, D), dtype, 1.0, 6)
    y.backward(dy.cuda())
    y64.backward(dy.double())
    for a, bb in zip((x, w1, b1, w2, b2), p64):
        assert _rel(a.grad, bb.grad) < _tol(dtype, True) * 1.5


@pytest.mark.parametrize('M,N,K', [(777, 520, 128), (4000, 1032, 256), (25096, 3072, 768)])
def test_gelu_saved_derivative_epilogues_bf16(ops, M, N, K):
    """EGV_ACT_GELU_D (the video MLP in the bf16 mode): the forward GEMM returns gelu(xW^T + b) and saves gelu'(xW^T + b) where
    EGV_ACT_GELU saves the pre-activation; the data-gradient GEMM with dact = EGV_ACT_GELU_D multiplies by the saved tensor.  On all
    three GEMM kernels (generic, DMA ring, persistent): same output as the EGV_ACT_GELU call, saved tensor and gradient vs fp64."""
    from egovlpv2_amd import _lib as L
    x = _rnd((M, K), torch.bfloat16, 1.0, 1).cuda()
    w = _rnd((N, K), torch.float32, 0.06, 2).cuda().to(torch.bfloat16)
    b = _rnd((N,), torch.float32, 0.3, 3).cuda()
    y, pre = torch.empty(M, N, device='cuda', dtype=torch.bfloat16), torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    y2, dsave = torch.empty_like(y), torch.empty_like(y)
    kw = dict(M=M, N=N, K=K, lda=K, ldb=K, ldc=N)
    ops.gemm(x, w, y, bias=b, act=L.ACT_GELU, pre=pre, **kw)
    ops.gemm(x, w, y2, bias=b, act=L.ACT_GELU_D, pre=dsave, **kw)
    assert _rel(y2, y.double().cpu()) < 2e-3                          # (the compiler contracts x * Phi(x) differently beside the derivative: not bit-equal)
    rows = torch.cat([torch.arange(0, min(M, 300)), torch.arange(max(0, M - 300), M)]).unique()
    z = x[rows.cuda()].double().cpu() @ w.double().cpu().t() + b.double().cpu()
    dref = 0.5 * (1 + torch.erf(z / 2 ** 0.5)) + z * torch.exp(-0.5 * z * z) / (2 * torch.pi) ** 0.5
    assert _rel(dsave[rows.cuda()], dref) < 4e-3
    # data gradient: dx = (dy W2) * saved, against the EGV_ACT_GELU form on the saved pre-activation and against fp64
    dy = _rnd((M, K), torch.bfloat16, 1.0, 4).cuda()
    dz1, dz2 = torch.empty(M, N, device='cuda', dtype=torch.bfloat16), torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    kd = dict(M=M, N=N, K=K, lda=K, ldb=K, ldc=N)
    ops.gemm(dy, w, dz1, aux=pre, dact=L.ACT_GELU, **kd)
    ops.gemm(dy, w, dz2, aux=dsave, dact=L.ACT_GELU_D, **kd)
    g64 = (dy[rows.cuda()].double().cpu() @ w.double().cpu().t()) * dref
    assert _rel(dz2[rows.cuda()], g64) < 6e-3
    assert _rel(dz1[rows.cuda()], g64) < 6e-3


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('D', [768, 1024])
def test_layernorm(ops, dtype, D):
    M = 517
    x = _rnd((M, D), dtype, 2.0, 1).cuda().requires_grad_(True)
    g = (1 + _rnd((D,), torch.float32, 0.1, 2)).cuda().requires_grad_(True)
    b = _rnd((D,), torch.float32, 0.1, 3).cuda().requires_grad_(True)
    y = ops.layernorm(x, g, b, 1e-5)
    x64, g64, b64 = [t.detach().double().cpu().requires_grad_(True) for t in (x, g, b)]
    y64 = torch.nn.functional.layer_norm(x64, (D,), g64, b64, 1e-5)
    assert _rel(y, y64) < _tol(dtype)
    dy = _rnd((M, D), dtype, 1.0, 5)
    y.backward(dy.cuda())
    y64.backward(dy.double())
    assert _rel(x.grad, x64.grad) < _tol(dtype, True)
    assert _rel(g.grad, g64.grad) < _tol(dtype, True)
    assert _rel(b.grad, b64.grad) < _tol(dtype, True)


def _divided_ref(qkv, B, Fr, N, H, mode):
    """fp64 restatement of VarAttention's attention core (video_transformer.py:121-150)."""
    S = 1 + Fr * N
    dh = 64
    q, k, v = qkv.reshape(B, S, 3, H, dh).permute(2, 0, 3, 1, 4)
    q = q * dh ** -0.5
    cls = torch.softmax(q[:, :, :1] @ k.transpose(-1, -2), -1) @ v

    def grp(t):
        t = t[:, :, 1:].reshape(B, H, Fr, N, dh)
        return t if mode == 'space' else t.transpose(2, 3)
    qg, kg, vg = grp(q), grp(k), grp(v)
    G = qg.shape[2]
    kc = k[:, :, :1].unsqueeze(2).expand(B, H, G, 1, dh)
    vc = v[:, :, :1].unsqueeze(2).expand(B, H, G, 1, dh)
    kk, vv = torch.cat([kc, kg], 3), torch.cat([vc, vg], 3)
    og = torch.softmax(qg @ kk.transpose(-1, -2), -1) @ vv
    if mode != 'space':
        og = og.transpose(2, 3)
    out = torch.cat([cls, og.reshape(B, H, Fr * N, dh)], 2)
    return out.permute(0, 2, 1, 3).reshape(B * S, H * dh)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('mode,Fr,N', [('space', 3, 70), ('time', 5, 9), ('space', 2, 196), ('time', 16, 4), ('space', 2, 256), ('time', 32, 5), ('time', 20, 3)])
def test_divided_attention(ops, dtype, mode, Fr, N):
    B, H = 2, 3
    S = 1 + Fr * N
    qkv = _rnd((B * S, 3 * H * 64), dtype, 1.5, 7).cuda().requires_grad_(True)
    o = ops.divided_attention(qkv, B, Fr, N, H, mode)
    q64 = qkv.detach().double().cpu().requires_grad_(True)
    o64 = _divided_ref(q64, B, Fr, N, H, mode)
    assert _rel(o, o64) < _tol(dtype)
    do = _rnd((B * S, H * 64), dtype, 1.0, 8)
    o.backward(do.cuda())
    o64.backward(do.double())
    assert _rel(qkv.grad, q64.grad) < _tol(dtype, True)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('nq,nk,masked,nsplit', [(16, 16, True, 1), (32, 197, False, 1), (300, 32, True, 4), (5, 70, False, 1), (32, 3137, False, 1), (20, 1000, False, 1)])
def test_plain_attention(ops, dtype, nq, nk, masked, nsplit):
    B, H = 2, 3
    D = H * 64
    q = _rnd((B * nq, D), dtype, 1.0, 1).cuda().requires_grad_(True)
    kv = _rnd((B * nk, 2 * D), dtype, 1.0, 2).cuda().requires_grad_(True)
    mask = None
    if masked:
        m = torch.ones(B, nk)
        m[0, nk // 2:] = 0
        m[1, -1] = 0
        mask = ((1 - m) * torch.finfo(torch.float32).min).cuda()
    scale = 0.125
    o = ops.plain_attention(q, kv[:, :D], kv[:, D:], B, H, nq, nk, scale, mask=mask, dkv_nsplit=nsplit)
    q64 = q.detach().double().cpu().requires_grad_(True)
    kv64 = kv.detach().double().cpu().requires_grad_(True)
    qh = q64.reshape(B, nq, H, 64).transpose(1, 2)
    kh = kv64[:, :D].reshape(B, nk, H, 64).transpose(1, 2)
    vh = kv64[:, D:].reshape(B, nk, H, 64).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * scale
    if masked:
        s = s + mask.double().cpu().view(B, 1, 1, nk)
    o64 = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B * nq, D)
    assert _rel(o, o64) < _tol(dtype)
    do = _rnd((B * nq, D), dtype, 1.0, 3)
    o.backward(do.cuda())
    o64.backward(do.double())
    assert _rel(q.grad, q64.grad) < _tol(dtype, True)
    assert _rel(kv.grad, kv64.grad) < _tol(dtype, True)


@pytest.mark.parametrize('nq,nk,masked', [(3137, 32, True), (200, 20, True), (128, 32, False), (1000, 7, False)])
def test_many_queries_few_keys_attention_bf16(ops, nq, nk, masked, monkeypatch):
    """image -> text cross attention shapes (video_transformer.py:155-185) on the one-launch kernels of egv_attn_cross.hip, against fp64:
    full size per sample (3137 queries, 32 keys, text mask), ragged query and key counts, no mask."""
    B, H = 2, 3
    D = H * 64
    q = _rnd((B * nq, D), torch.bfloat16, 1.0, 11).cuda().requires_grad_(True)
    kv = _rnd((B * nk, 2 * D), torch.bfloat16, 1.0, 12).cuda().requires_grad_(True)
    mask = None
    if masked:
        m = torch.ones(B, nk)
        m[0, nk // 2:] = 0
        m[1, -1] = 0
        mask = ((1 - m) * -10000.0).cuda()
    o = ops.plain_attention(q, kv[:, :D], kv[:, D:], B, H, nq, nk, 0.125, mask=mask)
    q64 = q.detach().double().cpu().requires_grad_(True)
    kv64 = kv.detach().double().cpu().requires_grad_(True)
    qh = q64.reshape(B, nq, H, 64).transpose(1, 2)
    kh = kv64[:, :D].reshape(B, nk, H, 64).transpose(1, 2)
    vh = kv64[:, D:].reshape(B, nk, H, 64).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * 0.125
    if masked:
        s = s + mask.double().cpu().view(B, 1, 1, nk)
    o64 = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B * nq, D)
    assert _rel(o, o64) < _tol(torch.bfloat16)
    do = _rnd((B * nq, D), torch.bfloat16, 1.0, 13)
    o.backward(do.cuda())
    o64.backward(do.double())
    assert _rel(q.grad, q64.grad) < _tol(torch.bfloat16, True)
    assert _rel(kv.grad, kv64.grad) < _tol(torch.bfloat16, True)
    # rows of one sample must not leak into the next: the last query rows of sample 0 and the first of sample 1, row by row
    for r in (nq - 1, nq, B * nq - 1):
        assert _rel(o[r], o64[r]) < 2 * _tol(torch.bfloat16)
        assert _rel(q.grad[r], q64.grad[r]) < 3 * _tol(torch.bfloat16, True)
    # deterministic: a second backward pass gives the same bits (fixed-order partial sums, no atomics)
    g1 = kv.grad.clone()
    kv.grad = None
    q.grad = None
    ops.plain_attention(q, kv[:, :D], kv[:, D:], B, H, nq, nk, 0.125, mask=mask).backward(do.cuda())
    assert torch.equal(g1, kv.grad)


def _attn_keep_mult(ops, dtype, B, H, nq, nk, p, seed):
    """Recover the kernels' dropout multiplier M[b,h,q,k] (0 or 1/(1-p)): with Q = K = 0 the probabilities are uniform, and
    with V = identity on a 64-key chunk O[q, d] = M[q, chunk*64 + d] / nk.  The mask is a function of indices only."""
    D = H * 64
    z = torch.zeros(B * nq, D, dtype=dtype, device='cuda')
    zk = torch.zeros(B * nk, D, dtype=dtype, device='cuda')
    M = torch.zeros(B, H, nq, nk, dtype=torch.float64)
    for c0 in range(0, nk, 64):
        n = min(64, nk - c0)
        v = torch.zeros(B, nk, H, 64)
        v[:, c0:c0 + n, :, :n] = torch.eye(n).view(1, n, 1, n)
        o = ops.plain_attention(z, zk, v.reshape(B * nk, D).to(dtype).cuda(), B, H, nq, nk, 1.0, drop_p=p, drop_seed=seed)
        M[:, :, :, c0:c0 + n] = (o.double().cpu().reshape(B, nq, H, 64).transpose(1, 2) * nk)[..., :n]
    return M


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('nq,nk,masked,nsplit', [(32, 32, True, 1), (32, 300, False, 1), (300, 32, True, 4), (32, 1100, False, 1), (9, 600, False, 1)])
def test_plain_attention_dropout(ops, dtype, nq, nk, masked, nsplit):
    """roberta.py:313 attention-probability dropout inside the kernels: forward and all three gradients against torch
    math using the very mask the kernels generate (recovered through the forward itself)."""
    B, H, p, seed = 2, 2, 0.25, 1234
    D = H * 64
    M = _attn_keep_mult(ops, dtype, B, H, nq, nk, p, seed)
    inv = 1.0 / (1.0 - p)
    on = (M - inv).abs() < 0.02 * inv
    assert bool((on | (M.abs() < 1e-6)).all()), "multiplier is 0 or 1/(1-p)"
    keep = on.double().mean().item()
    assert abs(keep - (1 - p)) < 0.02, keep
    M = on.double() * inv
    assert (M[0] != M[1]).any() and (M[:, 0] != M[:, 1]).any(), "mask must differ per batch row and per head"
    M2 = _attn_keep_mult(ops, dtype, B, H, nq, nk, p, seed + 1)
    assert ((M2 > 0) != on).double().mean().item() > 0.2, "mask must depend on the seed"

    q = _rnd((B * nq, D), dtype, 1.0, 1).cuda().requires_grad_(True)
    kv = _rnd((B * nk, 2 * D), dtype, 1.0, 2).cuda().requires_grad_(True)
    mask = None
    if masked:
        m = torch.ones(B, nk)
        m[0, nk // 2:] = 0
        mask = ((1 - m) * torch.finfo(torch.float32).min).cuda()
    scale = 0.125
    o = ops.plain_attention(q, kv[:, :D], kv[:, D:], B, H, nq, nk, scale, mask=mask, dkv_nsplit=nsplit, drop_p=p, drop_seed=seed)
    q64 = q.detach().double().cpu().requires_grad_(True)
    kv64 = kv.detach().double().cpu().requires_grad_(True)
    qh = q64.reshape(B, nq, H, 64).transpose(1, 2)
    kh = kv64[:, :D].reshape(B, nk, H, 64).transpose(1, 2)
    vh = kv64[:, D:].reshape(B, nk, H, 64).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * scale
    if masked:
        s = s + mask.double().cpu().view(B, 1, 1, nk)
    o64 = ((torch.softmax(s, -1) * M) @ vh).transpose(1, 2).reshape(B * nq, D)
    assert _rel(o, o64) < _tol(dtype)
    do = _rnd((B * nq, D), dtype, 1.0, 3)
    o.backward(do.cuda())
    o64.backward(do.double())
    assert _rel(q.grad, q64.grad) < _tol(dtype, True)
    assert _rel(kv.grad, kv64.grad) < _tol(dtype, True)


@pytest.mark.parametrize('dtype', DTYPES)
def test_dropout_add(ops, dtype):
    """dense -> dropout -> (+ residuals) of RobertaSelfOutput / RobertaOutput (roberta.py:342, :422, :486-488)."""
    n, p = (257, 768), 0.1
    x = _rnd(n, dtype, 1.0, 1).cuda().requires_grad_(True)
    r1 = _rnd(n, dtype, 1.0, 2).cuda().requires_grad_(True)
    r2 = _rnd(n, dtype, 1.0, 3).cuda().requires_grad_(True)
    y = ops.dropout_add(x, p, 77, r1=r1, r2=r2)
    y0 = ops.dropout_add(x, p, 77)
    xd = x.detach().double()
    keep = (y0.detach().double().abs() > 0) | (xd == 0)
    rate = keep.double().mean().item()
    assert abs(rate - (1 - p)) < 0.01, rate
    ref0 = torch.where(keep, xd / (1 - p), torch.zeros_like(xd))
    assert _rel(y0, ref0) < _tol(dtype)
    assert _rel(y, ref0 + r1.detach().double() + r2.detach().double()) < _tol(dtype)
    assert torch.equal(y0, ops.dropout_add(x, p, 77)), "same seed, same mask"
    other = (ops.dropout_add(x, p, 78).detach().abs() > 0)
    assert (other != keep).double().mean().item() > 0.05, "mask must depend on the seed"
    dy = _rnd(n, dtype, 1.0, 4).cuda()
    y.backward(dy)
    assert _rel(x.grad, torch.where(keep, dy.double() / (1 - p), torch.zeros_like(xd))) < _tol(dtype)
    assert torch.equal(r1.grad, dy) and torch.equal(r2.grad, dy)


@pytest.mark.parametrize('dtype', DTYPES)
def test_patch_tokens(ops, dtype):
    B, Fr, R, P, D = 2, 3, 64, 16, 128
    N = (R // P) ** 2
    video = _rnd((B, Fr, 3, R, R), torch.float32, 1.0, 1).cuda()
    w = _rnd((D, 3, P, P), torch.float32, 0.05, 2).cuda().requires_grad_(True)
    b = _rnd((D,), torch.float32, 0.1, 3).cuda().requires_grad_(True)
    cls = _rnd((1, 1, D), torch.float32, 0.5, 4).cuda().requires_grad_(True)
    pos = _rnd((1, 1 + N, D), torch.float32, 0.5, 5).cuda().requires_grad_(True)
    tem = _rnd((1, Fr, D), torch.float32, 0.5, 6).cuda().requires_grad_(True)
    y = ops.patch_tokens(video, w, b, cls, pos, tem, dtype)
    v64 = video.to(dtype).double().cpu()
    w64 = w.detach().to(dtype).double().cpu().requires_grad_(True)
    b64, c64, p64, t64 = [t.detach().double().cpu().requires_grad_(True) for t in (b, cls, pos, tem)]
    e = torch.nn.functional.conv2d(v64.reshape(B * Fr, 3, R, R), w64, b64, stride=P).flatten(2).transpose(2, 1).reshape(B, Fr * N, D)
    body = p64[:, 1:].unsqueeze(1) + t64.unsqueeze(2)
    y64 = torch.cat([(c64 + p64[:, :1]).expand(B, 1, D), e + body.reshape(1, Fr * N, D)], 1)
    assert _rel(y, y64) < _tol(dtype)
    dy = _rnd((B, 1 + Fr * N, D), dtype, 1.0, 7)
    y.backward(dy.cuda())
    y64.backward(dy.double())
    for a, bb in ((w, w64), (b, b64), (cls, c64), (pos, p64), (tem, t64)):
        assert _rel(a.grad, bb.grad) < _tol(dtype, True)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('B,L', [(3, 12), (600, 32)])               # (600, 32): 19 200 tokens = two chunks of the table-gradient kernel
def test_text_embed(ops, dtype, B, L):
    D, V = 128, 500
    ids = torch.randint(3, V, (B, L), generator=torch.Generator().manual_seed(1))
    ids[:, 0] = 0
    ids[0, 7:] = 1
    ids[1, 10:] = 1
    ids[B - 1, L - 3:] = 1
    word = _rnd((V, D), torch.float32, 1.0, 2).cuda().requires_grad_(True)
    pos = _rnd((40, D), torch.float32, 1.0, 3).cuda().requires_grad_(True)
    typ = _rnd((1, D), torch.float32, 1.0, 4).cuda().requires_grad_(True)
    y = ops.text_embed(ids.cuda(), word, pos, typ, 1, dtype)
    m = ids.ne(1).long()
    pid = torch.cumsum(m, 1) * m + 1
    w64, p64, t64 = [t.detach().double().cpu().requires_grad_(True) for t in (word, pos, typ)]
    y64 = (torch.nn.functional.embedding(ids, w64, padding_idx=1) + t64[0] + torch.nn.functional.embedding(pid, p64, padding_idx=1)).reshape(B * L, D)
    assert _rel(y, y64) < _tol(dtype)
    dy = _rnd((B * L, D), dtype, 1.0, 5)
    y.backward(dy.cuda())
    y64.backward(dy.double())
    tolg = 1e-5 if B * L < 1000 else 2e-5                                  # fp32 sums over up to B rows
    assert _rel(word.grad, w64.grad) < tolg
    assert _rel(pos.grad, p64.grad) < tolg
    assert _rel(typ.grad, t64.grad) < tolg


@pytest.mark.parametrize('dtype', DTYPES)
def test_vocab_linear_and_ce(ops, dtype):
    R, K, V = 40, 128, 1003
    x = _rnd((R, K), dtype, 1.0, 1).cuda().requires_grad_(True)
    w = _rnd((V, K), torch.float32, 0.2, 2).cuda().requires_grad_(True)
    b = _rnd((V,), torch.float32, 0.5, 3).cuda().requires_grad_(True)
    labels = torch.randint(0, V, (R,), generator=torch.Generator().manual_seed(4))
    labels[::3] = -100
    logits = ops.vocab_linear(x, w, b, V)
    assert logits.shape[1] % 128 == 0
    total = ops.cross_entropy_sum(logits, labels.cuda(), V, -100)
    x64 = x.detach().double().cpu().requires_grad_(True)
    w64 = w.detach().to(dtype).double().cpu().requires_grad_(True)
    b64 = b.detach().double().cpu().requires_grad_(True)
    lg = x64 @ w64.t() + b64
    if dtype == torch.bfloat16:
        lg = lg + (lg.detach().to(torch.bfloat16).double() - lg.detach())
    t64 = torch.nn.functional.cross_entropy(lg, labels, ignore_index=-100, reduction='sum')
    assert abs(total.item() - t64.item()) < (2e-5 if dtype == torch.float32 else 4e-3) * abs(t64.item())
    (total * 0.5).backward()
    (t64 * 0.5).backward()
    assert _rel(x.grad, x64.grad) < _tol(dtype, True)
    assert _rel(w.grad, w64.grad) < _tol(dtype, True)
    assert _rel(b.grad, b64.grad) < _tol(dtype, True)


def test_sim_matrix_and_egonce(ops):
    n, d = 24, 512
    a = _rnd((n, d), torch.float32, 1.0, 1).cuda().requires_grad_(True)
    b = _rnd((n, d), torch.float32, 1.0, 2).cuda().requires_grad_(True)
    nv = (torch.rand(n, 30, generator=torch.Generator().manual_seed(3)) < 0.2).float()
    vv = (torch.rand(n, 20, generator=torch.Generator().manual_seed(4)) < 0.2).float()
    sim = ops.sim_matrix_f32(a, b)
    sv = ops.sim_matrix_f32(vv.cuda(), vv.cuda())
    sn = ops.sim_matrix_f32(nv.cuda(), nv.cuda())
    loss, mb = ops.egonce(sim, sv, sn, 0.05, True, True)

    def sm(x, y, eps=1e-8):
        return (x / x.norm(dim=1, keepdim=True).clamp_min(eps)) @ (y / y.norm(dim=1, keepdim=True).clamp_min(eps)).t()
    a64, b64 = a.detach().double().cpu().requires_grad_(True), b.detach().double().cpu().requires_grad_(True)
    s64 = sm(a64, b64)
    sv64, sn64 = sm(vv.double(), vv.double()), sm(nv.double(), nv.double())
    mask = (sv64 * sn64 + torch.eye(n, dtype=torch.float64)) > 0
    i_sm, j_sm = torch.softmax(s64 / 0.05, 1), torch.softmax(s64.t() / 0.05, 1)
    l64 = -torch.log((i_sm * mask).sum(1)).mean() - torch.log((j_sm * mask).sum(1)).mean()
    assert _rel(sim, s64) < 2e-5
    assert torch.equal(mb.cpu(), mask)
    assert abs(loss.item() - l64.item()) < 2e-5 * abs(l64.item())
    loss.backward()
    l64.backward()
    assert _rel(a.grad, a64.grad) < 2e-4
    assert _rel(b.grad, b64.grad) < 2e-4


@pytest.mark.parametrize('n,m,d', [(5, 9, 100), (8, 8, 4096), (64, 64, 256), (70, 70, 256)])
def test_sim_matrix_rectangular_and_both_paths(ops, n, m, d):
    """sim_matrix (model.py:576-584) forward and both gradients for rectangular and ragged shapes: the one-wave-per-entry kernels
    (up to 64 x 64 entries: the EgoNCE matrices) and the GEMM path above that, against fp64"""
    a = _rnd((n, d), torch.float32, 1.0, 11).cuda().requires_grad_(True)
    b = _rnd((m, d), torch.float32, 1.0, 12).cuda().requires_grad_(True)
    w = _rnd((n, m), torch.float32, 1.0, 13)
    sim = ops.sim_matrix_f32(a, b)
    (sim * w.cuda()).sum().backward()
    a64, b64 = a.detach().double().cpu().requires_grad_(True), b.detach().double().cpu().requires_grad_(True)
    s64 = (a64 / a64.norm(dim=1, keepdim=True)) @ (b64 / b64.norm(dim=1, keepdim=True)).t()
    (s64 * w.double()).sum().backward()
    assert _rel(sim, s64) < 2e-5
    assert _rel(a.grad, a64.grad) < 2e-4
    assert _rel(b.grad, b64.grad) < 2e-4


def test_cpu_tensor_is_refused(ops):
    with pytest.raises(RuntimeError):
        ops.linear(torch.randn(4, 8), torch.randn(8, 8))


@pytest.mark.parametrize('shape', [(1000, 768, 768), (700, 2304, 768), (513, 768, 3072), (3137, 3072, 768), (300, 768, 128)])
def test_linear_large_tiles_bf16(ops, shape):
    """shapes that take the 256-row DMA-staged kernel (egv_gemm2.hip): fwd, dgrad on W^T, wgrad with split reduction"""
    M, N, K = shape
    dtype = torch.bfloat16
    x = _rnd((M, K), dtype, 1.0, 1).cuda().requires_grad_(True)
    w = _rnd((N, K), torch.float32, 0.05, 2).cuda().requires_grad_(True)
    b = _rnd((N,), torch.float32, 0.5, 3).cuda().requires_grad_(True)
    r = _rnd((M, N), dtype, 1.0, 4).cuda()
    y = ops.linear(x, w, b, res1=r)
    wq = w.detach().to(dtype).double().cpu()
    x64 = x.detach().double().cpu().requires_grad_(True)
    w64, b64 = wq.clone().requires_grad_(True), b.detach().double().cpu().requires_grad_(True)
    y64 = x64 @ w64.t() + b64 + r.double().cpu()
    assert _rel(y, y64) < _tol(dtype)
    dy = _rnd((M, N), dtype, 1.0, 5)
    y.backward(dy.cuda())
    y64.backward(dy.double())
    assert _rel(x.grad, x64.grad) < _tol(dtype, True)
    assert _rel(w.grad, w64.grad) < _tol(dtype, True)
    assert _rel(b.grad, b64.grad) < _tol(dtype, True)


# ---- BASELINE.json full sizes (B=8, 16 x 224^2: M = 8 * 3137 = 25096 token rows), checked on samples ------------------------
FULL_M = 8 * (1 + 16 * 196)


@pytest.mark.parametrize('N,K,epi', [(2304, 768, 'bias'), (768, 768, 'res'), (3072, 768, 'gelu'), (768, 3072, 'res')])
def test_full_size_linear_sampled_rows_bf16(ops, N, K, epi):
    """The four hot Linear shapes at the full token count: forward, input gradient and weight gradient.  Rows are sampled
    (first / last rows of 256-row tiles, the ragged last tile of 8 rows, random ones) and compared with fp64 math on the same
    bf16 inputs; sampled weight-gradient entries (each a reduction over all rows) and the bias gradient against fp64."""
    M = FULL_M
    x = _rnd((M, K), torch.bfloat16, 1.0, 1).cuda().requires_grad_(True)
    w = _rnd((N, K), torch.float32, 0.05, 2).cuda().requires_grad_(True)
    b = _rnd((N,), torch.float32, 0.5, 3).cuda().requires_grad_(True)
    res = _rnd((M, N), torch.bfloat16, 1.0, 4).cuda() if epi == 'res' else None
    y = ops.linear(x, w, b, act='gelu' if epi == 'gelu' else 'none', res1=res)
    rows = torch.tensor(sorted({0, 1, 255, 256, 4095, 12543, 12544, M - 9, M - 8, M - 1} |
                               set(torch.randint(0, M, (40,), generator=torch.Generator().manual_seed(5)).tolist())))
    xs = x.detach()[rows.cuda()].double().cpu()
    w64 = w.detach().to(torch.bfloat16).double().cpu()
    z = xs @ w64.t() + b.detach().double().cpu()
    ref = gelu64(z) if epi == 'gelu' else z
    if res is not None:
        ref = ref + res[rows.cuda()].double().cpu()
    assert _rel(y.detach()[rows.cuda()], ref) < 6e-3
    dy = _rnd((M, N), torch.bfloat16, 1.0, 6).cuda()
    y.backward(dy)
    dys = dy[rows.cuda()].double().cpu()
    dz = dys * (0.5 * (1 + torch.erf(z / math.sqrt(2))) + z * torch.exp(-0.5 * z * z) / math.sqrt(2 * math.pi)) if epi == 'gelu' else dys
    if epi == 'gelu':                       # the kernel rounds dz to bf16 before the two backward GEMMs
        dz = dz.to(torch.bfloat16).double()
    assert _rel(x.grad[rows.cuda()], dz @ w64) < 2e-2
    # weight gradient on sampled entries (each one reduces over all 25096 rows) and the whole bias gradient, fp64 reference
    if epi != 'gelu':
        g = torch.Generator().manual_seed(7)
        ns = torch.cat([torch.tensor([0, N - 1, 127, 128]), torch.randint(0, N, (60,), generator=g)])
        ks = torch.cat([torch.tensor([0, K - 1, 31, 32]), torch.randint(0, K, (60,), generator=g)])
        want = (dy[:, ns.cuda()].double() * x.detach()[:, ks.cuda()].double()).sum(0).cpu()
        got = w.grad[ns.cuda(), ks.cuda()].double().cpu()
        assert _rel(got, want) < 1e-4
        assert _rel(b.grad, dy.double().sum(0)) < 1e-4


def test_full_size_divided_attention_sampled_problems_bf16(ops):
    """Space and time attention at B=8, 16 frames, 196 patches, 12 heads: sampled (sample, frame/patch, head) problems against
    fp64 softmax attention on the same bf16 qkv, plus the frame-permutation equivariance of space attention (patch rows of a
    frame only see that frame and the CLS row, so permuting whole frames permutes the patch outputs)."""
    B, Fr, N, H = 8, 16, 196, 12
    S, D = 1 + Fr * N, H * 64
    qkv = _rnd((B * S, 3 * D), torch.bfloat16, 1.0, 1).cuda()
    q64 = qkv.double().cpu().reshape(B, S, 3, H, 64)
    gen = torch.Generator().manual_seed(2)
    for mode in ('space', 'time'):
        o = ops.divided_attention(qkv, B, Fr, N, H, mode).double().cpu().reshape(B, S, H, 64)
        for _ in range(6):
            b, h = int(torch.randint(0, B, (1,), generator=gen)), int(torch.randint(0, H, (1,), generator=gen))
            if mode == 'space':
                f = int(torch.randint(0, Fr, (1,), generator=gen))
                rows = torch.arange(1 + f * N, 1 + (f + 1) * N)
            else:
                n = int(torch.randint(0, N, (1,), generator=gen))
                rows = 1 + n + N * torch.arange(Fr)
            keys = torch.cat([torch.zeros(1, dtype=torch.long), rows])
            q, k, v = q64[b, rows, 0, h], q64[b, keys, 1, h], q64[b, keys, 2, h]
            ref = torch.softmax(q @ k.t() * 0.125, -1) @ v
            assert _rel(o[b, rows, h], ref) < 6e-3, (mode, b, h)
        b, h = 3, 7                                            # the CLS query sees all S keys
        ref = torch.softmax(q64[b, :1, 0, h] @ q64[b, :, 1, h].t() * 0.125, -1) @ q64[b, :, 2, h]
        assert _rel(o[b, :1, h], ref) < 6e-3
    # equivariance: swap frames 2 and 9 of every sample
    perm = torch.arange(S)
    a, c = torch.arange(1 + 2 * N, 1 + 3 * N), torch.arange(1 + 9 * N, 1 + 10 * N)
    perm[a], perm[c] = c, a
    qp = qkv.reshape(B, S, 3 * D)[:, perm.cuda()].reshape(B * S, 3 * D).contiguous()
    o1 = ops.divided_attention(qkv, B, Fr, N, H, 'space').reshape(B, S, D)
    o2 = ops.divided_attention(qp, B, Fr, N, H, 'space').reshape(B, S, D)
    assert torch.equal(o2[:, 1:], o1[:, perm.cuda()][:, 1:]), "patch rows: bitwise equal under a frame permutation"
    assert _rel(o2[:, 0], o1[:, 0]) < 6e-3                     # CLS row: same value, different summation order


def test_fused_attention_backward_matches_kernel_pair(ops):
    """dQ / dK / dV of the space attention from the one-pass kernel (attn_bwd_fused_kernel) against the dQ + dK/dV kernel pair on
    the same bf16 inputs (same bf16 roundings of P and dS, different summation order), run-to-run bitwise reproducibility (the
    cross-wave dQ sum has a fixed order, no atomics), and the CLS row's gradients from the kernel's per-group partials against the
    one-query / one-key launches."""
    for (B, Fr, N, H) in [(2, 3, 70, 3), (1, 2, 196, 2), (2, 2, 223, 1), (1, 1, 65, 1)]:
        S = 1 + Fr * N
        qkv = _rnd((B * S, 3 * H * 64), torch.bfloat16, 1.0, 11).cuda().requires_grad_(True)
        do = _rnd((B * S, H * 64), torch.bfloat16, 1.0, 12).cuda()
        grads = {}
        try:
            for fused, cls in ((False, False), (True, True), (True, True), (True, False)):
                ops.FUSED_ATTN_BWD = ops.FUSED_ATTN_CLS = True            # one forward for all (its CLS row may come from the group launch)
                o = ops.divided_attention(qkv, B, Fr, N, H, 'space')
                ops.FUSED_ATTN_BWD, ops.FUSED_ATTN_CLS = fused, cls
                g, = torch.autograd.grad(o, qkv, do)
                grads.setdefault((fused, cls), []).append(g)
        finally:
            ops.FUSED_ATTN_BWD = ops.FUSED_ATTN_CLS = True
        assert torch.equal(grads[(True, True)][0], grads[(True, True)][1]), (B, Fr, N, H)
        ref = grads[(False, False)][0].double().cpu()
        assert _rel(grads[(True, True)][0], ref) < 4e-3, (B, Fr, N, H)
        assert _rel(grads[(True, False)][0], ref) < 4e-3, (B, Fr, N, H)
        # the CLS row (first row of every sample): its dQ / dK / dV come from the per-group partials of the one-pass kernel
        cls_rows = torch.arange(B) * S
        assert _rel(grads[(True, True)][0][cls_rows.cuda()], ref[cls_rows]) < 4e-3, (B, Fr, N, H)
        patch = torch.ones(B * S, dtype=torch.bool); patch[cls_rows] = False
        assert torch.equal(grads[(True, True)][0][patch.cuda()], grads[(True, False)][0][patch.cuda()]), "patch rows do not depend on who writes the CLS row"


def test_prepare_weights_matches_per_tensor_casts(ops):
    """egv_cast_weights (one launch for all Linear weights of a step) must produce exactly the copies the lazy per-tensor
    path makes, and seed both caches; non-qualifying tensors are left alone."""
    shapes = [(768, 768), (2304, 768), (768, 3072), (4096, 768), (64, 64), (130, 768)]
    ws = [_rnd(s, torch.float32, 0.3, i).cuda() for i, s in enumerate(shapes)]
    ops.invalidate_weight_cache()
    lazy = [(ops.compute_weight(w, torch.bfloat16).clone(), ops.compute_weight_t(w, torch.bfloat16)) for w in ws]
    lazy = [(a, None if t is None else t.clone()) for a, t in lazy]
    ops.invalidate_weight_cache()
    ops.prepare_weights(ws, torch.bfloat16)
    for w, (a, t), s in zip(ws, lazy, shapes):
        assert torch.equal(ops.compute_weight(w, torch.bfloat16), a), s
        if t is not None:
            assert torch.equal(ops.compute_weight_t(w, torch.bfloat16), t), s
    assert torch.equal(lazy[0][0], ws[0].to(torch.bfloat16))
    ws[1].add_(1.0)                                            # a new parameter version must be re-cast
    ops.prepare_weights(ws, torch.bfloat16)
    assert torch.equal(ops.compute_weight(ws[1], torch.bfloat16), ws[1].to(torch.bfloat16))
    assert torch.equal(ops.compute_weight_t(ws[1], torch.bfloat16), ws[1].to(torch.bfloat16).t().contiguous())


@pytest.mark.parametrize('dtype', DTYPES)
def test_patch_tokens_from_uint8_clips(ops, dtype):
    """uint8 clips through egv_im2col_u8 == the reference's host-side ToTensor + Normalize (data_loader/transforms.py:17-19)
    followed by the float path."""
    B, Fr, H, P, D = 2, 3, 32, 16, 64
    g = torch.Generator().manual_seed(1)
    u8 = torch.randint(0, 256, (B, Fr, 3, H, H), generator=g, dtype=torch.uint8)
    mean = torch.tensor(ops.IMAGENET_MEAN).view(1, 1, 3, 1, 1)
    std = torch.tensor(ops.IMAGENET_STD).view(1, 1, 3, 1, 1)
    ref_in = (u8.float() / 255.0 - mean) / std
    N = (H // P) ** 2
    w = _rnd((D, 3, P, P), torch.float32, 0.05, 2).cuda()
    b = _rnd((D,), torch.float32, 0.1, 3).cuda()
    cls = _rnd((1, 1, D), torch.float32, 1.0, 4).cuda()
    pos = _rnd((1, 1 + N, D), torch.float32, 1.0, 5).cuda()
    tem = _rnd((1, Fr, D), torch.float32, 1.0, 6).cuda()
    a = ops.patch_tokens(u8.cuda(), w, b, cls, pos, tem, dtype)
    r = ops.patch_tokens(ref_in.cuda(), w, b, cls, pos, tem, dtype)
    assert _rel(a, r) < (1e-6 if dtype == torch.float32 else 4e-3)


# ---- round 2: persistent ping-pong GEMM / block-level entry points -------------------------------------------------------------
@pytest.mark.parametrize('N,K,kind', [(2304, 768, 'bias'), (3072, 768, 'gelu'), (768, 3072, 'res'), (768, 2304, 'plain'), (768, 768, 'plain'), (768, 768, 'gate_res2')])
def test_persistent_gemm_bitwise_equals_ring_gemm(ops, N, K, kind):
    """The kernel choice depends on the grid size (persistent ping-pong kernel from 64 tiles of 256x256 up -- with 192-row tiles
    for the N = 768 shapes at full M -- DMA-ring kernels below): all accumulate in the same K order and add the bias after the sum, so the first rows of a full-size call must be
    bit-identical to a call on those rows alone (what keeps FrozenInTime batch-composition independent in bf16)."""
    M, m = FULL_M, 1024
    x = _rnd((M, K), torch.bfloat16, 1.0, 11).cuda()
    w = _rnd((N, K), torch.float32, 0.05, 12).cuda()
    b = _rnd((N,), torch.float32, 0.5, 13).cuda() if kind != 'plain' else None
    res = _rnd((M, N), torch.bfloat16, 1.0, 14).cuda() if kind in ('res', 'gate_res2') else None
    res2 = _rnd((M, N), torch.bfloat16, 1.0, 15).cuda() if kind == 'gate_res2' else None
    gate = torch.tensor([0.37], device='cuda') if kind == 'gate_res2' else None
    act = 'gelu' if kind == 'gelu' else 'none'
    big = ops.linear(x, w, b, act=act, res1=res, res2=res2, gate=gate)
    small = ops.linear(x[:m].contiguous(), w, b, act=act, res1=None if res is None else res[:m].contiguous(),
                       res2=None if res2 is None else res2[:m].contiguous(), gate=gate)
    assert torch.equal(big[:m], small)
    # and against fp64 on sampled rows of the last (ragged, 8-row) tile
    rows = torch.arange(M - 8, M)
    z = x[rows.cuda()].double().cpu() @ w.to(torch.bfloat16).double().cpu().t()
    if b is not None:
        z = z + b.double().cpu()
    ref = gelu64(z) if kind == 'gelu' else z
    if gate is not None:
        ref = ref * float(gate)
    if res is not None:
        ref = ref + res[rows.cuda()].double().cpu()
    if res2 is not None:
        ref = ref + res2[rows.cuda()].double().cpu()
    assert _rel(big[rows.cuda()], ref) < 6e-3


@pytest.mark.parametrize('N,K,bias', [(768, 768, False), (2304, 768, False), (768, 768, True)])
def test_persistent_gemm_gated_plain_kind_bitwise_equals_ring_gemm(ops, N, K, bias):
    """The gate WITHOUT a residual or a saved pre-gate value -- egv_block.cpp's data gradient of the gated image-to-text projection,
    d_o = alpha (d_sr W_proj_i2t) -- takes the persistent kernel's plain kind (192- and 256-row tiles), whose spread epilogue (round 6)
    has to apply it: bitwise against the ring kernel on a slice of the rows, fp64 on the ragged last rows.  (Round 6's first spread
    epilogue dropped the factor; no per-op test saw it -- the end-to-end gradient check of test_base_f16_vs_golden did.)"""
    M, m = FULL_M, 1024
    x = _rnd((M, K), torch.bfloat16, 1.0, 31).cuda()
    w = _rnd((N, K), torch.float32, 0.05, 32).to(torch.bfloat16).cuda()
    b = _rnd((N,), torch.float32, 0.5, 33).cuda() if bias else None
    gate = torch.tensor([0.37], device='cuda')
    big = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
    small = torch.empty(m, N, dtype=torch.bfloat16, device='cuda')
    ops.gemm(x, w, big, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=b, gate=gate)
    ops.gemm(x[:m].contiguous(), w, small, M=m, N=N, K=K, lda=K, ldb=K, ldc=N, bias=b, gate=gate)
    assert torch.equal(big[:m], small)
    rows = torch.arange(M - 8, M)
    z = x[rows.cuda()].double().cpu() @ w.double().cpu().t()
    if b is not None:
        z = z + b.double().cpu()
    assert _rel(big[rows.cuda()], 0.37 * z) < 6e-3


def test_cls_only_attention_vs_fp64(ops):
    """ops.cls_attention: the CLS query of a divided space attention over ALL S keys, straight from the fused qkv matrix (the last block
    of a video pass, model.py::_video_block_tail): output and the whole dqkv (dK | dV of every row, dQ of the CLS rows, zeros elsewhere)
    against fp64 torch; full token geometry (S = 3137), B = 2."""
    B, S, H = 2, 3137, 12
    D = H * 64
    qkv = (_rnd((B * S, 3 * D), torch.float32, 1.0, 41) * 0.5).to(torch.bfloat16).cuda().requires_grad_(True)
    dO = _rnd((B, D), torch.float32, 1.0, 42).to(torch.bfloat16).cuda()
    O = ops.cls_attention(qkv, B, S, H)
    O.backward(dO)
    x = qkv.detach().double().cpu().reshape(B, S, 3, H, 64).requires_grad_(True)
    q = x[:, 0, 0]                                                # (B, H, 64)
    k, v = x[:, :, 1], x[:, :, 2]                                  # (B, S, H, 64)
    sc = torch.einsum('bhd,bshd->bhs', q, k) * 0.125
    o = torch.einsum('bhs,bshd->bhd', torch.softmax(sc, -1), v).reshape(B, D)
    o.backward(dO.double().cpu())
    assert _rel(O, o.detach()) < 1e-2
    g, gr = qkv.grad.double().cpu().reshape(B, S, 3, H, 64), x.grad
    assert _rel(g[:, :, 1:], gr[:, :, 1:]) < 2e-2                  # dK | dV of every row
    assert _rel(g[:, 0, 0], gr[:, 0, 0]) < 2e-2                    # dQ of the CLS rows
    assert float(g[:, 1:, 0].abs().max()) == 0.0                   # the other rows' queries took no part


@pytest.mark.parametrize('bm', [256, 224, 192, 160, 128])
@pytest.mark.parametrize('N,K,bias', [(768, 768, False), (2304, 768, True), (768, 2304, False)])
def test_persistent_gemm_every_tile_height_is_bitwise_the_same(ops, N, K, bias, bm, monkeypatch):
    """round 5: the persistent GEMM picks its tile height per call (A sub-tiles of 4 / 3 / 2 fragments: 256, 224, 192, 160 or 128 rows)
    to fill the last round of its walk.  A row's result must not depend on the height: every forced height (EGV_PP_FORCE_BM) against the
    height the launcher picks on its own, bit for bit over the WHOLE output at full M (ragged last tile included), plus fp64 on sampled
    rows."""
    M = FULL_M
    x = _rnd((M, K), torch.bfloat16, 1.0, 21).cuda()
    w = _rnd((N, K), torch.float32, 0.05, 22).cuda()
    b = _rnd((N,), torch.float32, 0.5, 23).cuda() if bias else None
    auto = ops.linear(x, w, b)
    monkeypatch.setenv('EGV_PP_FORCE_BM', str(bm))
    forced = ops.linear(x, w, b)
    monkeypatch.delenv('EGV_PP_FORCE_BM')
    assert torch.equal(auto, forced)
    rows = torch.cat([torch.arange(0, 8), torch.arange(M // 2 + 100, M // 2 + 108), torch.arange(M - 8, M)])
    z = x[rows.cuda()].double().cpu() @ w.to(torch.bfloat16).double().cpu().t()
    if b is not None:
        z = z + b.double().cpu()
    assert _rel(forced[rows.cuda()], z) < 6e-3


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('fused', [False, True])
def test_block_entry_points_match_per_op_composition(ops, dtype, fused, monkeypatch):
    """egv_vblock_* / egv_tlayer_* (csrc/egv_block.cpp) against the same block composed from the per-operation entry points:
    identical forward values (same kernels, same order), gradients equal up to the order of the skip-gradient sums.  (The video
    block with its residual sums in the GEMM epilogues, EGV_VIDEO_RES32=0; the fp32-stream form has its own test below.)"""
    monkeypatch.setenv('EGV_VIDEO_RES32', '0')
    dev = 'cuda'
    B, L, H, D, Hd, Fr, N = 2, 16, 12, 768, 3072, 4, 49
    S = 1 + Fr * N
    mk = lambda shape, sc=0.05, seed=0: (_rnd(shape, torch.float32, sc, seed).cuda()).requires_grad_(True)   # noqa: E731
    seeds = iter(range(100, 400))
    tp = []
    for _ in range(4):
        tp += [mk((D, D), seed=next(seeds)), mk((D,), seed=next(seeds))]
    tp += [mk((Hd, D), seed=next(seeds)), mk((Hd,), seed=next(seeds)), mk((D, Hd), seed=next(seeds)), mk((D,), seed=next(seeds)),
           mk((D,), 1.0, next(seeds)), mk((D,), seed=next(seeds)), mk((D,), 1.0, next(seeds)), mk((D,), seed=next(seeds))]
    if fused:
        for _ in range(4):
            tp += [mk((D, D), seed=next(seeds)), mk((D,), seed=next(seeds))]
        tp += [mk((1,), 1.0, next(seeds))]
    vp = [mk((3 * D, D), seed=next(seeds)), mk((3 * D,), seed=next(seeds)), mk((D, D), seed=next(seeds)), mk((D,), seed=next(seeds)),
          mk((3 * D, D), seed=next(seeds)), mk((3 * D,), seed=next(seeds)), mk((D, D), seed=next(seeds)), mk((D,), seed=next(seeds)),
          mk((Hd, D), seed=next(seeds)), mk((Hd,), seed=next(seeds)), mk((D, Hd), seed=next(seeds)), mk((D,), seed=next(seeds))]
    for _ in range(3):
        vp += [mk((D,), 1.0, next(seeds)), mk((D,), seed=next(seeds))]
    if fused:
        vp += [mk((2 * D, D), seed=next(seeds)), mk((2 * D,), seed=next(seeds)), mk((D, D), seed=next(seeds)), mk((D,), seed=next(seeds)),
               mk((D, D), seed=next(seeds)), mk((D,), seed=next(seeds)), mk((D,), 1.0, next(seeds)), mk((D,), seed=next(seeds)), mk((1,), 1.0, next(seeds))]
    hid0 = _rnd((B * L, D), torch.float32, 1.0, 1).cuda()
    x0 = _rnd((B * S, D), torch.float32, 1.0, 2).cuda()
    m = torch.ones(B, L, device=dev)
    m[0, -3:] = 0
    mask = ((1 - m) * torch.finfo(torch.float32).min).contiguous()

    def text_ref(hid, P, enc):
        q, k, v = ops.linear(hid, P[0], P[1]), ops.linear(hid, P[2], P[3]), ops.linear(hid, P[4], P[5])
        ctx = ops.plain_attention(q, k, v, B, H, L, L, 0.125, mask=mask)
        if enc is None:
            a = ops.linear(ctx, P[6], P[7], res1=hid)
        else:
            a0 = ops.linear(ctx, P[6], P[7])
            cq, ck, cv = ops.linear(a0, P[16], P[17]), ops.linear(enc, P[18], P[19]), ops.linear(enc, P[20], P[21])
            cctx = ops.plain_attention(cq, ck, cv, B, H, L, S, 0.125, mask=None)
            a = ops.linear(cctx, P[22], P[23], gate=P[24], res1=a0, res2=hid)
        a = ops.layernorm(a, P[12], P[13], 1e-5)
        return ops.layernorm(ops.mlp(a, P[8], P[9], P[10], P[11], res=a), P[14], P[15], 1e-5)

    def video_ref(x, P, y):
        h, xs = ops.layernorm_skip(x, P[12], P[13], 1e-5)
        tr = ops.linear(ops.divided_attention(ops.linear(h, P[0], P[1]), B, Fr, N, H, 'time'), P[2], P[3], res1=xs)
        s_ctx = ops.divided_attention(ops.linear(ops.layernorm(tr, P[14], P[15], 1e-5), P[4], P[5]), B, Fr, N, H, 'space')
        if y is None:
            sr = ops.linear(s_ctx, P[6], P[7], res1=xs)
        else:
            s = ops.linear(s_ctx, P[6], P[7])
            kv = ops.linear(y, P[18], P[19])
            hs, ss = ops.layernorm_skip(s, P[24], P[25], 1e-5)
            o = ops.plain_attention(ops.linear(hs, P[20], P[21]), kv[:, :D], kv[:, D:], B, H, S, L, 0.125, mask=mask)
            sr = ops.linear(o, P[22], P[23], gate=P[26], res1=ss, res2=xs)
        h2, srs = ops.layernorm_skip(sr, P[16], P[17], 1e-5)
        return ops.mlp(h2, P[8], P[9], P[10], P[11], res=srs)

    for name in ('text', 'video'):
        P = tp if name == 'text' else vp
        outs = []
        for which in ('ref', 'blk'):
            for p in P:
                p.grad = None
            hid = hid0.to(dtype).detach().requires_grad_(True)
            x = x0.to(dtype).detach().requires_grad_(True)
            if name == 'text':
                enc = x if fused else None
                out = text_ref(hid, P, enc) if which == 'ref' else ops.text_layer(hid, mask, P, B, L, H, Hd, 1e-5, enc=enc, S=S)
            else:
                y = hid if fused else None
                out = video_ref(x, P, y) if which == 'ref' else ops.video_block(x, P, B, Fr, N, H, Hd, 1e-5, y=y, y_mask=mask if fused else None, L=L)
            out.backward(_rnd(tuple(out.shape), torch.float32, 1.0, 9).cuda().to(dtype))
            torch.cuda.synchronize()
            outs.append((out.detach().clone(), None if hid.grad is None else hid.grad.clone(), None if x.grad is None else x.grad.clone(),
                         [p.grad.clone() for p in P]))
        (o0, h0, xg0, g0), (o1, h1, xg1, g1) = outs
        assert torch.equal(o0, o1), name
        tol = 1e-5 if dtype == torch.float32 else 1e-2
        for a, bb in ((h0, h1), (xg0, xg1)):
            if a is not None:
                assert _rel(bb, a.double().cpu()) < tol, name
        for i, (a, bb) in enumerate(zip(g0, g1)):
            if not (name == 'text' and i in (3, 19)):     # key biases have an exactly-zero true gradient: noise / noise
                # (the gate gradients are dot products of two noisy bf16 vectors: a looser bound)
                assert _rel(bb, a.double().cpu()) < (1e-5 if dtype == torch.float32 else (1e-1 if a.numel() == 1 else 2e-2)), (name, i)


def test_sum_layernorm_vs_torch(ops):
    """egv_sum_layernorm: s = base + d1 + d2 + gate * dg in fp32 (fp32 or bf16 base), its fp32 / bf16 forms and LayerNorm(s), against
    torch on the same operands; in-place form (sum16 aliasing d1); a width that is not a multiple of 256 and a ragged row count."""
    import ctypes as C
    from egovlpv2_amd import _lib as L
    for M, D in ((1571, 768), (333, 520)):
        base32 = _rnd((M, D), torch.float32, 2.0, 1).cuda()
        d1, d2, dg = (_rnd((M, D), torch.bfloat16, 1.0, 2 + i).cuda() for i in range(3))
        gate = torch.tensor([0.37], device='cuda')
        gamma, beta = _rnd((D,), torch.float32, 1.0, 7).cuda(), _rnd((D,), torch.float32, 0.1, 8).cuda()
        for b32 in (True, False):
            base = base32 if b32 else base32.bfloat16()
            s32 = torch.empty(M, D, device='cuda')
            s16 = torch.empty(M, D, device='cuda', dtype=torch.bfloat16)
            y = torch.empty_like(s16)
            stats = torch.empty(M, 2, device='cuda')
            p = lambda t: C.c_void_p(t.data_ptr())        # noqa: E731
            rc = L.lib.egv_sum_layernorm(p(base) if b32 else None, None if b32 else p(base), p(d1), p(d2), p(dg), p(gate), p(s32), p(s16), p(y),
                                         p(gamma), p(beta), p(stats), M, D, 1e-5, None)
            assert rc == 0, L.lib.egv_last_error()
            torch.cuda.synchronize()
            ref = base.double() + d1.double() + d2.double() + 0.37 * dg.double()
            assert _rel(s32, ref.cpu()) < 1e-6
            assert torch.equal(s16, s32.bfloat16())
            ln = torch.nn.functional.layer_norm(s32.double(), (D,), gamma.double(), beta.double(), 1e-5)
            assert _rel(y, ln.cpu()) < 4e-3
            assert _rel(stats[:, 0], s32.double().mean(1).cpu()) < 1e-5
            # in place: the Linear output becomes the sum, only the LayerNorm output is new
            d1c = d1.clone()
            rc = L.lib.egv_sum_layernorm(p(base) if b32 else None, None if b32 else p(base), p(d1c), None, None, None, None, p(d1c), p(y),
                                         p(gamma), p(beta), None, M, D, 1e-5, None)
            assert rc == 0, L.lib.egv_last_error()
            torch.cuda.synchronize()
            assert torch.equal(d1c, (base.float() + d1.float()).bfloat16())


@pytest.mark.parametrize('fused', [False, True])
def test_video_block_fp32_residual_stream(ops, fused, monkeypatch):
    """EGV_VIDEO_RES32 (the default of the bf16 mode): SpaceTimeBlock.forward with its three residual sums and LayerNorm inputs in fp32
    (video_transformer.py:217-226 under trainer_egoclip.py:143's autocast).  Against the block composed from the per-operation
    entry points (bf16 GEMMs without residual epilogues) with the sums and LayerNorms in torch fp32; the bf16 output is the rounding
    of the fp32 one; a second block continues from the fp32 value; over a stack of blocks the fp32 stream stays closer to fp32
    arithmetic than the bf16 stream does."""
    F = torch.nn.functional
    B, L, H, D, Hd, Fr, N = 2, 16, 12, 768, 3072, 4, 49
    S = 1 + Fr * N
    seeds = iter(range(500, 900))
    mk = lambda shape, sc=0.05: _rnd(shape, torch.float32, sc, next(seeds)).cuda()   # noqa: E731
    P = [mk((3 * D, D)), mk((3 * D,)), mk((D, D)), mk((D,)), mk((3 * D, D)), mk((3 * D,)), mk((D, D)), mk((D,)),
         mk((Hd, D)), mk((Hd,)), mk((D, Hd)), mk((D,))]
    for _ in range(3):
        P += [mk((D,), 1.0), mk((D,))]
    if fused:
        P += [mk((2 * D, D)), mk((2 * D,)), mk((D, D)), mk((D,)), mk((D, D)), mk((D,)), mk((D,), 1.0), mk((D,)), mk((1,), 1.0)]
    x32 = _rnd((B * S, D), torch.float32, 1.0, 2).cuda()
    x = x32.bfloat16()
    y = _rnd((B * L, D), torch.bfloat16, 1.0, 1).cuda() if fused else None
    m = torch.ones(B, L, device='cuda')
    m[0, -3:] = 0
    mask = ((1 - m) * torch.finfo(torch.float32).min).contiguous()
    bf = torch.bfloat16

    def ref(base):
        ln = lambda t, i: F.layer_norm(t, (D,), P[i], P[i + 1], 1e-5).to(bf)      # noqa: E731
        yt = ops.linear(ops.divided_attention(ops.linear(ln(base, 12), P[0], P[1]), B, Fr, N, H, 'time'), P[2], P[3])
        s_ctx = ops.divided_attention(ops.linear(ln(base + yt.float(), 14), P[4], P[5]), B, Fr, N, H, 'space')
        s = ops.linear(s_ctx, P[6], P[7])
        sr = base + s.float()
        if fused:
            kv = ops.linear(y, P[18], P[19])
            hs = ops.layernorm(s, P[24], P[25], 1e-5)
            o = ops.plain_attention(ops.linear(hs, P[20], P[21]), kv[:, :D], kv[:, D:], B, H, S, L, 0.125, mask=mask)
            sr = sr + P[26] * ops.linear(o, P[22], P[23]).float()
        return sr + ops.mlp(ln(sr, 16), P[8], P[9], P[10], P[11]).float()

    def blk(t):
        return ops.video_block(t, P, B, Fr, N, H, Hd, 1e-5, y=y, y_mask=mask if fused else None, L=L)

    with torch.no_grad():
        out = blk(x)                                               # head of the stream: the bf16 tensor is exact
        o32 = ops.stream32(out)
        assert o32 is not None and o32.dtype == torch.float32
        assert torch.equal(out, o32.to(bf))
        r1 = ref(x.float())
        assert _rel(o32, r1.double().cpu()) < 6e-3                 # (bf16 roundings of the LayerNorm outputs differ between torch and the kernel)
        out2 = blk(out)                                            # continues from the fp32 value, not from its rounding
        assert _rel(ops.stream32(out2), ref(o32).double().cpu()) < 6e-3
        if not fused:
            # a stack of blocks (the same weights again and again): distance from fp32 arithmetic, fp32 stream vs bf16 stream
            Pf = [p.clone() for p in P]
            xf = x.float()
            a = x
            monkeypatch.setenv('EGV_VIDEO_RES32', '0')
            b = x
            for _ in range(6):
                b = blk(b)
            monkeypatch.setenv('EGV_VIDEO_RES32', '1')
            for _ in range(6):
                xf = ops.video_block(xf, Pf, B, Fr, N, H, Hd, 1e-5)
                a = blk(a)
            e32, e16 = _rel(ops.stream32(a), xf.double().cpu()), _rel(b, xf.double().cpu())
            assert e32 < e16, (e32, e16)


@pytest.mark.parametrize('M', [25096, 4096 + 8, 1100])
def test_grouped_weight_gradients_bf16(ops, M):
    """egv_gemm_wgrad_grouped: the six weight gradients of a SpaceTimeBlock (video_transformer.py:53,56,120,152 autograd) in one
    launch, reduction splits summed by the last arriver, against fp64 math on the bf16 operands (2e-3: fp32 accumulation of M
    products); bias gradients and the gate factor included; bit-identical between two launches (the sum is in split order)."""
    D, Hd = 256, 1024
    shapes = [(D, Hd, True, False), (Hd, D, True, False), (D, D, True, True), (3 * D, D, True, False), (D, D, False, False), (3 * D, D, True, False)]
    probs, refs = [], []
    gate = torch.tensor([0.37], device='cuda')
    for i, (N, K, bias, gated) in enumerate(shapes):
        dy = _rnd((M, N), torch.bfloat16, 1.0, 10 + i).cuda()
        x = _rnd((M, K), torch.bfloat16, 1.0, 30 + i).cuda()
        probs.append((dy, x, bias, gate if gated else None))
        sc = 0.37 if gated else 1.0
        refs.append((sc * (dy.double().t() @ x.double()), sc * dy.double().sum(0)))
    outs = ops.wgrad_grouped(probs, M)
    torch.cuda.synchronize()
    for (dw, db), (rw, rb), (N, K, bias, _g) in zip(outs, refs, shapes):
        assert _rel(dw, rw) < 2e-3, (N, K, _rel(dw, rw))
        if bias:
            assert _rel(db, rb) < 2e-3, (N, K, _rel(db, rb))
    again = ops.wgrad_grouped(probs, M)
    torch.cuda.synchronize()
    for (dw, db), (dw2, db2) in zip(outs, again):
        assert torch.equal(dw, dw2)
        assert db is None or torch.equal(db, db2)


def test_grouped_weight_gradients_under_load_bf16(ops):
    """the in-launch hand-off of the grouped weight gradients must not depend on dispatch order or timing: the same group, launched
    while another stream keeps the chip busy with a persistent GEMM and a streaming copy, gives bit-identical results every time"""
    M, D = 25096, 768
    dy = [_rnd((M, n), torch.bfloat16, 1.0, 50 + i).cuda() for i, n in enumerate((D, 3 * D, D))]
    x = [_rnd((M, D), torch.bfloat16, 1.0, 60 + i).cuda() for i in range(3)]
    probs = [(dy[i], x[i], True, None) for i in range(3)]
    ref = [(a.clone(), b.clone()) for a, b in ops.wgrad_grouped(probs, M)]
    torch.cuda.synchronize()
    a = _rnd((M, 768), torch.bfloat16, 1.0, 70).cuda()
    w = _rnd((3072, 768), torch.float32, 0.05, 71).cuda()
    big = torch.empty(256 << 20, dtype=torch.uint8, device='cuda')
    side = torch.cuda.Stream()
    for it in range(6):
        with torch.cuda.stream(side):
            for _ in range(3):
                ops.linear(a, w, None)
                big.copy_(big.flip(0)) if it % 2 else None
        outs = ops.wgrad_grouped(probs, M)
        torch.cuda.synchronize()
        for (dw, db), (rw, rb) in zip(outs, ref):
            assert torch.equal(dw, rw) and torch.equal(db, rb), it


# ---------------------------------------------------------------------------------------------------------------------------
# MX-fp8 operand format and GEMM (BASELINE.json configs[4])
# ---------------------------------------------------------------------------------------------------------------------------
def _mx_inputs(R, K, seed, spread=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(R, K, generator=g)
    if spread:      # block maxima across ~40 binades, exact powers of two and 1.75 * 2^e (the boundary of the scale rule), zero blocks
        x = x * torch.exp2(torch.randint(-20, 20, (R, K // 32, 1), generator=g).float()).expand(R, K // 32, 32).reshape(R, K)
        x[0, :32] = 0.0
        x[1, :32] = 0.0; x[1, 3] = 1.75 * 2.0 ** -3
        x[2, :32] = 0.0; x[2, 5] = -(1.75 + 2.0 ** -7) * 2.0 ** 4
        x[3, :32] = 0.0; x[3, 7] = 2.0 ** 9
    return x.to(torch.bfloat16)


@pytest.mark.parametrize('R,K,role', [(300, 384, 0), (256, 1024, 1), (77, 128, 0), (1024, 4096, 1), (4113, 1024, 0)])
def test_mx_quantisation_is_bit_exact(ops, R, K, role):
    """egv_quant_mx against the oracle: every e4m3 code and every E8M0 scale byte, in the lane order of its GEMM role"""
    from oracle import mx_quant as O
    x = _mx_inputs(R, K, 5 + R)
    q, sc = ops.quant_mx(x.cuda(), role)
    codes, e8 = O.quantize(x)
    assert torch.equal(q.cpu(), codes)
    assert torch.equal(sc.cpu(), O.scale_layout(e8, role))
    # the format's own bound: |dequant - x| <= 2^-4 of the block maximum (3 mantissa bits at the top binade, coarser below)
    d = O.dequantize(codes, e8)
    amax = x.float().reshape(R, K // 32, 32).abs().amax(-1, keepdim=True).expand(R, K // 32, 32).reshape(R, K)
    assert ((d - x.float()).abs() <= amax * 2.0 ** -4 + 1e-30).all()


@pytest.mark.parametrize('M,N,K', [(256, 256, 384), (1000, 768, 768), (4113, 1024, 1024), (515, 3072, 1024), (2049, 1024, 4096), (300, 320, 512)])
def test_mx_gemm_vs_dequantised_reference(ops, M, N, K):
    """egv_gemm_mx = exact product of the dequantised operands up to fp32 accumulation and the bf16 rounding of the output;
    ragged M (partial last tile), N not a multiple of the 256-wide tile, bias epilogue"""
    from oracle import mx_quant as O
    a, b = _mx_inputs(M, K, 1, spread=False), (_mx_inputs(N, K, 2, spread=False).float() * 0.05).to(torch.bfloat16)
    bias = torch.randn(N)
    aq, asc = ops.quant_mx(a.cuda(), 0)
    bq, bsc = ops.quant_mx(b.cuda(), 1)
    out = ops.gemm_mx(aq, asc, bq, bsc, M, N, K, bias=bias.cuda()).float().cpu()
    ref = (O.gemm_ref(a, b) + bias.double()).float()
    assert _rel(out, ref) < 3e-3                      # bf16 output rounding (2^-9 per element)
    assert (out - ref).abs().max() <= 2.0 ** -7 * ref.abs().max()


def test_mx_gemm_epilogues_vs_reference(ops):
    """the epilogue kinds the block executor uses: GELU with the saved pre-activation (fc1), residual (proj / fc2), GELU' operand
    (fc2 dgrad)"""
    from oracle import mx_quant as O
    M, N, K = 1030, 1024, 512
    a, b = _mx_inputs(M, K, 3, spread=False), (_mx_inputs(N, K, 4, spread=False).float() * 0.05).to(torch.bfloat16)
    bias = torch.randn(N)
    aq, asc = ops.quant_mx(a.cuda(), 0)
    bq, bsc = ops.quant_mx(b.cuda(), 1)
    lin = (O.gemm_ref(a, b) + bias.double()).float()
    pre = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
    out = ops.gemm_mx(aq, asc, bq, bsc, M, N, K, bias=bias.cuda(), act=1, pre=pre)
    assert _rel(pre.float().cpu(), lin) < 3e-3
    assert _rel(out.float().cpu(), torch.nn.functional.gelu(lin)) < 4e-3
    res = torch.randn(M, N).to(torch.bfloat16)
    out = ops.gemm_mx(aq, asc, bq, bsc, M, N, K, bias=bias.cuda(), res1=res.cuda())
    assert _rel(out.float().cpu(), lin + res.float()) < 3e-3
    u = torch.randn(M, N).to(torch.bfloat16)
    out = ops.gemm_mx(aq, asc, bq, bsc, M, N, K, aux=u.cuda(), dact=1)
    uu = u.float().requires_grad_(True)
    torch.nn.functional.gelu(uu).sum().backward()
    assert _rel(out.float().cpu(), (lin - bias) * uu.grad) < 4e-3


@pytest.mark.parametrize('M,D', [(4113, 1024), (394, 512), (1000, 768)])
def test_layernorm_with_fused_mx_output(ops, M, D):
    """egv_layernorm_fwd_mx: the bf16 output equals egv_layernorm_fwd's, and its MX-fp8 codes / scale bytes equal what the standalone
    quantiser makes of that output (bit for bit: the block executor may use either)"""
    x = _rnd((M, D), torch.bfloat16, 1.5, 3).cuda()
    g, b = _rnd((D,), torch.float32, 1.0, 4).cuda(), _rnd((D,), torch.float32, 0.3, 5).cuda()
    y, q, sc = ops.layernorm_mx(x, g, b, 1e-5)
    y0 = ops.layernorm(x, g, b, 1e-5)
    assert torch.equal(y, y0)
    q0, sc0 = ops.quant_mx(y0, 0)
    assert torch.equal(q, q0) and torch.equal(sc, sc0)


@pytest.mark.parametrize('M,N,K', [(1030, 1024, 512), (4113, 4096, 1024), (394, 2048, 512)])
def test_mx_gemm_quantised_output_is_the_quantiser_s(ops, M, N, K):
    """egv_gemm_mx(out_q, out_scales): the MX-fp8 form of the output written by the GEMM's epilogue (fc1's GELU with saved
    pre-activation, fc2's GELU' data gradient) equals egv_quant_mx of the bf16 output, bit for bit, and the bf16 outputs do not change"""
    a, b = _mx_inputs(M, K, 3, spread=False), (_mx_inputs(N, K, 4, spread=False).float() * 0.05).to(torch.bfloat16)
    bias = torch.randn(N).cuda()
    aq, asc = ops.quant_mx(a.cuda(), 0)
    bq, bsc = ops.quant_mx(b.cuda(), 1)
    pre0 = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
    pre1 = torch.empty_like(pre0)
    out0 = ops.gemm_mx(aq, asc, bq, bsc, M, N, K, bias=bias, act=1, pre=pre0)
    out1, oq, osc = ops.gemm_mx(aq, asc, bq, bsc, M, N, K, bias=bias, act=1, pre=pre1, quant_out=True)
    assert torch.equal(out0, out1) and torch.equal(pre0, pre1)
    q0, s0 = ops.quant_mx(out0, 0)
    assert torch.equal(oq, q0) and torch.equal(osc, s0)
    u = torch.randn(M, N).to(torch.bfloat16).cuda()
    out0 = ops.gemm_mx(aq, asc, bq, bsc, M, N, K, aux=u, dact=1)
    out1, oq, osc = ops.gemm_mx(aq, asc, bq, bsc, M, N, K, aux=u, dact=1, quant_out=True)
    assert torch.equal(out0, out1)
    q0, s0 = ops.quant_mx(out0, 0)
    assert torch.equal(oq, q0) and torch.equal(osc, s0)
