"""torch.autograd plumbing over the HIP C ABI (include/egovlp_hip.h).

PyTorch is used here for device memory, streams and the autograd graph only: every forward and
backward computation below is a call into ``libegovlp_hip.so``.  There is no CPU or ATen fallback;
tensors must live on a ROCm device.

Activations are stored in the model's compute dtype (torch.float32 -> EGV_F32, torch.bfloat16 ->
EGV_BF16); parameters and their gradients are fp32.  In bf16 mode weights are cast once per parameter
version (``compute_weight``) -- the cast kernel is part of the timed step.
"""
from __future__ import annotations

import ctypes as C
import os

import torch
from torch.autograd import Function

from . import _lib as L
from . import switches as SW
from ._lib import lib, check, AttnDesc

ACT = {'none': L.ACT_NONE, 'gelu': L.ACT_GELU, 'relu': L.ACT_RELU, 'tanh': L.ACT_TANH}


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return L.EGV_BF16
    if t.dtype == torch.float32:
        return L.EGV_F32
    raise TypeError(f"unsupported activation dtype {t.dtype}")


def _p(t):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _st():
    """raw hipStream_t of torch's current stream (called once per launch: the C getter is ~20x cheaper than building a
    torch.cuda.Stream object)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("egovlpv2_amd: the hot path runs only on a ROCm device through libegovlp_hip.so "
                           "(no CPU fallback); got a CPU tensor")


# ---- scratch space (stream-ordered reuse on the current stream) ---------------------------------------
_ws = {}


def workspace(nbytes: int, device, slot: int = 0) -> torch.Tensor:
    key = (device, slot, _st())      # one scratch buffer per stream
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


# ---- compute-dtype copies of fp32 master weights -------------------------------------------------------
_wcache = {}


def _cast_event():
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    return ev


_stream_seen = {}        # stream handle -> (id of the cast event this stream already waits for, ids of copies recorded on it)


def _cross_stream(ent):
    """a cached copy made on another stream: order this stream after the cast kernel, keep the allocator informed -- once per
    stream and cast (all copies of one prepare_weights launch share one event), not once per use"""
    st = _st()
    if ent[3] != st:
        seen = _stream_seen.get(st)
        if seen is None or seen[0] != id(ent[4]):
            cur = torch.cuda.current_stream()
            cur.wait_event(ent[4])
            seen = _stream_seen[st] = (id(ent[4]), set(), cur, ent[4])
        if id(ent[1]) not in seen[1]:
            ent[1].record_stream(seen[2])
            seen[1].add(id(ent[1]))
    return ent[1]


def compute_weight(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """fp32 parameter -> tensor in the compute dtype (cached per parameter version)."""
    if dtype == torch.float32:
        return w
    key = id(w)
    ent = _wcache.get(key)
    if ent is not None and ent[0] == w._version and ent[1].device == w.device and ent[2] is w:
        return _cross_stream(ent)
    out = torch.empty(w.shape, dtype=dtype, device=w.device)
    src = w.detach()
    if not src.is_contiguous():
        src = src.contiguous()
    check(lib.egv_cast(L.EGV_F32, _dt(out), _p(src), _p(out), src.numel(), _st()), 'egv_cast')
    _wcache[key] = (w._version, out, w, _st(), _cast_event())
    return out


_wtcache = {}


def compute_weight_t(w: torch.Tensor, dtype: torch.dtype):
    """transposed bf16 compute copy W^T [K_in, N_out] of an fp32 weight [N_out, K_in] (cached per parameter version):
    lets dgrad run in the K-contiguous (NT) GEMM form on the DMA-staged kernel.  None when not applicable."""
    if dtype != torch.bfloat16 or w.dim() != 2 or (w.shape[0] % 64) or (w.shape[1] % 8):
        return None
    key = id(w)
    ent = _wtcache.get(key)
    if ent is not None and ent[0] == w._version and ent[2] is w and (ent[1] is None or ent[1].device == w.device):
        return None if ent[1] is None else _cross_stream(ent)
    N, K = w.shape
    out = torch.empty(K, N, dtype=dtype, device=w.device)
    src = w.detach()
    if not src.is_contiguous():
        src = src.contiguous()
    check(lib.egv_cast_transpose(_p(src), _p(out), N, K, _st()), 'egv_cast_transpose')
    _wtcache[key] = (w._version, out, w, _st(), _cast_event())
    return out


_wprep = {}
_wmerged = {}            # id(first weight of a group) -> (versions, merged W [sum R, C], merged W^T [C, sum R], merged bias or None, members, stream, event)


_wmx = {}    # id(fp32 weight) -> (version, codes [R,C], scales, codes of W^T [C,R], scales, weight, stream, event): MX-fp8 copies


def mx_weight(w):
    """(q, q_scales, qt, qt_scales) of a weight prepared by prepare_weights(..., fp8=[...]) for its current version, else None"""
    e = _wmx.get(id(w))
    if e is None or e[0] != w._version or e[5] is not w:
        return None
    _cross_stream((e[0], e[1], None, e[6], e[7]))
    return e[1], e[2], e[3], e[4]


def prepare_weights(weights, dtype: torch.dtype, groups=(), fp8=()):
    """One launch that makes the bf16 W and W^T compute copies of every listed fp32 [R, C] weight (R, C multiples of 64) and
    seeds the two caches above, so that the forward/backward of the step finds them ready.  Weights that do not qualify are
    left to the lazy per-tensor path.  A no-op when the copies of the current parameter versions already exist.

    groups: tuples (weights, biases) of Linears that read the SAME input (query / key / value of a RoBERTa layer, key / value of its
    text-to-image cross attention): their copies are laid out as ONE [sum R, C] matrix (row blocks) and ONE [C, sum R] transposed matrix
    (column blocks, written with a row pitch), and their biases are concatenated by a second small launch, so that the layer can run
    them as one GEMM (merged_weights).  The per-weight entries of the caches are views of the merged buffers.

    fp8: weights (a subset of `weights`, not in a group, R and C multiples of 128) that also get MX-fp8 copies of W and W^T (one more
    launch over the bf16 copies: egv_quant_mx_batch, role 1), for the MX-fp8 forward / dgrad GEMMs of the video blocks (mx_weight)."""
    if dtype != torch.bfloat16:
        return
    ws = [w for w in weights if w.dim() == 2 and w.dtype == torch.float32 and w.is_contiguous() and w.is_cuda
          and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0]
    if not ws:
        return
    e0 = _wcache.get(id(ws[0]))
    if e0 is not None and e0[0] == ws[0]._version and e0[2] is ws[0] and all(
            (c := _wcache.get(id(w))) is not None and c[0] == w._version and c[2] is w for w in ws):
        return
    import numpy as np
    key = tuple(id(w) for w in ws) + ('mx',) + tuple(id(w) for w in fp8)
    ent = _wprep.get(key)
    ptrs = [w.data_ptr() for w in ws]
    if ent is None or ent['ptrs'] != ptrs:
        dev = ws[0].device
        in_list = {id(w) for w in ws}
        member = {}                                    # id(w) -> (merged W, merged W^T, row offset, total rows)
        merged = []
        for gw, gb in groups:
            if not all(id(w) in in_list for w in gw) or len({w.shape[1] for w in gw}) != 1:
                continue
            R, Cc = sum(w.shape[0] for w in gw), gw[0].shape[1]
            mw, mt = torch.empty(R, Cc, dtype=dtype, device=dev), torch.empty(Cc, R, dtype=dtype, device=dev)
            mb = torch.empty(R, dtype=torch.float32, device=dev) if (gb and all(b_ is not None for b_ in gb)) else None
            off = 0
            for w in gw:
                member[id(w)] = (mw, mt, off, R)
                off += w.shape[0]
            merged.append((gw, gb, mw, mt, mb))
        outs, ldts = [], []
        for w in ws:
            m = member.get(id(w))
            if m is None:
                outs.append((torch.empty(w.shape, dtype=dtype, device=dev), torch.empty(w.shape[1], w.shape[0], dtype=dtype, device=dev)))
                ldts.append(w.shape[0])
            else:
                mw, mt, off, R = m
                outs.append((mw[off:off + w.shape[0]], mt[:, off:off + w.shape[0]]))       # W^T block: row pitch R (only the merged matrix is used)
                ldts.append(R)
        arr = np.zeros(len(ws), dtype=[('s', '<u8'), ('d', '<u8'), ('t', '<u8'), ('R', '<i4'), ('C', '<i4'), ('ldt', '<i4'), ('pad', '<i4')])
        arr['s'] = ptrs
        arr['d'] = [o[0].data_ptr() for o in outs]
        arr['t'] = [o[1].data_ptr() for o in outs]
        arr['R'] = [w.shape[0] for w in ws]
        arr['C'] = [w.shape[1] for w in ws]
        arr['ldt'] = ldts
        prefix = np.zeros(len(ws) + 1, dtype=np.int32)
        prefix[1:] = np.cumsum([(w.shape[0] // 64) * (w.shape[1] // 64) for w in ws])
        pin = torch.from_numpy(arr.view(np.uint8).copy()).pin_memory()
        pre = torch.from_numpy(prefix).pin_memory()
        segs = [(b_, mb, o) for gw, gb, mw, mt, mb in merged if mb is not None
                for b_, o in zip(gb, np.cumsum([0] + [w.shape[0] for w in gw])[:-1])]
        seg_tab = seg_pin = None
        if segs:
            sarr = np.zeros(len(segs), dtype=[('s', '<u8'), ('d', '<u8'), ('n', '<i8')])
            sarr['s'] = [b_.data_ptr() for b_, _, _ in segs]
            sarr['d'] = [mb.data_ptr() + 4 * int(o) for _, mb, o in segs]
            sarr['n'] = [b_.numel() for b_, _, _ in segs]
            seg_pin = torch.from_numpy(sarr.view(np.uint8).copy()).pin_memory()
            seg_tab = seg_pin.to(dev, non_blocking=True)
        ent = _wprep[key] = {'ptrs': ptrs, 'outs': outs, 'table': pin.to(dev, non_blocking=True), 'prefix': pre.to(dev, non_blocking=True),
                             'ntiles': int(prefix[-1]), 'pins': (pin, pre, seg_pin), 'refs': ws, 'merged': merged, 'segs': seg_tab, 'nseg': len(segs),
                             'bias_ptrs': [b_.data_ptr() for b_, _, _ in segs], 'mx': None}
        want = {id(w) for w in fp8}
        mxw = [(w, o, ot) for w, (o, ot) in zip(ws, outs) if id(w) in want and id(w) not in member and w.shape[0] % 128 == 0 and w.shape[1] % 128 == 0]
        if mxw:
            recs, mouts = [], []
            for w, o, ot in mxw:
                R, Cc = w.shape
                q, qt = torch.empty(R, Cc, dtype=torch.uint8, device=dev), torch.empty(Cc, R, dtype=torch.uint8, device=dev)
                sq = torch.full((lib.egv_mx_scale_bytes(R, Cc, 1),), 0x7f, dtype=torch.uint8, device=dev)
                sqt = torch.full((lib.egv_mx_scale_bytes(Cc, R, 1),), 0x7f, dtype=torch.uint8, device=dev)
                mouts.append((q, sq, qt, sqt))
                recs.append((o.data_ptr(), q.data_ptr(), sq.data_ptr(), R, Cc, Cc, 1))
                recs.append((ot.data_ptr(), qt.data_ptr(), sqt.data_ptr(), Cc, R, R, 1))
            marr = np.zeros(len(recs), dtype=[('s', '<u8'), ('q', '<u8'), ('c', '<u8'), ('R', '<i4'), ('K', '<i4'), ('ld', '<i4'), ('role', '<i4')])
            for k, name in enumerate(('s', 'q', 'c', 'R', 'K', 'ld', 'role')):
                marr[name] = [r[k] for r in recs]
            mpre = np.zeros(len(recs) + 1, dtype=np.int32)
            mpre[1:] = np.cumsum([(r[3] * (r[4] // 32) + 255) // 256 for r in recs])
            mpin, mppin = torch.from_numpy(marr.view(np.uint8).copy()).pin_memory(), torch.from_numpy(mpre).pin_memory()
            ent['mx'] = {'w': [m[0] for m in mxw], 'outs': mouts, 'table': mpin.to(dev, non_blocking=True), 'prefix': mppin.to(dev, non_blocking=True),
                         'n': len(recs), 'nblocks': int(mpre[-1]), 'pins': (mpin, mppin)}
    check(lib.egv_cast_weights_ld(_p(ent['table']), _p(ent['prefix']), len(ws), ent['ntiles'], _st()), 'egv_cast_weights_ld')
    if ent['nseg']:
        check(lib.egv_copy_segments(_p(ent['segs']), ent['nseg'], _st()), 'egv_copy_segments')
    if ent['mx'] is not None:
        m = ent['mx']
        check(lib.egv_quant_mx_batch(_p(m['table']), _p(m['prefix']), m['n'], m['nblocks'], _st()), 'egv_quant_mx_batch')
    ev, st = _cast_event(), _st()
    if ent['mx'] is not None:
        for w, (q, sq, qt, sqt) in zip(ent['mx']['w'], ent['mx']['outs']):
            _wmx[id(w)] = (w._version, q, sq, qt, sqt, w, st, ev)
    for w, (o, ot) in zip(ws, ent['outs']):
        _wcache[id(w)] = (w._version, o, w, st, ev)
        # a member of a merged group has no transposed copy of its own (its block of the merged W^T has the group's row pitch):
        # an un-merged dgrad on it reads W as a [reduction, out] operand
        _wtcache[id(w)] = (w._version, ot if ot.is_contiguous() else None, w, st, ev)
    for gw, gb, mw, mt, mb in ent['merged']:
        vers = tuple(w._version for w in gw) + tuple(b_._version for b_ in (gb or ()) if b_ is not None)
        _wmerged[id(gw[0])] = (vers, mw, mt, mb, (gw, gb), st, ev)


def merged_weights(ws, bs):
    """(W [sum R, C], W^T [C, sum R], bias [sum R]) of a group prepared by prepare_weights(..., groups=...), or None when the group has
    no current merged copies (parameters changed since, another dtype, lazy per-tensor casts)"""
    ent = _wmerged.get(id(ws[0]))
    if ent is None:
        return None
    gw, gb = ent[4]
    if len(gw) != len(ws) or any(a is not b for a, b in zip(gw, ws)) or (ent[3] is not None and any(a is not b for a, b in zip(gb, bs))):
        return None
    vers = tuple(w._version for w in ws) + (tuple(b_._version for b_ in bs) if ent[3] is not None else ())
    if vers != ent[0] or ent[3] is None:
        return None
    c = _wcache.get(id(ws[0]))
    if c is None or c[4] is not ent[6]:                     # the per-weight copies were re-made by another path since
        return None
    _cross_stream((ent[0], ent[1], None, ent[5], ent[6]))
    return ent[1], ent[2], ent[3]


def dgrad(dz, weight, dx, M, N, K, gate=None, aux=None, dact=0):
    """dx[M,K] = gate * (dz[M,N] @ W[N,K]) * act'(aux): NT form on the transposed bf16 copy when available, else the
    generic kernel reading W as a [reduction, out] operand."""
    wt = compute_weight_t(weight, dz.dtype)
    if wt is not None:
        gemm(dz, wt, dx, a_trans=0, b_trans=0, M=M, N=K, K=N, lda=N, ldb=N, ldc=K, gate=gate, aux=aux, dact=dact)
    else:
        gemm(dz, compute_weight(weight, dz.dtype), dx, a_trans=0, b_trans=1, M=M, N=K, K=N, lda=N, ldb=K, ldc=K, gate=gate,
             aux=aux, dact=dact)


def cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    if x.dtype == dtype:
        return x
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    check(lib.egv_cast(_dt(x), _dt(out), _p(x), _p(out), x.numel(), _st()), 'egv_cast')
    return out


class CastFn(Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return cast(x, dtype)

    @staticmethod
    def backward(ctx, g):
        return cast(g, ctx.src), None


# ---- raw launch helpers ------------------------------------------------------------------------------
def gemm(x, w, out, *, a_trans=0, b_trans=0, M, N, K, lda, ldb, ldc, bias=None, act=0, gate=None, res1=None,
         res2=None, pre=None, aux=None, dact=0, scale=1.0, out_f32=0):
    check(lib.egv_gemm(_dt(x), a_trans, b_trans, M, N, K, _p(x), lda, _p(w), ldb, _p(out), ldc, out_f32, _p(bias), act,
                       _p(gate), _p(res1), _p(res2), _p(pre), _p(aux), dact, ldc, float(scale), _st()), 'egv_gemm')


def quant_mx(x, role):
    """bf16 [R, K] -> (codes uint8 [R, K], scale bytes uint8) in the MXFP8 E4M3 format of egv_quant_mx; role 0 = the GEMM's A
    operand (activations / output gradients), role 1 = its B operand (weights)"""
    R, K = x.shape
    assert x.dtype == torch.bfloat16 and x.stride(1) == 1
    q = torch.empty(R, K, dtype=torch.uint8, device=x.device)
    nb = lib.egv_mx_scale_bytes(R, K, role)
    assert nb > 0, (R, K, role)
    sc = torch.full((nb,), 0x7f, dtype=torch.uint8, device=x.device)
    check(lib.egv_quant_mx(_p(x), R, K, x.stride(0), _p(q), _p(sc), role, _st()), 'egv_quant_mx')
    return q, sc


def layernorm_mx(x, gamma, beta, eps):
    """bf16 LayerNorm that also returns the MX-fp8 form (role 0) of its output: (y, codes, scale bytes) -- egv_layernorm_fwd_mx"""
    M, D = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    y = torch.empty_like(x)
    q = torch.empty(M, D, dtype=torch.uint8, device=x.device)
    sc = torch.full((lib.egv_mx_scale_bytes(M, D, 0),), 0x7f, dtype=torch.uint8, device=x.device)
    stats = torch.empty(M, 2, dtype=torch.float32, device=x.device)
    check(lib.egv_layernorm_fwd_mx(_p(x), _p(y), _p(gamma), _p(beta), _p(stats), _p(q), _p(sc), M, D, float(eps), _st()), 'egv_layernorm_fwd_mx')
    return y, q, sc


def gemm_mx(aq, asc, bq, bsc, M, N, K, *, bias=None, act=0, res1=None, pre=None, aux=None, dact=0, quant_out=False):
    """bf16 C[M,N] = epi(A B^T) on MX-fp8 operands (egv_gemm_mx); aq [M,K] / bq [N,K] codes, asc / bsc their scale bytes.
    quant_out: also return C in MX-fp8 form (codes, role-0 scale bytes) written by the GEMM's epilogue"""
    out = torch.empty(M, N, dtype=torch.bfloat16, device=aq.device)
    oq = osc = None
    if quant_out:
        oq = torch.empty(M, N, dtype=torch.uint8, device=aq.device)
        osc = torch.full((lib.egv_mx_scale_bytes(M, N, 0),), 0x7f, dtype=torch.uint8, device=aq.device)
    check(lib.egv_gemm_mx(M, N, K, _p(aq), _p(asc), _p(bq), _p(bsc), _p(out), N, _p(bias), act, _p(res1), _p(pre), _p(aux), dact, N,
                          _p(oq), _p(osc), _st()), 'egv_gemm_mx')
    return (out, oq, osc) if quant_out else out


def wgrad(dy, x, M, N, K, gate=None, scale=1.0, ldy=None, bias=False):
    """dW[N,K] fp32 = scale*gate * dy[M,N]^T x[M,K]  (and, with bias=True, db[N] = scale*gate * colsum(dy) from the same pass)"""
    dw = torch.empty(N, K, dtype=torch.float32, device=dy.device)
    db = torch.empty(N, dtype=torch.float32, device=dy.device) if bias else None
    nb = lib.egv_gemm_wgrad_workspace_bytes(N, K, M)
    ws = workspace(nb, dy.device)
    check(lib.egv_gemm_wgrad(_dt(dy), M, N, K, _p(dy), N if ldy is None else ldy, _p(x), K, _p(dw), _p(db), float(scale),
                             _p(gate), _p(ws), nb, _st()), 'egv_gemm_wgrad')
    return (dw, db) if bias else dw


def wgrad_grouped(problems, M, cus=0):
    """Weight (and bias) gradients of several Linear layers over the same M tokens in ONE launch (egv_gemm_wgrad_grouped):
    problems = [(dy [M,N] bf16, x [M,K] bf16, want_bias, gate or None), ...]; cus = CUs granted to the persistent launch
    (0: all); returns [(dW [N,K] fp32, db [N] fp32 or None), ...]"""
    n = len(problems)
    arr = (L.WgradProblem * n)()
    outs = []
    for i, (dy, x, bias, gate) in enumerate(problems):
        N, K = dy.shape[1], x.shape[1]
        dw = torch.empty(N, K, dtype=torch.float32, device=dy.device)
        db = torch.empty(N, dtype=torch.float32, device=dy.device) if bias else None
        arr[i].dy, arr[i].ldy, arr[i].x, arr[i].ldx = _p(dy), dy.stride(0), _p(x), x.stride(0)
        arr[i].dw, arr[i].db, arr[i].gate, arr[i].N, arr[i].K = _p(dw), _p(db), _p(gate), N, K
        outs.append((dw, db))
    nb = lib.egv_gemm_wgrad_grouped_workspace_bytes(M, n, arr, cus)
    if nb < 0:
        raise L.EgvError("egv_gemm_wgrad_grouped: unsupported group")
    ws = workspace(nb, problems[0][0].device, slot=3)
    check(lib.egv_gemm_wgrad_grouped(L.EGV_BF16, M, n, arr, cus, _p(ws), nb, _st()), 'egv_gemm_wgrad_grouped')
    return outs


def colsum(dy, M, N, gate=None, scale=1.0, ld=None):
    out = torch.empty(N, dtype=torch.float32, device=dy.device)
    ws = workspace(lib.egv_colsum_workspace_bytes(M, N), dy.device)
    check(lib.egv_colsum(_dt(dy), _p(dy), M, N, N if ld is None else ld, _p(out), float(scale), _p(gate), _p(ws), _st()),
          'egv_colsum')
    return out


def dot(a, b):
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    ws = workspace(4096, a.device)
    check(lib.egv_dot(_dt(a), _p(a), _p(b), a.numel(), _p(out), 1.0, _p(ws), _st()), 'egv_dot')
    return out


# ---- companion streams ------------------------------------------------------------------------------------
_wg_streams = {}


def companion_stream(device, kind='wgrad'):
    """A stream for work that runs beside the calling stream (kind 'wgrad': weight gradients, 'text': the text tower).  HIP
    priority EGV_SIDE_PRIORITY (default 1 = low: the companions' workgroups only take CUs the calling stream's kernels leave free;
    0 = a plain torch stream); EGV_TEXT_PRIORITY overrides it for the text stream."""
    prio = int(SW.value('EGV_SIDE_PRIORITY'))
    if kind == 'text' and SW.value('EGV_TEXT_PRIORITY') != '':
        prio = int(SW.value('EGV_TEXT_PRIORITY'))
    if prio == 0:
        return torch.cuda.Stream(device=device)
    h = C.c_void_p()
    with torch.cuda.device(device):
        check(lib.egv_stream_create(prio, C.byref(h)), 'egv_stream_create')
    return torch.cuda.ExternalStream(h.value, device=device)



def _wgrad_fork(M, fn, uses):
    """dW = dy^T x and dx = dy W of one layer are independent; run the weight-gradient GEMM(s) `fn` on a companion stream
    of the current stream so that the two grids fill each other's tail waves and prologue/epilogue bubbles (a 768x768
    layer is 594 tiles on 512 resident slots: the last wave is 16% full).  The companion starts after everything queued
    on the current stream so far; `uses` (current-stream tensors fn reads) are recorded on it.  Returns (out, join):
    join() orders the current stream after fn, to be called before the gradients are handed back to autograd.
    EGV_NO_OVERLAP=1 (or a small problem) runs fn inline."""
    if M < 4096 or SW.on('EGV_NO_OVERLAP'):
        return fn(), (lambda: None)
    cur = torch.cuda.current_stream()
    key = (cur.device.index, cur.cuda_stream)
    st = _wg_streams.get(key)
    if st is None:
        st = _wg_streams[key] = companion_stream(cur.device)
    st.wait_stream(cur)
    for t in uses:
        if t is not None:
            t.record_stream(st)
    with torch.cuda.stream(st):
        out = fn()
        done = torch.cuda.Event()
        done.record(st)

    def join():
        cur.wait_event(done)
        for t in (out if isinstance(out, (tuple, list)) else (out,)):
            if torch.is_tensor(t):
                t.record_stream(cur)
    return out, join


# ---- Linear (+bias, activation, gate, residuals) --------------------------------------------------------
class LinearFn(Function):
    """y = gate * act(x W^T + b) + res1 + res2   (gate requires act == none; its pre-gate value is saved)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gate, res1, res2, act):
        _need_gpu(x)
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M, K = x2.shape
        N = weight.shape[0]
        w = compute_weight(weight, x.dtype)
        y = torch.empty(M, N, dtype=x.dtype, device=x.device)
        pre = torch.empty_like(y) if (gate is not None or act == L.ACT_GELU) else None
        assert not (gate is not None and act != L.ACT_NONE), "gate requires act == none"
        r1 = res1.reshape(M, N).contiguous() if res1 is not None else None
        r2 = res2.reshape(M, N).contiguous() if res2 is not None else None
        gemm(x2, w, y, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, act=act, gate=gate, res1=r1, res2=r2, pre=pre)
        ctx.act = act
        ctx.dims = (M, N, K)
        ctx.xshape = shp
        ctx.has = (bias is not None, gate is not None, res1 is not None, res2 is not None)
        ctx.save_for_backward(x2, weight, gate, pre, y if act in (L.ACT_RELU, L.ACT_TANH) else None)
        return y.reshape(*shp[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, gate, pre, yact = ctx.saved_tensors
        M, N, K = ctx.dims
        has_b, has_g, has_r1, has_r2 = ctx.has
        dy2 = dy.reshape(M, N)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dz = dy2
        if ctx.act in (L.ACT_RELU, L.ACT_TANH):
            dz = torch.empty_like(dy2)
            check(lib.egv_act_bwd(_dt(dy2), _p(dy2), _p(yact), _p(dz), dy2.numel(), ctx.act, _st()), 'egv_act_bwd')
        elif ctx.act == L.ACT_GELU:
            dz = torch.empty_like(dy2)
            check(lib.egv_act_bwd(_dt(dy2), _p(dy2), _p(pre), _p(dz), dy2.numel(), L.ACT_GELU, _st()), 'egv_act_bwd')
        dx = dw = db = dg = None
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_b and ctx.needs_input_grad[2]

        def weight_grads():
            if need_w:
                return wgrad(dz, x2, M, N, K, gate=gate, bias=True) if need_b else (wgrad(dz, x2, M, N, K, gate=gate), None)
            return None, (colsum(dz, M, N, gate=gate) if need_b else None)
        (dw, db), join = _wgrad_fork(M if (need_x and need_w) else 0, weight_grads, (dz, x2))
        if need_x:
            dx = torch.empty(M, K, dtype=dy2.dtype, device=dy2.device)
            dgrad(dz, weight, dx, M, N, K, gate=gate)
            dx = dx.reshape(ctx.xshape)
        if has_g and ctx.needs_input_grad[3]:
            dg = dot(dy2, pre)
        join()
        dr1 = dy if has_r1 and ctx.needs_input_grad[4] else None
        dr2 = dy if has_r2 and ctx.needs_input_grad[5] else None
        return dx, dw, db, dg, dr1, dr2, None


def linear(x, weight, bias=None, act='none', gate=None, res1=None, res2=None):
    return LinearFn.apply(x, weight, bias, gate, res1, res2, ACT[act])


class MlpFn(Function):
    """y = res + W2 gelu(W1 x + b1) + b2  -- Mlp / intermediate+output of video_transformer.py:42-58, roberta.py:397-426.
    GELU' is fused into the fc2 dgrad epilogue; pre-activation and activation are saved."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, res):
        _need_gpu(x)
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M, K = x2.shape
        Hd = w1.shape[0]
        N = w2.shape[0]
        c1, c2 = compute_weight(w1, x.dtype), compute_weight(w2, x.dtype)
        pre = torch.empty(M, Hd, dtype=x.dtype, device=x.device)
        h = torch.empty_like(pre)
        gemm(x2, c1, h, M=M, N=Hd, K=K, lda=K, ldb=K, ldc=Hd, bias=b1, act=L.ACT_GELU, pre=pre)
        y = torch.empty(M, N, dtype=x.dtype, device=x.device)
        r = res.reshape(M, N).contiguous() if res is not None else None
        gemm(h, c2, y, M=M, N=N, K=Hd, lda=Hd, ldb=Hd, ldc=N, bias=b2, res1=r)
        ctx.dims = (M, K, Hd, N)
        ctx.xshape = shp
        ctx.has_res = res is not None
        ctx.save_for_backward(x2, w1, w2, pre, h)
        return y.reshape(*shp[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w2, pre, h = ctx.saved_tensors
        M, K, Hd, N = ctx.dims
        dy2 = dy.reshape(M, N)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        (dw2, db2), join2 = _wgrad_fork(M, lambda: wgrad(dy2, h, M, N, Hd, bias=True), (dy2, h))
        dpre = torch.empty(M, Hd, dtype=dy2.dtype, device=dy2.device)
        dgrad(dy2, w2, dpre, M, N, Hd, aux=pre, dact=L.ACT_GELU)
        need_x = ctx.needs_input_grad[0]
        (dw1, db1), join1 = _wgrad_fork(M if need_x else 0, lambda: wgrad(dpre, x2, M, Hd, K, bias=True), (dpre, x2))
        dx = None
        if need_x:
            dx = torch.empty(M, K, dtype=dy2.dtype, device=dy2.device)
            dgrad(dpre, w1, dx, M, Hd, K)
            dx = dx.reshape(ctx.xshape)
        join2()
        join1()
        return dx, dw1, db1, dw2, db2, (dy if ctx.has_res else None)


def mlp(x, w1, b1, w2, b2, res=None):
    return MlpFn.apply(x, w1, b1, w2, b2, res)


class VocabLinearFn(Function):
    """MLM decoder (heads.py:44-50): logits = x W^T + bias with the vocabulary axis padded to a multiple of 128
    columns (row pitch stays 16-byte aligned; columns >= V are never read and get zero gradient)."""

    @staticmethod
    def forward(ctx, x, weight, bias, V):
        _need_gpu(x)
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        M, K = x2.shape
        Vp = (V + 127) // 128 * 128
        w = compute_weight(weight, x.dtype)
        y = torch.empty(M, Vp, dtype=x.dtype, device=x.device)
        gemm(x2, w, y, M=M, N=V, K=K, lda=K, ldb=K, ldc=Vp, bias=bias)
        ctx.dims = (M, K, V, Vp)
        ctx.xshape = x.shape
        ctx.save_for_backward(x2, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        M, K, V, Vp = ctx.dims
        dy = dy.contiguous()
        w = compute_weight(weight, dy.dtype)
        dx = torch.empty(M, K, dtype=dy.dtype, device=dy.device)
        if dy.dtype == torch.bfloat16 and M % 8 == 0:
            # dx = dy W reduces over the 50265 vocabulary entries with only M x K = 256 x 768 outputs: 12 tiles for a plain
            # GEMM.  The split-K weight-gradient kernel is the right shape for it: dx^T[K,M] = W^T[K,V] dy^T[V,M] is a
            # "weight gradient" with W as the row operand and dy^T as the column operand, reduced over V in ~40 slabs.
            dyT = torch.empty(V, M, dtype=dy.dtype, device=dy.device)
            check(lib.egv_transpose(L.EGV_BF16, L.EGV_BF16, _p(dy), _p(dyT), M, V, Vp, _st()), 'egv_transpose(dy)')
            dxT = wgrad(w, dyT, V, K, M)
            check(lib.egv_transpose(L.EGV_F32, L.EGV_BF16, _p(dxT), _p(dx), K, M, M, _st()), 'egv_transpose(dx)')
        else:
            gemm(dy, w, dx, a_trans=0, b_trans=1, M=M, N=K, K=V, lda=Vp, ldb=K, ldc=K)
        dw, db = wgrad(dy, x2, M, V, K, ldy=Vp, bias=True)
        return dx.reshape(ctx.xshape), dw, db, None


def vocab_linear(x, weight, bias, V):
    return VocabLinearFn.apply(x, weight, bias, V)


# ---- LayerNorm ---------------------------------------------------------------------------------------
class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _need_gpu(x)
        shp = x.shape
        D = shp[-1]
        x2 = x.reshape(-1, D)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M = x2.shape[0]
        y = torch.empty_like(x2)
        stats = torch.empty(M, 2, dtype=torch.float32, device=x.device)
        check(lib.egv_layernorm_fwd(_dt(x2), _p(x2), _p(y), _p(gamma), _p(beta), _p(stats), M, D, float(eps), _st()),
              'egv_layernorm_fwd')
        ctx.save_for_backward(x2, gamma, stats)
        ctx.xshape = shp
        return y.reshape(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, gamma, stats = ctx.saved_tensors
        M, D = x2.shape
        dy2 = dy.reshape(M, D)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dx = torch.empty_like(x2)
        gb = torch.empty(2, D, dtype=torch.float32, device=x2.device)     # [dgamma ; dbeta]: one reduction launch
        dg, db = gb[0], gb[1]
        ws = workspace(lib.egv_layernorm_bwd_workspace_bytes(M, D), x2.device)
        check(lib.egv_layernorm_bwd(_dt(x2), _p(dy2), _p(x2), _p(stats), _p(gamma), None, _p(dx), _p(dg), _p(db), M, D,
                                    _p(ws), _st()), 'egv_layernorm_bwd')
        return dx.reshape(ctx.xshape), dg, db, None


def layernorm(x, gamma, beta, eps):
    return LayerNormFn.apply(x, gamma, beta, eps)


class LayerNormSkipFn(Function):
    """(LayerNorm(x), x) for the pre-norm residual pattern  out = x + f(LayerNorm(x)):  the second output is x itself, to
    be used for the skip connection.  Both gradients then arrive at this one node and the LayerNorm backward kernel adds the
    skip gradient while it writes dx -- instead of autograd launching a separate full-size add."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _need_gpu(x)
        assert x.dim() == 2 and x.is_contiguous()
        M, D = x.shape
        y = torch.empty_like(x)
        stats = torch.empty(M, 2, dtype=torch.float32, device=x.device)
        check(lib.egv_layernorm_fwd(_dt(x), _p(x), _p(y), _p(gamma), _p(beta), _p(stats), M, D, float(eps), _st()),
              'egv_layernorm_fwd')
        ctx.save_for_backward(x, gamma, stats)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        x2, gamma, stats = ctx.saved_tensors
        M, D = x2.shape
        if dy is None:                                   # only the skip path was used
            return dskip, None, None, None
        dy2 = dy.contiguous()
        add = None if dskip is None else dskip.contiguous()
        dx = torch.empty_like(x2)
        gb = torch.empty(2, D, dtype=torch.float32, device=x2.device)
        dg, db = gb[0], gb[1]
        ws = workspace(lib.egv_layernorm_bwd_workspace_bytes(M, D), x2.device)
        check(lib.egv_layernorm_bwd(_dt(x2), _p(dy2), _p(x2), _p(stats), _p(gamma), _p(add), _p(dx), _p(dg), _p(db), M, D,
                                    _p(ws), _st()), 'egv_layernorm_bwd')
        return dx, dg, db, None


def layernorm_skip(x, gamma, beta, eps):
    return LayerNormSkipFn.apply(x, gamma, beta, eps)


# ---- attention ---------------------------------------------------------------------------------------
def _rowset(bs, base, gs, istride, n):
    return (int(bs), int(base), int(gs), int(istride), int(n))


def _mk_desc(Q, K, V, O, lse, B, G, H, qset, kset, extra, scale, mask=None, dO=None, dQ=None, dK=None, dV=None, delta=None,
             nsplit=1, ws=None, ws_bytes=0, drop_p=0.0, drop_seed=0):
    d = AttnDesc()
    d.drop_p, d.drop_seed = float(drop_p), int(drop_seed) & 0xFFFFFFFF
    d.Q, d.K, d.V, d.O = _p(Q), _p(K), _p(V), _p(O)
    d.dO, d.dQ, d.dK, d.dV = _p(dO), _p(dQ), _p(dK), _p(dV)
    d.ldq, d.ldk, d.ldv, d.ldo = Q.stride(0), K.stride(0), V.stride(0), O.stride(0)
    d.lddq = dQ.stride(0) if dQ is not None else 0
    d.lddk = dK.stride(0) if dK is not None else 0
    d.lddv = dV.stride(0) if dV is not None else 0
    d.qoff = d.koff = d.voff = d.ooff = d.dqoff = d.dkoff = d.dvoff = 0
    d.lse, d.delta = _p(lse), _p(delta)
    d.B, d.G, d.H = B, G, H
    d.q_bs, d.q_base, d.q_gs, d.q_is, d.q_n = qset
    d.k_bs, d.k_base, d.k_gs, d.k_is, d.k_n = kset
    if extra is None:
        d.extra, d.extra_bs, d.extra_row = 0, 0, 0
    else:
        d.extra, d.extra_bs, d.extra_row = 1, int(extra[0]), int(extra[1])
    d.scale = float(scale)
    d.mask = _p(mask)
    d.mask_ld = mask.stride(0) if mask is not None else 0
    d.nsplit = nsplit
    d.ws = _p(ws)
    d.ws_bytes = ws_bytes
    return d


def _dkv_ws(B, G, H, k_n, nsplit, device):
    nb = lib.egv_attn_bwd_dkv_workspace_bytes(B, G, H, k_n, nsplit)
    return (workspace(nb, device, slot=1), nb) if nb else (None, 0)


def _split_ws(which, B, G, H, n_own, nsplit, device):
    nb = lib.egv_attn_split_workspace_bytes(which, B, G, H, n_own, nsplit)
    return (workspace(nb, device, slot=1), nb) if nb else (None, 0)


def _nsplit_for(n_other):
    """split count for launches whose other side is too long for one workgroup / the MFMA kernels (<= 224 rows)"""
    return 1 if n_other <= 224 else max(2, min(32, n_other // 384))     # 8 splits at S = 3137: the combine pass stays short


FUSED_ATTN_BWD = SW.on('EGV_ATTN_FUSED_BWD')
FUSED_ATTN_CLS = SW.on('EGV_ATTN_FUSED_CLS')     # the one-pass kernel also produces the CLS row's gradients


class DividedAttnFn(Function):
    """Divided space / time attention core of VarAttention (video_transformer.py:121-150) on the fused qkv buffer
    [B*S, 3*D]: CLS query over all S keys, patch queries over [CLS ; own frame | own patch column]."""

    @staticmethod
    def forward(ctx, qkv, B, Fr, N, H, mode):
        _need_gpu(qkv)
        S = 1 + Fr * N
        D = H * 64
        M = B * S
        assert qkv.shape == (M, 3 * D) and qkv.is_contiguous()
        Q, K, V = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        O = torch.empty(M, D, dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty(M, H, dtype=torch.float32, device=qkv.device)
        scale = 64 ** -0.5
        if mode == 'space':
            G, gset = Fr, _rowset(S, 1, N, 1, N)
        else:
            G, gset = N, _rowset(S, 1, 1, N, Fr)
        dt = _dt(qkv)
        d1 = _mk_desc(Q, K, V, O, lse, B, G, H, gset, gset, (S, 0), scale)
        covers = False
        if FUSED_ATTN_CLS:                       # the group launch can also serve the CLS query (per-group partial softmax states)
            nbx = lib.egv_attn_fwd_extra_workspace_bytes(B, G, H)
            wsx = torch.empty(nbx // 4, dtype=torch.float32, device=qkv.device)
            dx = _mk_desc(Q, K, V, O, lse, B, G, H, gset, gset, (S, 0), scale, ws=wsx, ws_bytes=nbx)
            covers = lib.egv_attn_fwd_covers_extra(dt, C.byref(dx)) == 1
            if covers:
                d1 = dx
        check(lib.egv_attn_fwd(dt, C.byref(d1), _st()), 'egv_attn_fwd(groups)')
        if covers:
            ctx.cfg = (B, Fr, N, H, mode)
            ctx.save_for_backward(qkv, O, lse)
            return O
        ns = _nsplit_for(S)
        ws, nb = _split_ws(0, B, 1, H, 1, ns, qkv.device)
        d2 = _mk_desc(Q, K, V, O, lse, B, 1, H, _rowset(S, 0, 0, 1, 1), _rowset(S, 0, 0, 1, S), None, scale, nsplit=ns, ws=ws,
                      ws_bytes=nb)
        check(lib.egv_attn_fwd(dt, C.byref(d2), _st()), 'egv_attn_fwd(cls)')
        ctx.cfg = (B, Fr, N, H, mode)
        ctx.save_for_backward(qkv, O, lse)
        return O

    @staticmethod
    def backward(ctx, dO):
        qkv, O, lse = ctx.saved_tensors
        B, Fr, N, H, mode = ctx.cfg
        S = 1 + Fr * N
        D = H * 64
        M = B * S
        dO = dO.contiguous()
        Q, K, V = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        dqkv = torch.empty_like(qkv)
        dQ, dK, dV = dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:]
        delta = torch.empty(M, H, dtype=torch.float32, device=qkv.device)
        scale = 64 ** -0.5
        if mode == 'space':
            G, gset = Fr, _rowset(S, 1, N, 1, N)
        else:
            G, gset = N, _rowset(S, 1, 1, N, Fr)
        cls1, allS = _rowset(S, 0, 0, 1, 1), _rowset(S, 0, 0, 1, S)
        dt = _dt(qkv)
        kw = dict(dO=dO, dQ=dQ, dK=dK, dV=dV, delta=delta)
        # groups: dQ, dK, dV (and delta) in one kernel where the shape allows it (bf16 space attention); given a workspace it also
        # produces the CLS row's three gradients, and nothing else is launched
        rc = 1
        if FUSED_ATTN_BWD:
            nbf = lib.egv_attn_bwd_fused_workspace_bytes(B, G, H)
            wsf = torch.empty(nbf // 4, dtype=torch.float32, device=qkv.device) if FUSED_ATTN_CLS else None
            d1 = _mk_desc(Q, K, V, O, lse, B, G, H, gset, gset, (S, 0), scale, ws=wsf, ws_bytes=nbf if wsf is not None else 0, **kw)
            rc = lib.egv_attn_bwd_fused(dt, C.byref(d1), _st())
            if rc not in (0, 1):
                check(rc, 'egv_attn_bwd_fused(groups)')
            if rc == 0 and wsf is not None:
                return dqkv, None, None, None, None, None
            if rc == 1 and wsf is not None and lib.egv_attn_bwd_pair_covers_extra(dt, C.byref(d1)) == 1:
                # one-tile groups (time attention): the kernel pair leaves the CLS row's gradients as partials, one small sum
                check(lib.egv_attn_bwd_dq(dt, C.byref(d1), _st()), 'egv_attn_bwd_dq(groups)')
                check(lib.egv_attn_bwd_dkv(dt, C.byref(d1), _st()), 'egv_attn_bwd_dkv(groups)')
                check(lib.egv_attn_bwd_extra_reduce(dt, C.byref(d1), 1, _st()), 'egv_attn_bwd_extra_reduce')
                return dqkv, None, None, None, None, None
        ns = _nsplit_for(S)
        ws, nb = _split_ws(1, B, 1, H, 1, ns, qkv.device)
        d2 = _mk_desc(Q, K, V, O, lse, B, 1, H, cls1, allS, None, scale, nsplit=ns, ws=ws, ws_bytes=nb, **kw)
        check(lib.egv_attn_bwd_dq(dt, C.byref(d2), _st()), 'egv_attn_bwd_dq(cls)')
        if rc == 1:                                  # query-owned dQ, then key-owned group keys <- [CLS query ; group queries]
            d1 = _mk_desc(Q, K, V, O, lse, B, G, H, gset, gset, (S, 0), scale, **kw)
            check(lib.egv_attn_bwd_dq(dt, C.byref(d1), _st()), 'egv_attn_bwd_dq(groups)')
            check(lib.egv_attn_bwd_dkv(dt, C.byref(d1), _st()), 'egv_attn_bwd_dkv(groups)')
        # CLS key <- all S queries (split + reduce)
        nsplit = _nsplit_for(S)
        ws, nb = _dkv_ws(B, 1, H, 1, nsplit, qkv.device)
        d4 = _mk_desc(Q, K, V, O, lse, B, 1, H, allS, cls1, None, scale, nsplit=nsplit, ws=ws, ws_bytes=nb, **kw)
        check(lib.egv_attn_bwd_dkv(dt, C.byref(d4), _st()), 'egv_attn_bwd_dkv(cls)')
        return dqkv, None, None, None, None, None


def divided_attention(qkv, B, Fr, N, H, mode):
    return DividedAttnFn.apply(qkv, B, Fr, N, H, mode)


class PlainAttnFn(Function):
    """softmax(scale * q k^T + mask) v per (batch, head): RoBERTa self attention (roberta.py:257-327), text->image
    cross attention (same class, keys = video tokens, no mask) and image->text cross attention
    (video_transformer.py:159-182).  q [B*nq, D], k/v [B*nk, D] may be column slices (stride(1) == 1)."""

    @staticmethod
    def forward(ctx, q, k, v, B, H, nq, nk, scale, mask, dkv_nsplit, drop_p=0.0, drop_seed=0):
        _need_gpu(q)
        D = H * 64
        for t in (q, k, v):
            assert t.dim() == 2 and t.stride(1) == 1 and t.shape[1] == D
        O = torch.empty(B * nq, D, dtype=q.dtype, device=q.device)
        lse = torch.empty(B * nq, H, dtype=torch.float32, device=q.device)
        ns = 1 if nk <= 224 else (nk + 223) // 224             # MFMA kernel per 224-key chunk + combine
        ws, nb = _split_ws(0, B, 1, H, nq, ns, q.device)
        if q.dtype == torch.bfloat16 and nq <= 32 and nk >= 512 and mask is None:
            # few queries over many keys (text -> image): one streaming launch + a combination (egv_attn_cross.hip) when the workspace holds its partials
            nb = max(nb, lib.egv_attn_fewq_workspace_bytes(B, 1, H, nk))
            ws = workspace(nb, q.device, slot=1)
        d = _mk_desc(q, k, v, O, lse, B, 1, H, _rowset(nq, 0, 0, 1, nq), _rowset(nk, 0, 0, 1, nk), None, scale, mask=mask,
                     nsplit=ns, ws=ws, ws_bytes=nb, drop_p=drop_p, drop_seed=drop_seed)
        check(lib.egv_attn_fwd(_dt(q), C.byref(d), _st()), 'egv_attn_fwd')
        ctx.cfg = (B, H, nq, nk, scale, dkv_nsplit, drop_p, drop_seed)
        ctx.save_for_backward(q, k, v, O, lse, mask)
        return O

    @staticmethod
    def backward(ctx, dO):
        q, k, v, O, lse, mask = ctx.saved_tensors
        B, H, nq, nk, scale, nsplit, drop_p, drop_seed = ctx.cfg
        D = H * 64
        dO = dO.contiguous()
        dq = torch.empty(B * nq, D, dtype=q.dtype, device=q.device)
        dk = torch.empty(B * nk, D, dtype=q.dtype, device=q.device)
        dv = torch.empty(B * nk, D, dtype=q.dtype, device=q.device)
        delta = torch.empty(B * nq, H, dtype=torch.float32, device=q.device)
        qs, ks = _rowset(nq, 0, 0, 1, nq), _rowset(nk, 0, 0, 1, nk)
        kw = dict(mask=mask, dO=dO, dQ=dq, dK=dk, dV=dv, delta=delta, drop_p=drop_p, drop_seed=drop_seed)
        fewk = q.dtype == torch.bfloat16 and nk <= 32 and nq >= 128 and drop_p <= 0.0
        fewq = q.dtype == torch.bfloat16 and nq <= 32 and nk >= 512 and mask is None
        if fewk or fewq:
            # many queries over few keys (image -> text) or few queries over many keys (text -> image): dQ, dK, dV in one launch + a
            # fixed-order partial sum (egv_attn_cross.hip)
            nb = lib.egv_attn_fewkeys_workspace_bytes(B, 1, H, nq) if fewk else lib.egv_attn_fewq_workspace_bytes(B, 1, H, nk)
            ws = torch.empty(nb // 4, dtype=torch.float32, device=q.device)
            d = _mk_desc(q, k, v, O, lse, B, 1, H, qs, ks, None, scale, nsplit=1, ws=ws, ws_bytes=nb, **kw)
            rc = lib.egv_attn_bwd_fused(_dt(q), C.byref(d), _st())
            if rc != 1:
                check(rc, 'egv_attn_bwd_fused(few keys)')
                return dq, dk, dv, None, None, None, None, None, None, None, None, None
        ns = 1 if nk <= 224 else (nk + 223) // 224
        ws, nb = _split_ws(1, B, 1, H, nq, ns, q.device)
        d = _mk_desc(q, k, v, O, lse, B, 1, H, qs, ks, None, scale, nsplit=ns, ws=ws, ws_bytes=nb, **kw)
        check(lib.egv_attn_bwd_dq(_dt(q), C.byref(d), _st()), 'egv_attn_bwd_dq')
        nsplit = max(nsplit, 1 if nq <= 224 else (nq + 223) // 224)      # MFMA kernel per 224-query chunk
        ws, nb = _dkv_ws(B, 1, H, nk, nsplit, q.device)
        d = _mk_desc(q, k, v, O, lse, B, 1, H, qs, ks, None, scale, nsplit=nsplit, ws=ws, ws_bytes=nb, **kw)
        check(lib.egv_attn_bwd_dkv(_dt(q), C.byref(d), _st()), 'egv_attn_bwd_dkv')
        return dq, dk, dv, None, None, None, None, None, None, None, None, None


def plain_attention(q, k, v, B, H, nq, nk, scale, mask=None, dkv_nsplit=1, drop_p=0.0, drop_seed=0):
    """drop_p > 0: dropout on the softmax probabilities (roberta.py:313); the keep mask is a pure function of
    (drop_seed, batch, head, query, key), so the backward kernels regenerate it instead of storing it."""
    return PlainAttnFn.apply(q, k, v, B, H, nq, nk, scale, mask, dkv_nsplit, drop_p, drop_seed)


class DropoutAddFn(Function):
    """y = dropout(x, p) + r1 + r2 in one pass (RobertaSelfOutput / RobertaOutput: dense -> dropout -> residual add,
    roberta.py:342, :422, :486-488; embeddings dropout :203 with no residual).  Counter-based mask: element i is kept iff
    hash(seed, i) >= p, so backward re-derives the mask from the seed."""

    @staticmethod
    def forward(ctx, x, r1, r2, p, seed):
        _need_gpu(x)
        x = x.contiguous()
        for r in (r1, r2):
            assert r is None or (r.shape == x.shape and r.dtype == x.dtype and r.is_contiguous())
        y = torch.empty_like(x)
        check(lib.egv_dropout_add(_dt(x), _p(x), _p(r1), _p(r2), _p(y), x.numel(), float(p), int(seed) & 0xFFFFFFFF, _st()),
              'egv_dropout_add')
        ctx.cfg = (p, seed, r1 is not None, r2 is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed, has1, has2 = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        check(lib.egv_dropout_add(_dt(dy), _p(dy), None, None, _p(dx), dy.numel(), float(p), int(seed) & 0xFFFFFFFF, _st()),
              'egv_dropout_add(bwd)')
        return dx, (dy if has1 else None), (dy if has2 else None), None, None


def dropout_add(x, p, seed, r1=None, r2=None):
    return DropoutAddFn.apply(x, r1, r2, p, seed)


# ---- block-level calls (csrc/egv_block.cpp): one C-ABI call per SpaceTimeBlock / RobertaLayer and direction -----------------
def _side_stream_ptr():
    """raw handle of the weight-gradient companion stream of the current stream (None: single-stream mode)"""
    if SW.on('EGV_NO_OVERLAP'):
        return None
    cur = torch.cuda.current_stream()
    key = (cur.device.index, cur.cuda_stream)
    if any(k[0] == key[0] and v.cuda_stream == key[1] for k, v in _wg_streams.items()):
        return None                                         # the current stream IS a companion (work forked onto it): no companion of a companion
    st = _wg_streams.get(key)
    if st is None:
        st = _wg_streams[key] = companion_stream(cur.device)
    return st.cuda_stream


def weight_gradient_stream_of(stream):
    """the weight-gradient companion of `stream` (created on first use) -- model.py forks the MLM pass's top onto the text stream's"""
    key = (stream.device.index, stream.cuda_stream)
    st = _wg_streams.get(key)
    if st is None:
        st = _wg_streams[key] = companion_stream(stream.device)
    return st


def _fill_weights(d, weights, dtype):
    for i, w in enumerate(weights):
        d.w[i] = _p(compute_weight(w, dtype))
        wt = compute_weight_t(w, dtype)
        d.wt[i] = _p(wt) if wt is not None else None


class _GradPack:
    """one flat fp32 buffer holding the gradients of every parameter of a block call; views are handed to autograd"""

    def __init__(self, params, device, order=None, reuse=None):
        """order: the parameter indices in the order they are laid out in the flat buffer (default: as listed) -- weights whose
        gradients one GEMM writes as one matrix (merged query / key / value) are made adjacent this way; views stay in list order.
        reuse: {parameter index: view of an EARLIER call's buffer} -- those gradients are accumulated into that view by the grouped
        weight-gradient launch itself (egv_wgrad_problem::accumulate) and get no storage here; `flat` then holds the others only."""
        sizes = [p.numel() for p in params]
        reuse = reuse or {}
        self.layout = list(order if order is not None else range(len(params)))
        self.flat = torch.empty(sum(sizes[i] for i in self.layout if i not in reuse), dtype=torch.float32, device=device)
        self.views = [None] * len(params)
        self.offset = {}                                   # parameter index -> element offset inside self.flat
        off = 0
        for i in self.layout:
            if i in reuse:
                self.views[i] = reuse[i]
                continue
            self.views[i] = self.flat[off:off + sizes[i]].view(params[i].shape)
            self.offset[i] = off
            off += sizes[i]


# Parameter gradients of a block that is used several times per step (EgoNCE / MLM / ITM passes share the backbones): every
# backward call writes its own flat pack; calls after the first add theirs to the first one with ONE flat add, and only the call
# that completes the step's uses hands gradients to autograd -- instead of autograd accumulating ~27 tensors per block call
# with one small ATen add each (809 launches per step).  AccumulateGrad (and with it DDP's reducer hook) then fires once per
# parameter and step.  If a backward pass ends with uses outstanding (partial graphs), the pending sums are flushed into .grad.
_acc = {}
_acc_cb = [False]
_pack_hook = [None]
_defer_allowed = [True]
_deferred = {'sides': {}, 'handed': []}


def set_pack_hook(fn):
    """fn(flat, params) is called -- inside backward, with the stream current on which the buffer is complete (the calling
    stream of the block, or its weight-gradient companion when the block's join is deferred) -- when the flat fp32 gradient
    buffer of a block holds the sum over all of the step's uses of that block, just before its views go to autograd
    (trainer/grad_sync.py starts the data-parallel all-reduce of the buffer there).  None removes the hook."""
    _pack_hook[0] = fn


def set_defer_wgrad_join(on: bool):
    """Deferred join (default on): a video block's backward call returns while its grouped weight-gradient launch is still
    running on the companion stream; the calling stream is joined ONCE, at the end of the backward pass.  Only taken when
    nothing can read the gradients before that: parameters enter backward with .grad None and without tensor hooks, and the
    model is not running under DistributedDataParallel (whose reducer copies each gradient when it is produced) --
    FrozenInTime.forward switches it off there."""
    _defer_allowed[0] = bool(on)


def _acc_forward(key, track, fused):
    """key identifies the block; its parameters split into the SHARED set (used by the fused and the unfused form of the block)
    and the EXTRA set (cross-attention parameters, fused form only): each has its own use count.  track: the call was made with
    grad mode on and a parameter requires grad (decided by the caller of .apply: inside Function.forward grad mode is off and
    needs_input_grad is True for parameters even under torch.no_grad())."""
    if track:
        for k in ((key, 0), (key, 1)) if fused else ((key, 0),):
            e = _acc.setdefault(k, {'uses': 0, 'done': 0, 'flat': None, 'views': None, 'params': None, 'side': None, 'stream': None, 'event': None, 'small0': None})
            e['uses'] += 1


def _backward_done():
    """end of a backward pass (autograd engine callback): join the deferred weight-gradient streams, flush block gradients whose
    remaining uses did not take part in this pass, check that autograd kept the views it was handed."""
    _acc_cb[0] = False
    cur = torch.cuda.current_stream() if torch.cuda.is_available() else None
    for side, main in _deferred['sides'].values():
        main.wait_stream(side)
        if cur is not None and cur.cuda_stream != main.cuda_stream:
            cur.wait_stream(side)
    _deferred['sides'].clear()
    handed, _deferred['handed'] = _deferred['handed'], []
    pending = [k for k, e in _acc.items() if e['done'] and e['flat'] is not None]
    if pending and _pack_hook[0] is not None:
        _acc.clear()
        raise RuntimeError("egovlpv2_amd: a backward pass ended with block uses outstanding while a gradient-sync hook is active "
                           "(the flat per-block buffers would bypass the data-parallel reduction); run backward on the full loss")
    for key in pending:
        e = _acc[key]
        with torch.no_grad():
            for p, g in zip(e['params'], e['views']):
                if p.requires_grad:
                    p.grad = g.clone() if p.grad is None else p.grad.add_(g)
    # uses that did not come back in this pass belong to graph parts that may run in a later backward call: those calls find no
    # entry and hand their gradients to autograd tensor by tensor (the plain path)
    _acc.clear()
    for params, ptrs in handed:
        for p, ptr in zip(params, ptrs):
            g = p.grad
            if g is not None and g.data_ptr() != ptr:
                raise RuntimeError("egovlpv2_amd: autograd copied a block gradient before the deferred weight-gradient launch had "
                                   "finished (a hook or an existing .grad on a parameter?); call hipops.set_defer_wgrad_join(False)")


def _queue_done():
    if not _acc_cb[0]:
        _acc_cb[0] = True
        torch.autograd.Variable._execution_engine.queue_callback(_backward_done)


def _acc_part(k, flat, views, params, side=None, tail_only=False, small0=None):
    """side: (companion stream, calling stream) when this call's weight gradients are still running on the companion.
    small0 (first use): (element offset inside `flat` where the gradients that no grouped launch writes begin, the parameter indices before
    it), or None when this call had no grouped launch / another layout.
    tail_only (a later use under _acc_reuse): `flat` holds only that tail -- the launch has already added the rest into the first
    use's buffer."""
    e = _acc.get(k)
    if e is None:                                             # no bookkeeping for this use (earlier pass flushed it): plain path
        assert not tail_only
        if side is not None:                                  # nothing may touch the views before the join: join now
            side[1].wait_stream(side[0])
        return views
    cur = torch.cuda.current_stream()
    if e['flat'] is None:
        assert not tail_only
        e['flat'], e['views'], e['params'], e['side'], e['small0'] = flat, views, params, side, small0
    elif tail_only:
        dst = e['flat'][e['small0'][0]:]
        assert dst.numel() == flat.numel(), (dst.numel(), flat.numel())
        sd = side or e['side']
        if flat.numel():
            if sd is not None:                                # (the text projections of a fused video block write their gradients on the companion)
                sd[0].wait_stream(cur)
                with torch.cuda.stream(sd[0]):
                    dst.add_(flat)
                flat.record_stream(sd[0])
            else:
                dst.add_(flat)
        e['side'] = sd
    else:
        if e['stream'] != cur.cuda_stream:
            # the uses of one block ran on different streams (the model keeps a tower on one stream, so this is the exception):
            # order this stream after the previous contribution and keep the allocator informed
            cur.wait_event(e['event'])
            e['flat'].record_stream(cur)
        sd = side or e['side']
        if sd is not None:                                    # one of the two buffers is still being written on the companion
            sd[0].wait_stream(cur)
            with torch.cuda.stream(sd[0]):
                e['flat'].add_(flat)
            flat.record_stream(sd[0])
            e['flat'].record_stream(sd[0])
            e['side'] = sd
        else:
            e['flat'].add_(flat)
    e['stream'] = cur.cuda_stream
    if e['done'] + 1 < e['uses']:
        e['event'] = torch.cuda.Event()
        e['event'].record(cur if e['side'] is None else e['side'][0])
    e['done'] += 1
    if e['done'] >= e['uses']:
        del _acc[k]
        sd = e['side']
        if _pack_hook[0] is not None:
            if sd is not None:
                sd[0].wait_stream(cur)                        # (the LayerNorm / gate gradients of the buffer were written on the calling stream)
                with torch.cuda.stream(sd[0]):
                    _pack_hook[0](e['flat'], e['params'])
            else:
                _pack_hook[0](e['flat'], e['params'])
        if sd is not None:
            _deferred['handed'].append((e['params'], [v.data_ptr() for v in e['views']]))   # addresses only: a live reference to a view would make AccumulateGrad copy it
        return e['views']
    _queue_done()
    return [None] * len(params)


def _acc_reuse(key, params, nshared, layout, acc_params, side):
    """beta = 1 plan of a backward call (EGV_WGRAD_ACC): for each part of the block's parameters (0: those every use of the block has,
    1: the fusion extras) whose FIRST use of this step has already run on this stream, the gradients the grouped weight-gradient launch
    forms (`acc_params`: parameter indices) are accumulated into that use's buffer inside the launch -- no second 20-57 MB buffer, no
    flat add over it.  Returns ({parameter index: first use's view}, [part is accumulating]).  What the launch does not write
    (LayerNorm / gate gradients, the projections over the other modality's rows: < 1 % of the bytes unfused) must be the TAIL of the
    part's layout, so that one small add covers it; a part whose layout does not have that form is left on the flat-add path."""
    reuse, on = {}, [False, False]
    if not acc_params or not SW.on('EGV_WGRAD_ACC'):
        return reuse, on
    cur = torch.cuda.current_stream().cuda_stream
    for part, (lo, hi) in enumerate(((0, nshared), (nshared, len(params)))):
        e = _acc.get((key, part))
        if hi <= lo or e is None or e['flat'] is None or e['stream'] != cur or e.get('small0') is None:
            continue
        if e['small0'][1] != frozenset(i for i in acc_params if lo <= i < hi):
            continue                                        # the first use formed other gradients in its launch (another batch size: another form)
        es, ss = e['side'], side
        if (es is None) != (ss is None) or (es is not None and es[0].cuda_stream != ss[0].cuda_stream):
            continue                                        # both launches must sit on ONE stream (the companion, or the calling stream)
        reuse.update({i: e['views'][i - lo] for i in range(lo, hi) if i in acc_params})
        on[part] = True
    return reuse, on


def _part_small0(layout, lo, hi, acc_params):
    """number of accumulating parameters at the head of a part's layout if the others form its tail, else None"""
    idx = [i for i in layout if lo <= i < hi]
    n = 0
    while n < len(idx) and idx[n] in acc_params:
        n += 1
    return n if all(i not in acc_params for i in idx[n:]) else None


def _acc_backward(key, pack, params, nshared, side=None, acc_params=(), acc_on=(False, False)):
    out = []
    for part, (lo, hi) in enumerate(((0, nshared), (nshared, len(params)))):
        if hi <= lo:
            continue
        pp, vv = params[lo:hi], pack.views[lo:hi]
        idx = [i for i in pack.layout if lo <= i < hi]
        n0 = _part_small0(pack.layout, lo, hi, acc_params)
        if acc_on[part]:
            # the launch added the big gradients into the first use's buffer; what is left of this call is the tail of small ones
            small = [i for i in idx if i not in acc_params]
            a = pack.offset[small[0]] if small else 0
            b = (pack.offset[small[-1]] + params[small[-1]].numel()) if small else 0
            out += list(_acc_part((key, part), pack.flat[a:b], vv, pp, side, tail_only=True))
        else:
            a = pack.offset[idx[0]]
            b = pack.offset[idx[-1]] + params[idx[-1]].numel()
            small0 = None if (n0 is None or n0 == 0) else (sum(params[i].numel() for i in idx[:n0]), frozenset(idx[:n0]))
            out += list(_acc_part((key, part), pack.flat[a:b], vv, pp, side, small0=small0))
    return out


def begin_step(param_ids=None):
    """forget gradient bookkeeping of an aborted step (called at the top of FrozenInTime.forward).  param_ids: ids of the
    calling model's parameters -- only its blocks are forgotten; None: everything."""
    _first_vblock[0] = True
    # A backward pass that raised (out of memory, anomaly mode, an interrupt) never ran the engine's final callbacks: the
    # callback latch would stay set -- no later pass would queue `_backward_done`, and with it the join of the companion
    # streams -- and the weight-gradient launches it left behind would still be unjoined.  Recover here, once per step.
    _acc_cb[0] = False
    if _deferred['sides']:
        cur = torch.cuda.current_stream() if torch.cuda.is_available() else None
        for side, main in _deferred['sides'].values():
            main.wait_stream(side)
            if cur is not None and cur.cuda_stream != main.cuda_stream:
                cur.wait_stream(side)
        _deferred['sides'].clear()
    _deferred['handed'] = []
    if param_ids is None:
        _acc.clear()
        return
    for k in [k for k in _acc if k[0][1] in param_ids]:
        del _acc[k]


# the first video block call a step creates is the last one its backward pass runs (the engine runs nodes in reverse creation order):
# nothing but the patch-embedding gradient follows it on the calling stream, so its grouped weight-gradient launch may take the chip
_first_vblock = [False]


def _tracks_grad(params):
    return torch.is_grad_enabled() and any(p.requires_grad for p in params)


def _defer_side(params):
    """(companion stream, calling stream) if this backward call may return before its weight gradients are done, else None"""
    if not _defer_allowed[0] or SW.on('EGV_NO_OVERLAP') or not SW.on('EGV_WGRAD_DEFER'):
        return None
    for p in params:
        if p.grad is not None or p._backward_hooks or getattr(p, '_post_accumulate_grad_hooks', None):
            return None
    cur = torch.cuda.current_stream()
    st = _wg_streams.get((cur.device.index, cur.cuda_stream))
    return None if st is None else (st, cur)


# CUs the persistent GEMMs of a forward video block call plan for (0: all): set while two block chains run on two streams
_fwd_cus = [0]


def set_forward_cu_limit(n: int):
    _fwd_cus[0] = int(n)


class VideoBlockFn(Function):
    """SpaceTimeBlock.forward (video_transformer.py:214-228).  params = [W, b] x 6 (timeattn.qkv, timeattn.proj, attn.qkv,
    attn.proj, mlp.fc1, mlp.fc2), [gamma, beta] x 3 (norm3, norm1, norm2) and, for a fused block, [W, b] x 3 (qkv_text_i2t,
    qkv_i2t, proj_i2t), norm_i2t_i gamma / beta, alpha_i2t."""

    # flat gradient layout of a fused block: in each part (the 18 shared parameters | the fusion extras) what the grouped weight-gradient
    # launch writes comes first and the rest -- LayerNorm pairs; in the extras the text projection qkv_text_i2t, norm_i2t_i and the gate --
    # is the part's tail (one small add when a later use of the block accumulates: _acc_reuse)
    ORDER_FUSED = list(range(18)) + [20, 21, 22, 23, 18, 19, 24, 25, 26]

    @staticmethod
    def _desc(cfg, x, y, y_mask, params):
        B, Fr, N, H, Hd, eps, L_, _track, fp8 = cfg[:9]
        fused = L_ > 0
        d = L.VBlockDesc()
        d.dtype, d.B, d.F, d.N, d.H, d.D, d.Hd, d.L, d.eps = _dt(x), B, Fr, N, H, x.shape[1], Hd, L_, eps
        d.x = _p(x)
        nw = 9 if fused else 6
        ws = [params[2 * i] for i in range(6)] + ([params[18 + 2 * i] for i in range(3)] if fused else [])
        bs = [params[2 * i + 1] for i in range(6)] + ([params[19 + 2 * i] for i in range(3)] if fused else [])
        _fill_weights(d, ws, x.dtype)
        if fp8 and x.dtype == torch.bfloat16:
            d.flags |= L.BLOCK_FP8
            for i in ([0, 1, 2, 3, 4, 5] + ([7] if fused else [])):      # the Linears over the M video tokens without a gate
                m = mx_weight(ws[i])
                if m is None:
                    raise RuntimeError('video_block(fp8=True): no MX-fp8 copy of weight %d -- call prepare_weights(..., fp8=[...]) first' % i)
                d.wq[i], d.wq_s[i], d.wtq[i], d.wtq_s[i] = _p(m[0]), _p(m[1]), _p(m[2]), _p(m[3])
        for i in range(nw):
            d.b[i] = _p(bs[i])
        for i in range(3):
            d.ln_g[i], d.ln_b[i] = _p(params[12 + 2 * i]), _p(params[13 + 2 * i])
        if fused:
            d.ln_g[3], d.ln_b[3] = _p(params[24]), _p(params[25])
            d.alpha = _p(params[26])
            d.y, d.y_mask = _p(y), _p(y_mask)
        d.stream = _st()
        d.fwd_cus = _fwd_cus[0]
        return d

    @staticmethod
    def forward(ctx, cfg, x, y, y_mask, *params):
        _need_gpu(x)
        assert x.dim() == 2 and x.is_contiguous()
        d = VideoBlockFn._desc(cfg, x, y, y_mask, params)
        out = torch.empty_like(x)
        res32, x32 = cfg[9], cfg[10]
        out32 = None
        if res32:
            # the fp32 residual stream rides beside the bf16 tensors autograd sees (x32: the fp32 value x is the rounding of, or None)
            out32 = torch.empty(x.shape, dtype=torch.float32, device=x.device)
            d.flags |= L.BLOCK_RES_F32
            d.x32, d.out32 = _p(x32), _p(out32)
        if len(cfg) > 13 and cfg[13]:
            d.flags |= L.BLOCK_INFER                     # called under no_grad: no backward call will read this call's save buffer
        nsave = lib.egv_vblock_save_bytes(C.byref(d))
        pre_save, nxt = (cfg[11], cfg[12]) if res32 else (None, None)
        if pre_save is not None and pre_save.numel() == nsave:
            # the previous block's output pass already left norm3(x) and its statistics in this call's save buffer (LayerNorm fold)
            save = pre_save
            d.flags |= L.BLOCK_H3_READY
        else:
            save = torch.empty(nsave, dtype=torch.uint8, device=x.device)
        next_save = None
        if nxt is not None:
            # fold the NEXT block's first LayerNorm into this call's output pass: it is written into the save buffer the next call will use
            g_n, b_n, L_n = nxt[:3]
            dn = L.VBlockDesc()
            dn.dtype, dn.B, dn.F, dn.N, dn.H, dn.D, dn.Hd, dn.L, dn.eps = d.dtype, d.B, d.F, d.N, d.H, d.D, d.Hd, L_n, d.eps
            if len(nxt) > 3 and nxt[3]:
                dn.flags = L.BLOCK_RES_F32 | L.BLOCK_HEAD        # the consumer is the CLS-only last block's head call: its (shorter) save layout
            so, ho = L.i64(0), L.i64(0)
            check(lib.egv_vblock_next_slots(C.byref(dn), C.byref(so), C.byref(ho)), 'egv_vblock_next_slots')
            next_save = torch.empty(lib.egv_vblock_save_bytes(C.byref(dn)), dtype=torch.uint8, device=x.device)
            d.next_g, d.next_b = _p(g_n), _p(b_n)
            d.next_h, d.next_stats = next_save.data_ptr() + ho.value, next_save.data_ptr() + so.value
        ws = workspace(lib.egv_vblock_ws_bytes(C.byref(d), 0), x.device, slot=2)
        d.out, d.save, d.save_bytes, d.ws, d.ws_bytes = _p(out), _p(save), nsave, _p(ws), ws.numel()
        check(lib.egv_vblock_fwd(C.byref(d)), 'egv_vblock_fwd')
        ctx.cfg = cfg[:9]
        ctx.key = ('v', id(params[0]))
        ctx.tail = bool(cfg[7] and _first_vblock[0] and SW.on('EGV_WGRAD_TAIL'))
        if cfg[7]:
            _first_vblock[0] = False
        _acc_forward(ctx.key, cfg[7], cfg[6] > 0)
        ctx.recompute = None
        if len(cfg) > 14 and cfg[14]:
            # activation checkpointing: this call ran with EGV_BLOCK_INFER (what only a backward call reads was not even written) and its
            # save buffer dies here; the backward call re-runs the forward from the inputs (x and, with the fp32 stream, its fp32 value)
            ctx.recompute = (res32, x32)
            save = torch.empty(0, dtype=torch.uint8, device=x.device)
        ctx.save_for_backward(x, y, y_mask, save, *params)
        if res32:
            if next_save is None:
                next_save = torch.empty(0, dtype=torch.uint8, device=x.device)
            ctx.mark_non_differentiable(out32, next_save)
            ctx.set_materialize_grads(False)             # (no 77 MB zero gradient for out32 in every backward call)
            return out, out32, next_save
        return out

    @staticmethod
    def backward(ctx, dout, *_unused):
        x, y, y_mask, save, *params = ctx.saved_tensors
        cfg = ctx.cfg[:9]
        if dout is None:                                     # (only with set_materialize_grads(False): the output was not used)
            dout = torch.zeros_like(x)
        fused = cfg[6] > 0
        dout = dout.contiguous()
        if ctx.recompute is not None:
            res32, x32 = ctx.recompute
            df = VideoBlockFn._desc(cfg, x, y, y_mask, params)
            out_r = torch.empty_like(x)
            out32_r = None
            if res32:
                out32_r = torch.empty(x.shape, dtype=torch.float32, device=x.device)
                df.flags |= L.BLOCK_RES_F32
                df.x32, df.out32 = _p(x32), _p(out32_r)
            nsave = lib.egv_vblock_save_bytes(C.byref(df))
            save = torch.empty(nsave, dtype=torch.uint8, device=x.device)
            wsf = workspace(lib.egv_vblock_ws_bytes(C.byref(df), 0), x.device, slot=2)
            df.out, df.save, df.save_bytes, df.ws, df.ws_bytes = _p(out_r), _p(save), nsave, _p(wsf), wsf.numel()
            check(lib.egv_vblock_fwd(C.byref(df)), 'egv_vblock_fwd(recompute)')
            del out_r, out32_r
        d = VideoBlockFn._desc(cfg, x, y, y_mask, params)
        dx = torch.empty_like(x)
        dy = torch.empty_like(y) if (fused and ctx.needs_input_grad[2]) else None
        d.stream2 = _side_stream_ptr()
        side = _defer_side(params) if (d.stream2 and lib.egv_vblock_bwd_defers(C.byref(d))) else None
        # beta = 1: a later use of this block in the step adds the gradients of its grouped weight-gradient launch into the first use's
        # buffer inside the launch (only in the deferred-group form: that is where the launch exists)
        gmask = int(lib.egv_vblock_bwd_groups(C.byref(d))) if side is not None else 0
        acc_params = set()
        for w in range(9 if fused else 6):
            if (gmask >> w) & 1:
                acc_params.update((2 * w, 2 * w + 1) if w < 6 else (18 + 2 * (w - 6), 19 + 2 * (w - 6)))
        order = VideoBlockFn.ORDER_FUSED if fused else None
        reuse, acc_on = _acc_reuse(ctx.key, params, 18, order or range(len(params)), acc_params, side)
        for i in reuse:
            d.acc_mask |= 1 << (i // 2 if i < 12 else 6 + (i - 18) // 2)
        # LayerNorm gradients live as [gamma ; beta] pairs (one reduction launch per LayerNorm): params are ordered that way
        gp = _GradPack(params, x.device, order, reuse)
        nwsb = lib.egv_vblock_ws_bytes(C.byref(d), 1)
        ws = torch.empty(nwsb, dtype=torch.uint8, device=x.device)
        d.save, d.save_bytes, d.ws, d.ws_bytes = _p(save), save.numel(), _p(ws), nwsb
        d.dout, d.dx, d.dy = _p(dout), _p(dx), _p(dy)
        g = gp.views
        for i in range(6):
            d.dw[i], d.db[i] = _p(g[2 * i]), _p(g[2 * i + 1])
        for i in range(3):
            d.dln_g[i], d.dln_b[i] = _p(g[12 + 2 * i]), _p(g[13 + 2 * i])
        if fused:
            for i in range(3):
                d.dw[6 + i], d.db[6 + i] = _p(g[18 + 2 * i]), _p(g[19 + 2 * i])
            d.dln_g[3], d.dln_b[3] = _p(g[24]), _p(g[25])
            d.dalpha = _p(g[26])
        if side is not None:
            # the grouped weight-gradient launch of this call keeps running on the companion stream after the call returns:
            # everything it reads or writes must outlive it in the caching allocator, and the calling stream is joined at the
            # end of the backward pass (_backward_done)
            d.flags |= L.BLOCK_NO_JOIN | (L.BLOCK_TAIL if ctx.tail else 0)
            for t in (ws, save, dout, gp.flat, y):
                if t is not None:
                    t.record_stream(side[0])
            _deferred['sides'][side[0].cuda_stream] = side
            _queue_done()
        check(lib.egv_vblock_bwd(C.byref(d)), 'egv_vblock_bwd')
        return (None, dx, dy, None, *_acc_backward(ctx.key, gp, params, 18, side, acc_params, acc_on))


class VideoHeadFn(Function):
    """The part of a SpaceTimeBlock every output row depends on -- norm3, time attention + projection + residual, norm1 and the space
    attention's qkv projection (video_transformer.py:217-219, :120) -- as one C call per direction (EGV_BLOCK_HEAD), for a block whose
    output is read at the CLS rows only.  Returns qkv_s [M, 3D] (a view of the call's save buffer).  params = [W, b] of timeattn.qkv,
    timeattn.proj, attn.qkv, then [gamma, beta] of norm3, norm1.  Gradients go to autograd tensor by tensor (two such calls per step)."""

    @staticmethod
    def _desc(cfg, x, params):
        B, Fr, N, H, Hd, eps = cfg[:6]
        d = L.VBlockDesc()
        d.dtype, d.B, d.F, d.N, d.H, d.D, d.Hd, d.L, d.eps = _dt(x), B, Fr, N, H, x.shape[1], Hd, 0, eps
        d.x = _p(x)
        _fill_weights(d, [params[0], params[2], params[4]], x.dtype)   # weight slots 0..2: timeattn.qkv, timeattn.proj, attn.qkv
        d.b[0], d.b[1], d.b[2] = _p(params[1]), _p(params[3]), _p(params[5])
        d.ln_g[0], d.ln_b[0] = _p(params[6]), _p(params[7])        # norm3
        d.ln_g[1], d.ln_b[1] = _p(params[8]), _p(params[9])        # norm1
        d.flags = L.BLOCK_RES_F32 | L.BLOCK_HEAD
        d.stream = _st()
        return d

    @staticmethod
    def forward(ctx, cfg, x, *params):
        _need_gpu(x)
        assert x.dim() == 2 and x.is_contiguous() and x.dtype == torch.bfloat16
        d = VideoHeadFn._desc(cfg, x, params)
        x32, pre_save = cfg[6], cfg[7]
        B = cfg[0]
        d.x32 = _p(x32)
        nsave = lib.egv_vblock_save_bytes(C.byref(d))
        if pre_save is not None and pre_save.numel() == nsave:
            save = pre_save
            d.flags |= L.BLOCK_H3_READY
        else:
            save = torch.empty(nsave, dtype=torch.uint8, device=x.device)
        ws = workspace(lib.egv_vblock_ws_bytes(C.byref(d), 0), x.device, slot=2)
        d.save, d.save_bytes, d.ws, d.ws_bytes = _p(save), nsave, _p(ws), ws.numel()
        check(lib.egv_vblock_fwd(C.byref(d)), 'egv_vblock_fwd(head)')
        off = lib.egv_vblock_qkv_s_offset(C.byref(d))
        M, D = x.shape
        qkv = save[off:off + M * 3 * D * 2].view(torch.bfloat16).view(M, 3 * D)
        S = M // B
        # the CLS rows of the block input in fp32 (the space residual's base, video_transformer.py:222): a second output, so that
        # their gradient comes back INTO this call's dx instead of as a zero-padded [M, D] tensor autograd has to add
        xc = (x32 if x32 is not None else x).view(B, S, D)[:, 0].float()
        ctx.cfg = cfg[:6]
        ctx.key = ('vh', id(params[0]))
        _acc_forward(ctx.key, cfg[8], False)
        ctx.save_for_backward(x, save, *params)
        return qkv, xc

    @staticmethod
    def backward(ctx, dqkv, dxc):
        x, save, *params = ctx.saved_tensors
        d = VideoHeadFn._desc(ctx.cfg, x, params)
        if dqkv is None:
            dqkv = torch.zeros(x.shape[0], 3 * x.shape[1], dtype=x.dtype, device=x.device)
        dqkv = dqkv.contiguous()
        dx = torch.empty_like(x)
        gp = _GradPack(params, x.device)
        nwsb = lib.egv_vblock_ws_bytes(C.byref(d), 1)
        ws = torch.empty(nwsb, dtype=torch.uint8, device=x.device)
        d.save, d.save_bytes, d.ws, d.ws_bytes = _p(save), save.numel(), _p(ws), nwsb
        d.dout, d.dx = _p(dqkv), _p(dx)
        g = gp.views
        for i in range(3):
            d.dw[i], d.db[i] = _p(g[2 * i]), _p(g[2 * i + 1])
        d.dln_g[0], d.dln_b[0] = _p(g[6]), _p(g[7])
        d.dln_g[1], d.dln_b[1] = _p(g[8]), _p(g[9])
        d.stream2 = _side_stream_ptr()
        side = _defer_side(params) if (d.stream2 and lib.egv_vblock_bwd_defers(C.byref(d))) else None
        if side is not None:                                 # as VideoBlockFn.backward: the grouped launch outlives the call
            d.flags |= L.BLOCK_NO_JOIN
            for t in (ws, save, dqkv, gp.flat):
                t.record_stream(side[0])
            _deferred['sides'][side[0].cuda_stream] = side
            _queue_done()
        check(lib.egv_vblock_bwd(C.byref(d)), 'egv_vblock_bwd(head)')
        if dxc is not None:                                  # the space residual reads x at the CLS rows
            B = ctx.cfg[0]
            dx.view(B, x.shape[0] // B, x.shape[1])[:, 0] += dxc.to(dx.dtype)
        return (None, dx, *_acc_backward(ctx.key, gp, params, len(params), side))


def video_block_head(x, params, B, Fr, N, H, Hd, eps):
    """params: the 10 tensors VideoHeadFn lists; x carries its fp32 value (`._res32`) and, possibly, a folded norm3 (`._pre_save`).
    Returns (qkv_s [M, 3D] bf16, the fp32 CLS rows of x [B, D])."""
    pre = None
    ps = x.__dict__.pop('_pre_save', None)
    if ps is not None and ps[1] is params[6] and ps[2]:          # (the producer's token: the norm3 weight tensor itself, and the head-form layout)
        pre = ps[0]
    return VideoHeadFn.apply((B, Fr, N, H, Hd, float(eps), stream32(x), pre, _tracks_grad(params)), x, *params)


class ClsAttnFn(Function):
    """The CLS query of a divided space attention alone: softmax(q_cls K^T / 8) V over ALL S keys of its sample (video_transformer.py:129),
    straight from the fused qkv matrix [B*S, 3D] -> [B, D].  Forward / backward are the split (few queries, many keys) launches of the generic
    attention entry point with one query row per sample (workspace sized by _split_ws; delta from the fp32 output, egv_attn_desc::O32); the backward writes the whole dqkv matrix: dK | dV of
    every row, dQ of the CLS rows, zeros in the other rows' dQ."""

    @staticmethod
    def _views(qkv, B, S, D):
        q = qkv.view(B, S * 3 * D)[:, :D]                      # row b = the CLS row of sample b (row pitch S * 3D)
        return q, qkv[:, D:2 * D], qkv[:, 2 * D:]

    @staticmethod
    def forward(ctx, qkv, B, S, H):
        _need_gpu(qkv)
        D = H * 64
        assert qkv.shape == (B * S, 3 * D) and qkv.is_contiguous()
        q, k, v = ClsAttnFn._views(qkv, B, S, D)
        O = torch.empty(B, D, dtype=qkv.dtype, device=qkv.device)
        O32 = torch.empty(B, D, dtype=torch.float32, device=qkv.device) if qkv.dtype == torch.bfloat16 else None
        lse = torch.empty(B, H, dtype=torch.float32, device=qkv.device)
        ns = 1 if S <= 224 else (S + 223) // 224
        ws, nb = _split_ws(0, B, 1, H, 1, ns, qkv.device)
        d = _mk_desc(q, k, v, O, lse, B, 1, H, _rowset(1, 0, 0, 1, 1), _rowset(S, 0, 0, 1, S), None, 0.125, nsplit=ns, ws=ws, ws_bytes=nb)
        d.O32 = _p(O32)
        check(lib.egv_attn_fwd(_dt(qkv), C.byref(d), _st()), 'egv_attn_fwd(cls only)')
        ctx.cfg = (B, S, H)
        ctx.save_for_backward(qkv, O, O32, lse)
        return O

    @staticmethod
    def backward(ctx, dO):
        qkv, O, O32, lse = ctx.saved_tensors
        B, S, H = ctx.cfg
        D = H * 64
        dO = dO.contiguous()
        q, k, v = ClsAttnFn._views(qkv, B, S, D)
        dqkv = torch.empty_like(qkv)
        dqkv[:, :D].zero_()                                    # (the other rows' queries took no part)
        dq, dk, dv = ClsAttnFn._views(dqkv, B, S, D)
        delta = torch.empty(B, H, dtype=torch.float32, device=qkv.device)
        qs, ks = _rowset(1, 0, 0, 1, 1), _rowset(S, 0, 0, 1, S)
        kw = dict(dO=dO, dQ=dq, dK=dk, dV=dv, delta=delta)
        ns = 1 if S <= 224 else (S + 223) // 224
        ws, nb = _split_ws(1, B, 1, H, 1, ns, qkv.device)
        d = _mk_desc(q, k, v, O, lse, B, 1, H, qs, ks, None, 0.125, nsplit=ns, ws=ws, ws_bytes=nb, **kw)
        d.O32 = _p(O32)
        check(lib.egv_attn_bwd_dq(_dt(qkv), C.byref(d), _st()), 'egv_attn_bwd_dq(cls only)')
        ws, nb = _dkv_ws(B, 1, H, S, 1, qkv.device)
        d = _mk_desc(q, k, v, O, lse, B, 1, H, qs, ks, None, 0.125, nsplit=1, ws=ws, ws_bytes=nb, **kw)
        check(lib.egv_attn_bwd_dkv(_dt(qkv), C.byref(d), _st()), 'egv_attn_bwd_dkv(cls only)')
        return dqkv, None, None, None


def cls_attention(qkv, B, S, H):
    return ClsAttnFn.apply(qkv, B, S, H)


class StreamRowsFn(Function):
    """first row of every sample of a video residual-stream tensor, read from its fp32 value x32; the gradient goes to x (the
    bf16 tensor autograd tracks), exactly as for x.reshape(B, rows, -1)[:, 0]"""

    @staticmethod
    def forward(ctx, x, x32, B, rows):
        ctx.meta = (x.shape, x.dtype, B, rows)
        return x32.reshape(B, rows, -1)[:, 0].contiguous()

    @staticmethod
    def backward(ctx, g):
        shape, dtype, B, rows = ctx.meta
        dx = torch.zeros(shape, dtype=dtype, device=g.device)
        dx.reshape(B, rows, -1)[:, 0] = g.to(dtype)
        return dx, None, None, None


def stream_rows(x, x32, B, rows):
    return StreamRowsFn.apply(x, x32, B, rows)


def video_res32(x, fp8=False):
    """does a video block on x run with the fp32 residual stream (EGV_VIDEO_RES32; bf16 storage without MX-fp8 operands)?"""
    return x.dtype == torch.bfloat16 and not fp8 and SW.on('EGV_VIDEO_RES32')


def stream32(x):
    """the fp32 value of a video residual-stream tensor whose bf16 rounding is x (None: x itself is all there is)"""
    return getattr(x, '_res32', None)


def video_block(x, params, B, Fr, N, H, Hd, eps, y=None, y_mask=None, L=0, fp8=False, next_ln=None, recompute=False):
    """fp8: the forward / dgrad GEMMs over the video tokens on MX-fp8 operands (bf16 mode only; BASELINE.json configs[4]).
    With EGV_VIDEO_RES32 (bf16 mode) the residual stream is fp32, as under the reference's autocast (trainer_egoclip.py:143): the
    returned bf16 tensor -- the one autograd sees, GEMMs read and the backward pass uses -- carries its fp32 value as `._res32`,
    which the next block (and the final LayerNorm) picks up.
    next_ln = (gamma, beta, L_next) of the block that will consume the result (its norm3, and whether it is a fused block over L_next
    text tokens): this call's output pass then also writes that LayerNorm into the save buffer of the next call, which travels with
    the result as `._pre_save` and is taken by the FIRST video_block call on it (EGV_LN_FOLD; a second consumer normalises itself).
    recompute: activation checkpointing (the reference's yml `use_checkpoint`, model.py:239-266,326: torch.utils.checkpoint around every
    block) -- the call keeps its inputs only; its backward call first re-runs the forward into a fresh save buffer (bitwise the first
    run: same kernels, same inputs, no dropout in a video block), then runs the backward.  0.85 GB less per block call at configs[2],
    one more forward per block."""
    res32 = video_res32(x, fp8)
    recompute = bool(recompute) and _tracks_grad(params) and not fp8
    pre = None
    if recompute:
        x.__dict__.pop('_pre_save', None)                    # (a save buffer that is dropped after the forward: nothing to fold into)
        next_ln = None
    elif res32:
        ps = x.__dict__.pop('_pre_save', None)
        if ps is not None and ps[1] is params[12] and not ps[2]:   # (identity of the norm3 weight the producer normalised with -- not id(): ids are recycled)
            pre = ps[0]
        if next_ln is not None and not SW.on('EGV_LN_FOLD'):
            next_ln = None
    cfg = (B, Fr, N, H, Hd, float(eps), L if y is not None else 0, _tracks_grad(params), bool(fp8), res32, stream32(x) if res32 else None,
           pre, next_ln if res32 else None, ((not torch.is_grad_enabled()) and SW.on('EGV_INFER_LEAN')) or recompute, recompute)
    if not res32:
        return VideoBlockFn.apply(cfg, x, y, y_mask, *params)
    out, out32, nsv = VideoBlockFn.apply(cfg, x, y, y_mask, *params)
    out._res32 = out32
    if nsv.numel():
        out._pre_save = (nsv, next_ln[0], bool(len(next_ln) > 3 and next_ln[3]))
    return out


class TextLayerFn(Function):
    """RobertaLayer.forward (roberta.py:444-505).  params = [W, b] x 6 (query, key, value, attention.output.dense,
    intermediate.dense, output.dense), [gamma, beta] x 2 (attention.output.LayerNorm, output.LayerNorm) and, for a fused layer,
    [W, b] x 4 (crossattention_t2i.self.{query,key,value}, crossattention_t2i.output.dense), alpha_t2i."""

    @staticmethod
    def _desc(cfg, hid, mask, enc, params):
        B, Lt, H, Hd, eps, S, p, seeds, _track, res32 = cfg
        fused = S > 0
        d = L.TLayerDesc()
        d.dtype, d.B, d.L, d.H, d.D, d.Hd, d.S, d.eps, d.drop_p = (L.EGV_BF16 if res32 else _dt(hid)), B, Lt, H, hid.shape[1], Hd, S, eps, p
        if res32:
            d.flags = L.BLOCK_RES_F32
        for i, sd in enumerate(seeds):
            d.seeds[i] = sd & 0xFFFFFFFF
        d.hid, d.mask = _p(hid), _p(mask)
        nw = 10 if fused else 6
        ws = [params[2 * i] for i in range(6)] + ([params[16 + 2 * i] for i in range(4)] if fused else [])
        bs = [params[2 * i + 1] for i in range(6)] + ([params[17 + 2 * i] for i in range(4)] if fused else [])
        _fill_weights(d, ws, torch.bfloat16 if res32 else hid.dtype)
        for i in range(nw):
            d.b[i] = _p(bs[i])
        for i in range(2):
            d.ln_g[i], d.ln_b[i] = _p(params[12 + 2 * i]), _p(params[13 + 2 * i])
        if fused:
            d.alpha = _p(params[24])
            d.enc = _p(enc)
        if d.dtype == L.EGV_BF16:
            m = merged_weights([params[0], params[2], params[4]], [params[1], params[3], params[5]])
            if m is not None:
                d.w_qkv, d.wt_qkv, d.b_qkv = _p(m[0]), _p(m[1]), _p(m[2])
            if fused:
                m = merged_weights([params[18], params[20]], [params[19], params[21]])
                if m is not None:
                    d.w_ckv, d.wt_ckv, d.b_ckv = _p(m[0]), _p(m[1]), _p(m[2])
        d.stream = _st()
        return d

    # flat gradient layout: [Wq Wk Wv | bq bk bv | ...] and, in the fused extras, [Wcq bcq | Wck Wcv | bck bcv | ...]: the merged
    # projections write their weight / bias gradients as one matrix / one vector
    # (in each part the gradients of the grouped launch first, the rest -- LayerNorms; key | value of text-to-image over the video rows and
    # the gate -- as the tail: _acc_reuse)
    ORDER = [0, 2, 4, 1, 3, 5] + list(range(6, 16))
    ORDER_FUSED = ORDER + [16, 17, 22, 23, 18, 20, 19, 21, 24]

    @staticmethod
    def forward(ctx, cfg, hid, mask, enc, *params):
        _need_gpu(hid)
        assert hid.dim() == 2 and hid.is_contiguous()
        d = TextLayerFn._desc(cfg, hid, mask, enc, params)
        out = torch.empty_like(hid)
        nsave = lib.egv_tlayer_save_bytes(C.byref(d))
        save = torch.empty(nsave, dtype=torch.uint8, device=hid.device)
        ws = workspace(lib.egv_tlayer_ws_bytes(C.byref(d), 0), hid.device, slot=2)
        d.out, d.save, d.save_bytes, d.ws, d.ws_bytes = _p(out), _p(save), nsave, _p(ws), ws.numel()
        check(lib.egv_tlayer_fwd(C.byref(d)), 'egv_tlayer_fwd')
        ctx.cfg = cfg
        ctx.key = ('t', id(params[0]))
        _acc_forward(ctx.key, cfg[8], cfg[5] > 0)
        ctx.save_for_backward(hid, mask, enc, save, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        hid, mask, enc, save, *params = ctx.saved_tensors
        cfg = ctx.cfg
        fused = cfg[5] > 0
        dout = dout.contiguous()
        d = TextLayerFn._desc(cfg, hid, mask, enc, params)
        dhid = torch.empty_like(hid)
        denc = torch.empty_like(enc) if (fused and ctx.needs_input_grad[3]) else None
        if cfg[9] and dout.dtype != torch.float32:
            dout = dout.float()
        gmask = int(lib.egv_tlayer_bwd_groups(C.byref(d)))
        acc_params = set()
        for w in range(10 if fused else 6):
            if (gmask >> w) & 1:
                acc_params.update((2 * w, 2 * w + 1) if w < 6 else (16 + 2 * (w - 6), 17 + 2 * (w - 6)))
        order = TextLayerFn.ORDER_FUSED if fused else TextLayerFn.ORDER
        reuse, acc_on = _acc_reuse(ctx.key, params, 16, order, acc_params, None)
        for i in reuse:
            d.acc_mask |= 1 << (i // 2 if i < 12 else 6 + (i - 16) // 2)
        gp = _GradPack(params, hid.device, order, reuse)
        nwsb = lib.egv_tlayer_ws_bytes(C.byref(d), 1)
        ws = torch.empty(nwsb, dtype=torch.uint8, device=hid.device)
        d.save, d.save_bytes, d.ws, d.ws_bytes = _p(save), save.numel(), _p(ws), nwsb
        d.dout, d.dhid, d.denc = _p(dout), _p(dhid), _p(denc)
        g = gp.views
        for i in range(6):
            d.dw[i], d.db[i] = _p(g[2 * i]), _p(g[2 * i + 1])
        for i in range(2):
            d.dln_g[i], d.dln_b[i] = _p(g[12 + 2 * i]), _p(g[13 + 2 * i])
        if fused:
            for i in range(4):
                d.dw[6 + i], d.db[6 + i] = _p(g[16 + 2 * i]), _p(g[17 + 2 * i])
            d.dalpha = _p(g[24])
        d.stream2 = _side_stream_ptr()
        check(lib.egv_tlayer_bwd(C.byref(d)), 'egv_tlayer_bwd')
        return (None, dhid, None, denc, *_acc_backward(ctx.key, gp, params, 16, None, acc_params, acc_on))


def text_layer(hid, mask, params, B, Lt, H, Hd, eps, enc=None, S=0, drop_p=0.0, seeds=(0, 0, 0, 0, 0, 0), res32=False):
    """res32: hid (and the result) are the fp32 residual stream of a bf16 model -- bf16 GEMM operands, fp32 LayerNorm in / out and
    residual sums (EGV_BLOCK_RES_F32); enc stays bf16"""
    if res32:
        assert hid.dtype == torch.float32 and (enc is None or enc.dtype == torch.bfloat16)
    return TextLayerFn.apply((B, Lt, H, Hd, float(eps), S if enc is not None else 0, float(drop_p), tuple(seeds), _tracks_grad(params), bool(res32)),
                             hid, mask, enc, *params)


# ---- patch embedding + CLS + positional / temporal embedding ------------------------------------------------
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)     # data_loader/transforms.py:17-18 defaults


class PatchEmbedFn(Function):
    """video (B,F,3,H,W) fp32 / uint8 -> patch embeddings (B*F*N, D): Conv2d(k=s=P) as im2col + MFMA GEMM (+bias)
    (video_transformer.py:78-83).  A function of the pixels and the conv weights alone: the EgoNCE tower and the shared prefix of the
    MLM / ITM passes (different CLS tokens, same patches) use ONE call per step (model._patch_tokens keeps the result for the step)."""

    @staticmethod
    def forward(ctx, video, conv_w, conv_b, dtype):
        _need_gpu(video)
        B, Fr, Cc, Hh, Ww = video.shape
        D = conv_w.shape[0]
        P = conv_w.shape[-1]
        N = (Hh // P) * (Ww // P)
        Kp = Cc * P * P
        dt = L.EGV_BF16 if dtype == torch.bfloat16 else L.EGV_F32
        patches = torch.empty(B * Fr * N, Kp, dtype=dtype, device=video.device)
        if video.dtype == torch.uint8:
            # raw clips: ToTensor + Normalize (data_loader/transforms.py:17-19) happen inside the patchify kernel
            video = video.contiguous()
            mean, std = (C.c_float * 3)(*IMAGENET_MEAN), (C.c_float * 3)(*IMAGENET_STD)
            check(lib.egv_im2col_u8(dt, _p(video), _p(patches), B * Fr, Cc, Hh, Ww, P, mean, std, _st()), 'egv_im2col_u8')
        else:
            video = video.contiguous().float()
            check(lib.egv_im2col(dt, _p(video), _p(patches), B * Fr, Cc, Hh, Ww, P, _st()), 'egv_im2col')
        w = compute_weight(conv_w, dtype).reshape(D, Kp)
        emb = torch.empty(B * Fr * N, D, dtype=dtype, device=video.device)
        gemm(patches, w, emb, M=B * Fr * N, N=D, K=Kp, lda=Kp, ldb=Kp, ldc=D, bias=conv_b)
        ctx.cfg = (B * Fr * N, D, Kp)
        ctx.wshape = conv_w.shape
        ctx.save_for_backward(patches)
        return emb

    @staticmethod
    def backward(ctx, demb):
        (patches,) = ctx.saved_tensors
        M, D, Kp = ctx.cfg
        dw, db = wgrad(demb.contiguous(), patches, M, D, Kp, bias=True)
        return None, dw.reshape(ctx.wshape), db, None


class PatchTokensFn(Function):
    """patch embeddings (B*F*N, D) -> tokens (B, 1+F*N, D): CLS concat and pos / temporal embedding (video_transformer.py:356-371)."""

    @staticmethod
    def forward(ctx, emb, cls, pos, temporal, B, Fr, N):
        D = emb.shape[1]
        dt = _dt(emb)
        out = torch.empty(B, 1 + Fr * N, D, dtype=emb.dtype, device=emb.device)
        check(lib.egv_assemble_tokens(dt, _p(emb), _p(cls), _p(pos), _p(temporal), _p(out), B, Fr, N, D, _st()),
              'egv_assemble_tokens')
        ctx.cfg = (B, Fr, N, D, dt)
        ctx.shapes = (cls.shape, pos.shape, temporal.shape)
        return out

    @staticmethod
    def backward(ctx, dX):
        B, Fr, N, D, dt = ctx.cfg
        cshape, pshape, tshape = ctx.shapes
        dX = dX.contiguous()
        dev = dX.device
        dpatch = torch.empty(B * Fr * N, D, dtype=dX.dtype, device=dev)
        dcls = torch.empty(D, dtype=torch.float32, device=dev)
        dpos = torch.empty(1 + N, D, dtype=torch.float32, device=dev)
        dtem = torch.empty(Fr, D, dtype=torch.float32, device=dev)
        ws = workspace(lib.egv_assemble_tokens_bwd_workspace_bytes(Fr, N, D), dev, slot=1)
        check(lib.egv_assemble_tokens_bwd(dt, _p(dX), _p(dpatch), _p(dcls), _p(dpos), _p(dtem), B, Fr, N, D, _p(ws), _st()),
              'egv_assemble_tokens_bwd')
        return dpatch, dcls.reshape(cshape), dpos.reshape(pshape), dtem.reshape(tshape), None, None, None


def patch_embed(video, conv_w, conv_b, dtype):
    return PatchEmbedFn.apply(video, conv_w, conv_b, dtype)


def assemble_tokens(emb, cls, pos, temporal, B, Fr, N):
    return PatchTokensFn.apply(emb, cls, pos, temporal, B, Fr, N)


def patch_tokens(video, conv_w, conv_b, cls, pos, temporal, dtype, emb=None):
    """emb: the result of patch_embed on the same video and conv weights (computed here when None)"""
    B, Fr = video.shape[0], video.shape[1]
    P = conv_w.shape[-1]
    N = (video.shape[3] // P) * (video.shape[4] // P)
    if emb is None:
        emb = patch_embed(video, conv_w, conv_b, dtype)
    return assemble_tokens(emb, cls, pos, temporal, B, Fr, N)


# ---- RoBERTa embeddings --------------------------------------------------------------------------------
class TextEmbedFn(Function):
    @staticmethod
    def forward(ctx, ids, word, pos, typ, pad_id, dtype):
        _need_gpu(ids)
        B, Lt = ids.shape
        D = word.shape[1]
        ids = ids.contiguous()
        out = torch.empty(B * Lt, D, dtype=dtype, device=ids.device)
        dt = L.EGV_BF16 if dtype == torch.bfloat16 else L.EGV_F32
        check(lib.egv_text_embed_fwd(dt, _p(ids), _p(word), _p(pos), _p(typ), _p(out), B, Lt, D, pad_id, _st()),
              'egv_text_embed_fwd')
        ctx.cfg = (B, Lt, D, pad_id, dt, word.shape, pos.shape, typ.shape)
        ctx.save_for_backward(ids)
        return out

    @staticmethod
    def backward(ctx, de):
        (ids,) = ctx.saved_tensors
        B, Lt, D, pad_id, dt, wshape, pshape, tshape = ctx.cfg
        de = de.contiguous()
        dword = torch.zeros(wshape, dtype=torch.float32, device=de.device)
        dpos = torch.zeros(pshape, dtype=torch.float32, device=de.device)
        check(lib.egv_text_embed_bwd(dt, _p(ids), _p(de), _p(dword), _p(dpos), B, Lt, D, pad_id, _st()), 'egv_text_embed_bwd')
        dtyp = colsum(de, B * Lt, D).reshape(tshape)
        return None, dword, dpos, dtyp, None, None


def text_embed(ids, word, pos, typ, pad_id, dtype):
    return TextEmbedFn.apply(ids, word, pos, typ, pad_id, dtype)


# ---- losses ------------------------------------------------------------------------------------------------
class CrossEntropySumFn(Function):
    """sum over rows of CE(logits[r, :V], labels[r]) with ignore_index; logits may be column-padded (ld > V)."""

    @staticmethod
    def forward(ctx, logits, labels, V, ignore_index):
        _need_gpu(logits)
        R, ld = logits.shape
        assert logits.is_contiguous()
        assert labels.dtype == torch.int64, "cross_entropy_sum: labels must be int64 (the kernel reads 8-byte labels)"
        labels = labels.contiguous()
        lse = torch.empty(R, dtype=torch.float32, device=logits.device)
        row = torch.empty(R, dtype=torch.float32, device=logits.device)
        check(lib.egv_ce_fwd(_dt(logits), _p(logits), _p(labels), _p(lse), _p(row), R, V, ld, ignore_index, _st()), 'egv_ce_fwd')
        ones = torch.ones(R, dtype=torch.float32, device=logits.device)
        total = dot(row, ones).reshape(())
        ctx.cfg = (R, V, ld, ignore_index)
        ctx.save_for_backward(logits, labels, lse)
        return total

    @staticmethod
    def backward(ctx, g):
        logits, labels, lse = ctx.saved_tensors
        R, V, ld, ignore_index = ctx.cfg
        coef = g.reshape(1).float().contiguous()
        dl = torch.empty_like(logits)
        check(lib.egv_ce_bwd(_dt(logits), _p(logits), _p(labels), _p(lse), _p(coef), _p(dl), R, V, ld, ld, ignore_index, _st()),
              'egv_ce_bwd')
        return dl, None, None, None


def cross_entropy_sum(logits, labels, V, ignore_index=-100):
    return CrossEntropySumFn.apply(logits, labels, V, ignore_index)


_SIM_SMALL = 4096          # entries (64 x 64: the EgoNCE matrices of 8 ranks x 8 pairs) up to which sim_matrix takes the one-wave-per-entry kernels


class SimMatrixFn(Function):
    """sim_matrix (model.py:576-584) in fp32: a/max(|a|,eps) @ (b/max(|b|,eps))^T."""

    @staticmethod
    def forward(ctx, a, b, eps):
        _need_gpu(a)
        a = a.contiguous()
        b = b.contiguous()
        n, d = a.shape
        m = b.shape[0]
        an, bn = torch.empty_like(a), torch.empty_like(b)
        na = torch.empty(n, dtype=torch.float32, device=a.device)
        nb = torch.empty(m, dtype=torch.float32, device=a.device)
        check(lib.egv_l2norm_fwd(_p(a), _p(an), _p(na), n, d, eps, _st()), 'egv_l2norm_fwd')
        check(lib.egv_l2norm_fwd(_p(b), _p(bn), _p(nb), m, d, eps, _st()), 'egv_l2norm_fwd')
        sim = torch.empty(n, m, dtype=torch.float32, device=a.device)
        if n * m <= _SIM_SMALL:          # the EgoNCE branch's matrices: one wave per entry instead of one workgroup walking K alone
            check(lib.egv_sim_small_fwd(_p(an), _p(bn), _p(sim), n, m, d, _st()), 'egv_sim_small_fwd')
        else:
            gemm(an, bn, sim, M=n, N=m, K=d, lda=d, ldb=d, ldc=m)
        ctx.eps = eps
        ctx.save_for_backward(an, bn, na, nb)
        return sim

    @staticmethod
    def backward(ctx, ds):
        an, bn, na, nb = ctx.saved_tensors
        n, d = an.shape
        m = bn.shape[0]
        ds = ds.contiguous()
        da = db = None
        if ctx.needs_input_grad[0]:
            dan = torch.empty_like(an)
            if n * m <= _SIM_SMALL:
                check(lib.egv_sim_small_bwd(_p(ds), _p(bn), _p(dan), n, m, d, 0, _st()), 'egv_sim_small_bwd')
            else:
                gemm(ds, bn, dan, a_trans=0, b_trans=1, M=n, N=d, K=m, lda=m, ldb=d, ldc=d)
            da = torch.empty_like(an)
            check(lib.egv_l2norm_bwd(_p(dan), _p(an), _p(na), _p(da), n, d, ctx.eps, _st()), 'egv_l2norm_bwd')
        if ctx.needs_input_grad[1]:
            dbn = torch.empty_like(bn)
            if n * m <= _SIM_SMALL:
                check(lib.egv_sim_small_bwd(_p(ds), _p(an), _p(dbn), m, n, d, 1, _st()), 'egv_sim_small_bwd')
            else:
                gemm(ds, an, dbn, a_trans=1, b_trans=1, M=m, N=d, K=n, lda=m, ldb=d, ldc=d)
            db = torch.empty_like(bn)
            check(lib.egv_l2norm_bwd(_p(dbn), _p(bn), _p(nb), _p(db), m, d, ctx.eps, _st()), 'egv_l2norm_bwd')
        return da, db, None


def sim_matrix_f32(a, b, eps=1e-8):
    return SimMatrixFn.apply(a, b, eps)


class EgoNCEFn(Function):
    """EgoNCE.forward (loss.py:40-61): returns (loss, mask_bool)."""

    @staticmethod
    def forward(ctx, x, sim_v, sim_n, temperature, noun, verb):
        _need_gpu(x)
        x, sim_v, sim_n = x.contiguous(), sim_v.contiguous(), sim_n.contiguous()
        n = x.shape[0]
        stats = torch.empty(2 * n, 4, dtype=torch.float32, device=x.device)
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        mask = torch.empty(n, n, dtype=torch.uint8, device=x.device)
        check(lib.egv_egonce_fwd(_p(x), _p(sim_v), _p(sim_n), n, temperature, int(noun), int(verb), _p(stats), _p(loss), _p(mask),
                                 _st()), 'egv_egonce_fwd')
        ctx.cfg = (n, temperature, int(noun), int(verb))
        ctx.save_for_backward(x, sim_v, sim_n, stats)
        mb = mask.bool()
        ctx.mark_non_differentiable(mb)
        return loss.reshape(()), mb

    @staticmethod
    def backward(ctx, g, _gm):
        x, sim_v, sim_n, stats = ctx.saved_tensors
        n, temperature, noun, verb = ctx.cfg
        g = g.reshape(1).float().contiguous()
        dx = torch.empty_like(x)
        check(lib.egv_egonce_bwd(_p(x), _p(sim_v), _p(sim_n), _p(stats), _p(g), _p(dx), n, temperature, noun, verb, _st()),
              'egv_egonce_bwd')
        return dx, None, None, None, None, None


def egonce(x, sim_v, sim_n, temperature=0.05, noun=True, verb=True):
    return EgoNCEFn.apply(x, sim_v, sim_n, temperature, noun, verb)


# ---- GEMM instrumentation (bench.py roofline) -------------------------------------------------------------
def prof_enable(on: bool):
    lib.egv_prof_enable(1 if on else 0)


def prof_reset():
    lib.egv_prof_reset()


def prof_collect(max_records: int = 1 << 16):
    fl = (C.c_double * max_records)()
    by = (C.c_double * max_records)()
    ms = (C.c_float * max_records)()
    kd = (C.c_int * max_records)()
    cu = (C.c_int * max_records)()
    n = lib.egv_prof_collect3(fl, by, ms, kd, cu, max_records)
    return [(fl[i], ms[i], kd[i], by[i], cu[i]) for i in range(n)]


def invalidate_weight_cache():
    """Drop the compute-dtype weight copies (call once per optimisation step: the fp32 masters changed)."""
    _wcache.clear()
    _wtcache.clear()
    _wmerged.clear()
    _wmx.clear()
