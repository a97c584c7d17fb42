"""TEST INFRASTRUCTURE ONLY (oracle): CPU restatement of the MXFP8 E4M3 operand format of BASELINE.json configs[4] ("fp8 MFMA
weight path").  The reference (facebookresearch/EgoVLPv2) has no fp8 code -- parity for this path is pinned on the published
format instead: OCP Microscaling Formats (MX) v1.0, MXFP8 with E4M3 elements -- blocks of 32 elements along the contraction
dimension, one E8M0 scale 2^(s - 127) per block, elements OCP e4m3fn (torch.float8_e4m3fn, round to nearest even).  The scale
rule is the product's (egovlpv2_amd/csrc/egv_mx.hip): the smallest power of two that brings the block's largest magnitude to
<= 448.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import torch


def quantize(x: torch.Tensor):
    """x [R, K] (any float dtype; values are taken as given) -> (codes uint8 [R, K], scale exponents uint8 [R, K/32])"""
    R, K = x.shape
    assert K % 32 == 0
    v = x.float().reshape(R, K // 32, 32)
    amax = v.abs().amax(-1)
    bits = amax.contiguous().view(torch.int32)
    e8 = (bits >> 23) - 8 + ((bits & 0x7fffff) > 0x600000).to(torch.int32)
    e8 = e8.clamp(0, 254)
    scaled = torch.ldexp(v, (127 - e8).unsqueeze(-1)).clamp(-448.0, 448.0)
    codes = scaled.to(torch.float8_e4m3fn).view(torch.uint8).reshape(R, K)
    return codes, e8.to(torch.uint8)


def dequantize(codes: torch.Tensor, e8: torch.Tensor) -> torch.Tensor:
    R, K = codes.shape
    v = codes.view(torch.float8_e4m3fn).float().reshape(R, K // 32, 32)
    return torch.ldexp(v, (e8.to(torch.int32) - 127).unsqueeze(-1)).reshape(R, K)


def fake_quant(x: torch.Tensor) -> torch.Tensor:
    """quantise + dequantise along the last dimension (fp32 result)"""
    shp = x.shape
    c, e = quantize(x.reshape(-1, shp[-1]))
    return dequantize(c, e).reshape(shp)


def scale_layout(e8: torch.Tensor, role: int) -> torch.Tensor:
    """[R, K/32] scale exponents -> the byte array egv_quant_mx writes (rows past R hold 0x7f)"""
    R, KB = e8.shape
    assert KB % 4 == 0
    nblk = ((R + 191) // 192) * 4 if role == 0 else (R + 63) // 64
    out = torch.full((KB // 4, nblk, 4, 16, 4), 0x7f, dtype=torch.uint8)
    r = torch.arange(R)
    if role == 0:           # 48-row blocks: one sub-tile of a wave row of the GEMM's 192-row tile; byte = 16-row fragment 0..2
        blk, rb = r // 48, r % 48
        fr, byte = rb % 16, rb // 16
    else:                   # 64-row blocks in the B-row permutation of the GEMM tile
        blk, rb = r // 64, r % 64
        x = rb % 32
        fr, byte = ((x // 8) * 4) | (x % 4), (rb // 32) * 2 + ((x // 4) % 2)
    for kb in range(KB):
        out[kb // 4, blk, kb % 4, fr, byte] = e8[:, kb]
    return out.reshape(-1)


def gemm_ref(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """fp64 product of the dequantised operands: what an exact block-scaled fp8 GEMM returns for a [M,K], b [N,K]"""
    return fake_quant(a).double() @ fake_quant(b).double().t()


# ---- block-level oracle: the reference's Linear (video_transformer.py:53,56,120,152,166,183) under the MX-fp8 weight path --------
class MxLinearFn(torch.autograd.Function):
    """y = Q(x) Q(W)^T + b with both operands MX-quantised along the contraction dimension (dequantised-weight fp32 reference);
    dx = Q(dy) Q'(W) with the output gradient and the weight quantised along N (the contraction of the data gradient);
    dW = dy^T x and db = sum dy from the un-quantised (bf16-rounded) operands, as the product computes them."""

    @staticmethod
    def forward(ctx, x, w, b):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).to(torch.bfloat16).float()
        wb = w.to(torch.bfloat16).float()
        ctx.save_for_backward(x2, wb)
        ctx.shp, ctx.has_b = shp, b is not None
        y = fake_quant(x2) @ fake_quant(wb).t()
        if b is not None:
            y = y + b
        return y.reshape(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, wb = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1]).to(torch.bfloat16).float()
        dx = fake_quant(dy2) @ fake_quant(wb.t().contiguous()).t()
        dw = dy2.t() @ x2
        db = dy2.sum(0) if ctx.has_b else None
        return dx.reshape(ctx.shp), dw, db


import contextlib
import re

_VIDEO_LINEAR = re.compile(r'video_model\.blocks\.\d+\.(timeattn\.(qkv|proj)|attn\.(qkv|proj|qkv_i2t)|mlp\.fc[12])$')


@contextlib.contextmanager
def mx_video_linears(ref_model):
    """inside the context the oracle (oracle/ref_model.py) runs the forward / data gradient of every video-block Linear over the video
    tokens -- the ones the product runs on MX-fp8 operands -- through MxLinearFn"""
    orig = ref_model._lin

    def _lin(x, sd, prefix, bias=True):
        if _VIDEO_LINEAR.match(prefix) and x.shape[-1] % 128 == 0:
            return MxLinearFn.apply(x, sd[prefix + '.weight'], sd[prefix + '.bias'] if bias else None)
        return orig(x, sd, prefix, bias)
    ref_model._lin = _lin
    try:
        yield
    finally:
        ref_model._lin = orig
