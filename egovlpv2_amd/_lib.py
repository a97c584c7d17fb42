"""ctypes binding of libegovlp_hip.so (C ABI: include/egovlp_hip.h).

There is no fallback: if the shared library is missing or a symbol is absent, importing the product
path fails loudly.  ``build()`` in ``__graft_entry__.py`` (or ``egovlpv2_amd/csrc/build.sh``) produces
the library in-tree.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- must be imported first: the HIP runtime torch bundles has to be the one in the process

_HERE = os.path.dirname(os.path.abspath(__file__))
# EGV_LIB_PATH: an alternative build of the SAME library (kernel experiments of tools/); the product path never sets it
from . import switches as _sw  # noqa: E402

LIB_PATH = _sw.value('EGV_LIB_PATH') or os.path.join(_HERE, 'libegovlp_hip.so')

ABI_VERSION = 6
EGV_F32, EGV_BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_RELU, ACT_TANH, ACT_GELU_D = 0, 1, 2, 3, 4

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_longlong, C.c_float


class AttnDesc(C.Structure):
    """struct egv_attn_desc (include/egovlp_hip.h) -- field order is the ABI."""
    _fields_ = [
        ('Q', vp), ('K', vp), ('V', vp), ('O', vp), ('dO', vp), ('dQ', vp), ('dK', vp), ('dV', vp),
        ('ldq', i32), ('ldk', i32), ('ldv', i32), ('ldo', i32), ('lddq', i32), ('lddk', i32), ('lddv', i32),
        ('qoff', i32), ('koff', i32), ('voff', i32), ('ooff', i32), ('dqoff', i32), ('dkoff', i32), ('dvoff', i32),
        ('lse', vp), ('delta', vp),
        ('B', i32), ('G', i32), ('H', i32),
        ('q_bs', i64), ('q_base', i64), ('q_gs', i64), ('q_is', i64), ('q_n', i32),
        ('k_bs', i64), ('k_base', i64), ('k_gs', i64), ('k_is', i64), ('k_n', i32),
        ('extra', i32), ('extra_bs', i64), ('extra_row', i64),
        ('scale', f32),
        ('mask', vp), ('mask_ld', i32),
        ('nsplit', i32), ('ws', vp), ('ws_bytes', i64),
        ('drop_p', f32), ('drop_seed', C.c_uint),
        ('O32', vp),
    ]


class VBlockDesc(C.Structure):
    """struct egv_vblock_desc (include/egovlp_hip.h) -- field order is the ABI."""
    _fields_ = [
        ('dtype', i32), ('B', i32), ('F', i32), ('N', i32), ('H', i32), ('D', i32), ('Hd', i32), ('L', i32),
        ('eps', f32),
        ('x', vp), ('out', vp), ('y', vp), ('y_mask', vp),
        ('save', vp), ('save_bytes', i64), ('ws', vp), ('ws_bytes', i64),
        ('w', vp * 9), ('wt', vp * 9), ('b', vp * 9),
        ('ln_g', vp * 4), ('ln_b', vp * 4),
        ('alpha', vp),
        ('dout', vp), ('dx', vp), ('dy', vp),
        ('dw', vp * 9), ('db', vp * 9), ('dln_g', vp * 4), ('dln_b', vp * 4), ('dalpha', vp),
        ('stream', vp), ('stream2', vp),
        ('flags', i32),
        ('wq', vp * 9), ('wq_s', vp * 9), ('wtq', vp * 9), ('wtq_s', vp * 9),
        ('x32', vp), ('out32', vp),
        ('next_g', vp), ('next_b', vp), ('next_h', vp), ('next_stats', vp),
        ('fwd_cus', i32),
        ('acc_mask', C.c_uint),
    ]


class TLayerDesc(C.Structure):
    """struct egv_tlayer_desc (include/egovlp_hip.h) -- field order is the ABI."""
    _fields_ = [
        ('dtype', i32), ('B', i32), ('L', i32), ('H', i32), ('D', i32), ('Hd', i32), ('S', i32),
        ('eps', f32), ('drop_p', f32),
        ('seeds', C.c_uint * 6),
        ('hid', vp), ('out', vp), ('mask', vp), ('enc', vp),
        ('save', vp), ('save_bytes', i64), ('ws', vp), ('ws_bytes', i64),
        ('w', vp * 10), ('wt', vp * 10), ('b', vp * 10),
        ('ln_g', vp * 2), ('ln_b', vp * 2),
        ('alpha', vp),
        ('dout', vp), ('dhid', vp), ('denc', vp),
        ('dw', vp * 10), ('db', vp * 10), ('dln_g', vp * 2), ('dln_b', vp * 2), ('dalpha', vp),
        ('stream', vp), ('stream2', vp),
        ('flags', i32),
        ('w_qkv', vp), ('wt_qkv', vp), ('b_qkv', vp), ('w_ckv', vp), ('wt_ckv', vp), ('b_ckv', vp),
        ('acc_mask', C.c_uint),
    ]


class WgradProblem(C.Structure):
    """struct egv_wgrad_problem (include/egovlp_hip.h)"""
    _fields_ = [('dy', vp), ('ldy', i32), ('x', vp), ('ldx', i32), ('dw', vp), ('db', vp), ('gate', vp), ('N', i32), ('K', i32), ('accumulate', i32)]


BLOCK_NO_JOIN = 1
BLOCK_RES_F32 = 2
BLOCK_FP8 = 4
BLOCK_TAIL = 8
BLOCK_H3_READY = 16
BLOCK_HEAD = 32
BLOCK_INFER = 64


# name -> (restype, argtypes); every symbol declared in include/egovlp_hip.h
PROTOTYPES = {
    'egv_abi_version': (i32, []),
    'egv_last_error': (C.c_char_p, []),
    'egv_config_dump': (C.c_char_p, []),
    'egv_gemm': (i32, [i32, i32, i32, i32, i32, i32, vp, i32, vp, i32, vp, i32, i32, vp, i32, vp, vp, vp, vp, vp, i32, i32, f32, vp]),
    'egv_gemm_wgrad_workspace_bytes': (i64, [i32, i32, i32]),
    'egv_gemm_wgrad': (i32, [i32, i32, i32, i32, vp, i32, vp, i32, vp, vp, f32, vp, vp, i64, vp]),
    'egv_gemm_wgrad_grouped_workspace_bytes': (i64, [i32, i32, C.POINTER(WgradProblem), i32]),
    'egv_gemm_wgrad_grouped': (i32, [i32, i32, i32, C.POINTER(WgradProblem), i32, vp, i64, vp]),
    'egv_gemm_wgrad_group_reset': (i32, []),
    'egv_layernorm_fwd': (i32, [i32, vp, vp, vp, vp, vp, i32, i32, f32, vp]),
    'egv_layernorm_bwd_workspace_bytes': (i64, [i32, i32]),
    'egv_layernorm_bwd': (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]),
    'egv_layernorm_bwd2': (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]),
    'egv_colsum_workspace_bytes': (i64, [i32, i32]),
    'egv_colsum': (i32, [i32, vp, i32, i32, i32, vp, f32, vp, vp, vp]),
    'egv_dot': (i32, [i32, vp, vp, i64, vp, f32, vp, vp]),
    'egv_act_bwd': (i32, [i32, vp, vp, vp, i64, i32, vp]),
    'egv_dropout_add': (i32, [i32, vp, vp, vp, vp, i64, f32, C.c_uint, vp]),
    'egv_dropout_add_mixed': (i32, [i32, vp, vp, vp, i32, vp, i64, f32, C.c_uint, vp]),
    'egv_layernorm_fwd_res32': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, f32, vp]),
    'egv_sum_layernorm': (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp]),
    'egv_layernorm_bwd_res32': (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]),
    'egv_cast': (i32, [i32, i32, vp, vp, i64, vp]),
    'egv_cast_transpose': (i32, [vp, vp, i32, i32, vp]),
    'egv_transpose': (i32, [i32, i32, vp, vp, i32, i32, i32, vp]),
    'egv_cast_weights': (i32, [vp, vp, i32, i32, vp]),
    'egv_cast_weights_ld': (i32, [vp, vp, i32, i32, vp]),
    'egv_copy_segments': (i32, [vp, i32, vp]),
    'egv_layernorm_fwd_mx': (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp]),
    'egv_mx_scale_bytes': (i64, [i32, i32, i32]),
    'egv_quant_mx': (i32, [vp, i32, i32, i32, vp, vp, i32, vp]),
    'egv_quant_mx_batch': (i32, [vp, vp, i32, i32, vp]),
    'egv_gemm_mx': (i32, [i32, i32, i32, vp, vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, i32, vp, vp, vp]),
    'egv_stream_create': (i32, [i32, C.POINTER(vp)]),
    'egv_attn_split_workspace_bytes': (i64, [i32, i32, i32, i32, i32, i32]),
    'egv_attn_fwd': (i32, [i32, C.POINTER(AttnDesc), vp]),
    'egv_attn_fwd_extra_workspace_bytes': (i64, [i32, i32, i32]),
    'egv_attn_fwd_covers_extra': (i32, [i32, C.POINTER(AttnDesc)]),
    'egv_attn_bwd_dq': (i32, [i32, C.POINTER(AttnDesc), vp]),
    'egv_attn_bwd_dkv_workspace_bytes': (i64, [i32, i32, i32, i32, i32]),
    'egv_attn_bwd_dkv': (i32, [i32, C.POINTER(AttnDesc), vp]),
    'egv_attn_bwd_fused': (i32, [i32, C.POINTER(AttnDesc), vp]),
    'egv_attn_bwd_fused_workspace_bytes': (i64, [i32, i32, i32]),
    'egv_attn_fewkeys_workspace_bytes': (i64, [i32, i32, i32, i32]),
    'egv_attn_fewq_workspace_bytes': (i64, [i32, i32, i32, i32]),
    'egv_attn_bwd_pair_covers_extra': (i32, [i32, C.POINTER(AttnDesc)]),
    'egv_attn_bwd_extra_reduce': (i32, [i32, C.POINTER(AttnDesc), i32, vp]),
    'egv_im2col': (i32, [i32, vp, vp, i32, i32, i32, i32, i32, vp]),
    'egv_im2col_u8': (i32, [i32, vp, vp, i32, i32, i32, i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_float), vp]),
    'egv_assemble_tokens': (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    'egv_assemble_tokens_bwd_workspace_bytes': (i64, [i32, i32, i32]),
    'egv_assemble_tokens_bwd': (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    'egv_text_embed_fwd': (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    'egv_text_embed_bwd': (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    'egv_ce_fwd': (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i64, vp]),
    'egv_ce_bwd': (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, vp]),
    'egv_l2norm_fwd': (i32, [vp, vp, vp, i32, i32, f32, vp]),
    'egv_l2norm_bwd': (i32, [vp, vp, vp, vp, i32, i32, f32, vp]),
    'egv_sim_small_fwd': (i32, [vp, vp, vp, i32, i32, i32, vp]),
    'egv_sim_small_bwd': (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    'egv_egonce_fwd': (i32, [vp, vp, vp, i32, f32, i32, i32, vp, vp, vp, vp]),
    'egv_egonce_bwd': (i32, [vp, vp, vp, vp, vp, vp, i32, f32, i32, i32, vp]),
    'egv_adamw_step': (i32, [vp, vp, i32, i32, f32, f32, f32, f32, f32, f32, f32, vp]),
    'egv_vblock_qkv_s_offset': (i64, [C.POINTER(VBlockDesc)]),
    'egv_vblock_next_slots': (i32, [C.POINTER(VBlockDesc), C.POINTER(i64), C.POINTER(i64)]),
    'egv_vblock_save_bytes': (i64, [C.POINTER(VBlockDesc)]),
    'egv_vblock_ws_bytes': (i64, [C.POINTER(VBlockDesc), i32]),
    'egv_vblock_fwd': (i32, [C.POINTER(VBlockDesc)]),
    'egv_vblock_bwd': (i32, [C.POINTER(VBlockDesc)]),
    'egv_vblock_bwd_defers': (i32, [C.POINTER(VBlockDesc)]),
    'egv_vblock_bwd_groups': (C.c_uint, [C.POINTER(VBlockDesc)]),
    'egv_tlayer_save_bytes': (i64, [C.POINTER(TLayerDesc)]),
    'egv_tlayer_ws_bytes': (i64, [C.POINTER(TLayerDesc), i32]),
    'egv_tlayer_fwd': (i32, [C.POINTER(TLayerDesc)]),
    'egv_tlayer_bwd': (i32, [C.POINTER(TLayerDesc)]),
    'egv_tlayer_bwd_groups': (C.c_uint, [C.POINTER(TLayerDesc)]),
    'egv_prof_enable': (i32, [i32]),
    'egv_prof_reset': (i32, []),
    'egv_prof_collect': (i32, [C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int), i32]),
    'egv_prof_collect2': (i32, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int), i32]),
    'egv_prof_collect3': (i32, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_int), i32]),
}


class EgvError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or egovlpv2_amd/csrc/build.sh). There is no CPU fallback for this path (no CPU fallback).")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError -> loud failure on a stale library
        fn.restype = res
        fn.argtypes = args
    if lib.egv_abi_version() != ABI_VERSION:
        raise ImportError("libegovlp_hip.so ABI version mismatch")
    return lib


lib = _load()


def check(rc: int, what: str = ''):
    if rc != 0:
        raise EgvError(f"{what}: rc={rc}: {lib.egv_last_error().decode()}")
