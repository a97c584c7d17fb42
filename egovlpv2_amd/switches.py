"""Run-time switches of the Python side in ONE table (the C side's table lives in csrc/egv_api.cpp: `egv_config_dump()`).

The defaults ARE the benchmarked, tested configuration; an environment variable of the same name overrides one switch -- an A/B
aid (tools/ab_multi.sh), never a requirement.  `on(name)` / `value(name)` are the only readers of `os.environ` in the package
(tests/test_abi_and_host.py checks that, and pins the defaults), and they are read at the point of use, so a test or a tool may
flip a switch between steps of one process."""
from __future__ import annotations

import os

# name: (default, meaning)
SWITCHES = {
    'EGV_LIB_PATH': ('', 'alternative build of libegovlp_hip.so (kernel experiments of tools/)'),
    'EGV_NO_OVERLAP': ('0', 'everything on the calling stream: no text / weight-gradient companion streams'),
    'EGV_SIDE_PRIORITY': ('1', 'HIP priority of the companion streams (1 = low: they take the CUs the calling stream leaves free)'),
    'EGV_TEXT_PRIORITY': ('', 'priority of the text stream when it should differ from EGV_SIDE_PRIORITY'),
    'EGV_TEXT_STREAM': ('1', 'text tower on its companion stream'),
    'EGV_TAIL_REST_AUX': ('0', 'the B-row chain of the EgoNCE tower\'s CLS-only last block runs on the text stream\'s weight-gradient companion'),
    'EGV_TAIL_STREAM': ('1', 'MLM head + cross entropy and the EgoNCE tail on the text stream'),
    'EGV_WGRAD_DEFER': ('1', 'a video block backward returns while its grouped weight-gradient launch is still running'),
    'EGV_WGRAD_ACC': ('1', 'later uses of a block add their grouped weight gradients into the first use\'s buffer inside the launch (beta = 1)'),
    'EGV_WGRAD_TAIL': ('1', 'the last block of a backward pass gives its weight gradients 7/8 of the chip'),
    'EGV_ATTN_FUSED_BWD': ('1', 'per-op attention: one-launch backward where a kernel covers the shape'),
    'EGV_ATTN_FUSED_CLS': ('1', 'per-op attention: the group launches also serve the CLS row'),
    'EGV_TEXT_FP32': ('0', 'text-only tower pass in fp32 storage inside the bf16 model'),
    'EGV_TEXT_RES32': ('1', 'fp32 residual stream of the text tower in the bf16 mode'),
    'EGV_VIDEO_RES32': ('1', 'fp32 residual stream of the video tower in the bf16 mode (sums and LayerNorm inputs in fp32, as under autocast)'),
    'EGV_CLS_TAIL': ('1', 'the last block of a video pass computes its CLS rows only (attn.proj, image-to-text part and MLP on B rows instead of B*S)'),
    'EGV_INFER_LEAN': ('1', 'video blocks called under torch.no_grad() do not write what only a backward pass would read (the MLP\'s pre-activation): EGV_BLOCK_INFER'),
    'EGV_LN_FOLD': ('1', 'fp32 video stream: a block\'s output pass also writes the next block\'s first LayerNorm (one HBM pass less per block)'),
    'EGV_ITM_RES32': ('1', 'the ITM pass gathers the fp32 value of the shared video prefix with the clips (0: it restarts from the bf16 rows)'),
    'EGV_VIDEO_FP8': ('0', 'MX-fp8 forward / data-gradient GEMMs in the video blocks (configs[4])'),
    'EGV_ROBERTA_CHECKPOINT': ('', 'path of a pretrained RoBERTa state dict (text_params["pretrained_path"] wins)'),
    'EGV_VIT_CHECKPOINT': ('', 'path of a pretrained ViT state dict (video_params["pretrained_path"] wins)'),
    'EGV_MERGE_PROJ': ('1', 'query|key|value and text-to-image key|value of a RoBERTa layer as one GEMM each'),
    'EGV_NO_PREFIX_SHARING': ('0', 'recompute the unfused video prefix in the ITM pass as the reference does'),
    'EGV_TEXT_BATCH': ('1', 'EgoNCE text tower prefix and MLM text prefix as one pass over the concatenated batch'),
    'EGV_EGONCE_TAIL_LATE': ('1', 'differentiable EgoNCE tail created after both text prefixes'),
    'EGV_ITM_DRAW_EARLY': ('1', 'ITM negatives drawn before the MLM pass is enqueued'),
    'EGV_ITM_FIRST': ('0', 'create the ITM pass before the MLM pass (default: the reference order)'),
    'EGV_MLM_TOP_AUX': ('1', 'the late MLM top runs on the text stream\'s weight-gradient companion instead of the text stream'),
    'EGV_MLM_TOP_LATE': ('1', 'the MLM pass\'s last fused text layer + head + cross entropy are created after the ITM pass (their backward then runs first, beside the ITM tail)'),
    'EGV_EXCHANGE_HOST_TABLE': ('0', 'exchange the ITM request table through a host (gloo) all-gather also on RCCL'),
    'EGV_SYNC_WIRE': ('fp32', 'flat gradient sync: fp32 = in-place all-reduce, bf16 = bf16 on the links with fp32 accumulation on arrival (grad_sync.allreduce_bf16_wire)'),
    'EGV_SYNC_FORCE': ('0', 'run the flat gradient all-reduces at world size 1 (test aid)'),
    'EGV_ACT_CHECKPOINT': ('0', 'activation checkpointing of the video blocks: 1 = on, yml = as the yml use_checkpoint says (reference behaviour), 0 = keep every activation'),
    'EGV_ALLOW_UNSAFE_CHECKPOINT': ('0', 'allow full unpickling of a checkpoint whose safe load fails'),
}


def value(name: str) -> str:
    """the switch as a string: the environment's value if set (and non-empty), else the table's default"""
    default = SWITCHES[name][0]                     # KeyError: a switch that is not in the table is a programming error
    v = os.environ.get(name)
    return v if v not in (None, '') else default


def on(name: str) -> bool:
    return value(name) not in ('', '0')


def defaults() -> dict:
    return {k: v[0] for k, v in SWITCHES.items()}
