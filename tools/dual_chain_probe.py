"""Feasibility probe (round 5): two independent forward chains of video blocks on two HIP streams -- do the HBM-bound kernels of one
chain (LayerNorm / sum passes, attention) hide behind the GEMMs of the other?  Each chain = the 12 unfused blocks of the EgoNCE video
tower at configs[2] (B = 8, 16 x 224^2), no grad.  Sequential (both chains on one stream) against concurrent, for several CU shares
of the persistent GEMM grids (egv_vblock_desc::fwd_cus).  usage: python tools/dual_chain_probe.py"""
import os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', os.environ.get('EGV_HWQ', '8'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egovlpv2_amd import hipops as ops
from egovlpv2_amd.config import PathConfig
from egovlpv2_amd.synthetic import make_state_dict, make_batch
from egovlpv2_amd.model.model import FrozenInTime

dev = torch.device('cuda:0')
cfg = PathConfig(frames=16, drop_rate=0.0)
model = FrozenInTime({'model': 'SpaceTimeTransformer', 'num_frames': 16, 'pretrained': True}, {'model': 'roberta-base', 'pretrained': True, 'input': 'text'},
                     path_config=cfg, task_names='EgoNCE_MLM_ITM', compute_dtype=torch.bfloat16)
model.load_state_dict(make_state_dict(cfg, 0), strict=True)
model = model.to(dev).eval()
data, _, _ = make_batch(cfg, 8, 32, 1234)
video = data['video'].to(dev)
B = 8
with torch.no_grad():
    model._prepare_weights()
    x0 = model._patch_tokens(video, 'video_model.cls_token')
    x1 = x0.clone()
torch.cuda.synchronize()
sB = torch.cuda.Stream(device=dev)


def chain(x, n=12):
    for i in range(n):
        x = model._video_block(x, i, B, next_block=(i + 1, 0))
    return x


def run(mode, cus):
    ops.set_forward_cu_limit(cus)
    main = torch.cuda.current_stream()
    with torch.no_grad():
        for rep in range(3):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if mode == 'seq':
                chain(x0); chain(x1)
            elif mode == 'par':
                sB.wait_stream(main)
                # issue block by block, alternating, so that neither stream's queue runs dry
                xa, xb = x0, x1
                for i in range(12):
                    xa = model._video_block(xa, i, B, next_block=(i + 1, 0))
                    with torch.cuda.stream(sB):
                        xb = model._video_block(xb, i, B, next_block=(i + 1, 0))
                main.wait_stream(sB)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
    ops.set_forward_cu_limit(0)
    return ms


print('hw queues', os.environ.get('GPU_MAX_HW_QUEUES'))
print('sequential, whole chip      : %.2f ms for 2 x 12 blocks' % run('seq', 0))
for cus in (0, 224, 192, 176, 160, 144, 128):
    print('concurrent, GEMM share %3d  : %.2f ms' % (cus, run('par', cus)))
print('sequential, whole chip      : %.2f ms' % run('seq', 0))
