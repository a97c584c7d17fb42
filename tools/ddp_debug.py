"""debug: gradients of the tiny config with and without a 1-rank DistributedDataParallel wrap"""
import os, sys, types
import numpy as np, torch, torch.distributed as dist
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
from helpers import load_golden
from egovlpv2_amd.synthetic import make_state_dict, make_batch
from egovlpv2_amd.model.model import FrozenInTime
from egovlpv2_amd.model.loss import EgoNCE
from egovlpv2_amd.trainer.trainer_egoclip import AllGather_multi
from egovlpv2_amd import hipops as ops
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = '29671'
dist.init_process_group('gloo', rank=0, world_size=1)
_, cfg, B, L, wseed, _ = load_golden('tiny')
B = 4
sd = make_state_dict(cfg, wseed)
args = types.SimpleNamespace(world_size=1, rank=0)
def run(ddp, steps=2):
    m = FrozenInTime({'model': 'SpaceTimeTransformer', 'num_frames': cfg.frames, 'pretrained': True}, {'model': 'roberta-base', 'pretrained': True, 'input': 'text'},
                     path_config=cfg, task_names='EgoNCE_MLM_ITM', compute_dtype=torch.float32)
    m.load_state_dict(sd, strict=True); m = m.cuda()
    net = m
    if ddp:
        from torch.nn.parallel import DistributedDataParallel as DDP
        net = DDP(m, device_ids=[0], static_graph=True, gradient_as_bucket_view=True, find_unused_parameters=False)
    out = []
    for step in range(steps):
        data, noun, verb = make_batch(cfg, B, L, 500 + 10 * step)
        dev = {'video': data['video'].cuda(), 'text': {k: v.cuda() for k, v in data['text'].items()}, 'text_mlm_ids': data['text_mlm_ids'].cuda(), 'text_mlm_labels': data['text_mlm_labels'].cuda()}
        np.random.seed(40 + step); torch.manual_seed(40 + step)
        net.zero_grad(set_to_none=True)
        loss, ld, ret = net(dev, noun.cuda(), verb.cuda(), AllGather_multi.apply, 1, args, {'loss': {'type': 'EgoNCE'}}, EgoNCE(), 0, task_names='EgoNCE_MLM_ITM')
        loss.backward(); torch.cuda.synchronize()
        out.append({n: (None if p.grad is None else p.grad.detach().double().cpu().clone()) for n, p in m.named_parameters()})
        print('ddp' if ddp else 'plain', step, 'acc entries left', len(ops._acc), 'loss', float(loss))
    return out
a = run(False); b = run(True)
for step in range(2):
    bad = []
    for n in a[step]:
        x, y = a[step][n], b[step][n]
        if x is None or y is None:
            if not (x is None and y is None): bad.append((n, 'None mismatch', x is None, y is None))
            continue
        e = ((x - y).norm() / (x.norm() + 1e-9)).item()
        if e > 1e-4: bad.append((n, round(e, 4)))
    print('step', step, 'differences plain vs ddp:', len(bad), bad[:12])
