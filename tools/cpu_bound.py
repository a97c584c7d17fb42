"""How far ahead of the GPU does the host run?  Times step() enqueue (no sync) vs the synchronized step time."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from egovlpv2_amd import hipops as ops
from egovlpv2_amd.config import PathConfig
from egovlpv2_amd.synthetic import make_state_dict, make_batch
from egovlpv2_amd.model.model import FrozenInTime
from egovlpv2_amd.model.loss import EgoNCE
from egovlpv2_amd.trainer.trainer_egoclip import AllGather_multi
dev = torch.device('cuda:0')
cfg = PathConfig(frames=16, drop_rate=0.1)
model = FrozenInTime({'model': 'SpaceTimeTransformer', 'num_frames': 16, 'pretrained': True}, {'model': 'roberta-base', 'pretrained': True, 'input': 'text'},
                     path_config=cfg, task_names='EgoNCE_MLM_ITM', compute_dtype=torch.bfloat16)
model.load_state_dict(make_state_dict(cfg, 0), strict=True)
model = model.to(dev)
data, noun, verb = make_batch(cfg, 8, 32, 1234)
data = {'video': data['video'].to(dev), 'text': {k: v.to(dev) for k, v in data['text'].items()},
        'text_mlm_ids': data['text_mlm_ids'].to(dev), 'text_mlm_labels': data['text_mlm_labels'].to(dev)}
noun, verb = noun.to(dev), verb.to(dev)
args = types.SimpleNamespace(world_size=1, rank=0)
np.random.seed(1); torch.manual_seed(1)
def step(split=False):
    ops.invalidate_weight_cache()
    for p in model.parameters(): p.grad = None
    t0 = time.perf_counter()
    loss, ld, _ = model(data, noun, verb, AllGather_multi.apply, 1, args, {'loss': {'type': 'EgoNCE'}}, EgoNCE(), 0, task_names='EgoNCE_MLM_ITM')
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1
for _ in range(3): step()
torch.cuda.synchronize()
for _ in range(4):
    t0 = time.perf_counter()
    f, b = step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue fwd {f*1e3:6.1f} ms (incl. ITM host sync)  bwd {b*1e3:6.1f} ms  total host {1e3*(t1-t0):6.1f} ms   step with sync {1e3*(t2-t0):6.1f} ms", flush=True)
