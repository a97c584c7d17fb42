"""Is the step host-bound?  Wall time of the step against the time the host needs to ISSUE its forward and its backward
(perf_counter around the calls, no synchronisation in between; the forward contains the one host wait of the step, the ITM draw)."""
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
rec = []
def step():
    t0 = time.perf_counter()
    ops.invalidate_weight_cache()
    for p in model.parameters(): p.grad = None
    loss, ld, _ = model(data, noun, verb, AllGather_multi.apply, 1, args, {'loss': {'type': 'EgoNCE'}}, EgoNCE(), 0, task_names='EgoNCE_MLM_ITM')
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    return t0, t1, t2
for _ in range(4): step()
torch.cuda.synchronize()
N = 10
ts = time.perf_counter()
for _ in range(N):
    t0, t1, t2 = step()
    rec.append((t1 - t0, t2 - t1))
te0 = time.perf_counter()
torch.cuda.synchronize()
te = time.perf_counter()
print(f"wall {1e3 * (te - ts) / N:.2f} ms/step; host issue: forward {1e3 * np.mean([r[0] for r in rec]):.2f} ms, backward {1e3 * np.mean([r[1] for r in rec]):.2f} ms; "
      f"final drain {1e3 * (te - te0):.2f} ms (GPU work still queued when the host finished issuing the last step)")
