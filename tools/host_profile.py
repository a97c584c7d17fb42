"""cProfile of the host side of one training step (where does the ~90 ms of enqueue time go?)."""
import cProfile, pstats, io, os, sys, types
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
def step():
    ops.invalidate_weight_cache()
    for p in model.parameters(): p.grad = None
    loss, ld, _ = model(data, noun, verb, AllGather_multi.apply, 1, args, {'loss': {'type': 'EgoNCE'}}, EgoNCE(), 0, task_names='EgoNCE_MLM_ITM')
    loss.backward()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3): step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue()[:6000])
