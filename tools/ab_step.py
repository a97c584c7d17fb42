"""In-process interleaved A/B of Python-level switches of the training step (environment variables read at every forward call):
python tools/ab_step.py "EGV_TAIL_STREAM=0" "EGV_ITM_DRAW_EARLY=0" ...   -> baseline (no override) and each setting, R rounds of K steps,
interleaved; prints median / min ms per step.  (C-side switches are read once per process: use tools/sweep_env.sh for those.)"""
import os, sys, time, types, statistics
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from egovlpv2_amd import hipops as ops
from egovlpv2_amd.config import PathConfig
from egovlpv2_amd.synthetic import make_state_dict, make_batch
from egovlpv2_amd.model.model import FrozenInTime
from egovlpv2_amd.model.loss import EgoNCE
from egovlpv2_amd.trainer.trainer_egoclip import AllGather_multi

settings = [''] + [a for a in sys.argv[1:] if '=' in a]
R, K = int(os.environ.get("AB_ROUNDS", "6")), 5
dev = torch.device('cuda', 0)
cfg = PathConfig(frames=16, drop_rate=0.1)
model = FrozenInTime({'model': 'SpaceTimeTransformer', 'num_frames': cfg.frames, 'pretrained': True}, {'model': 'roberta-base', 'pretrained': True, 'input': 'text'},
                     path_config=cfg, task_names='EgoNCE_MLM_ITM', compute_dtype=torch.bfloat16)
model.load_state_dict(make_state_dict(cfg, 0), strict=True)
model = model.to(dev)
data, noun, verb = make_batch(cfg, 8, 32, 1234)
data = {'video': data['video'].to(dev), 'text': {k: v.to(dev) for k, v in data['text'].items()}, 'text_mlm_ids': data['text_mlm_ids'].to(dev), 'text_mlm_labels': data['text_mlm_labels'].to(dev)}
noun, verb = noun.to(dev), verb.to(dev)
args = types.SimpleNamespace(world_size=1, rank=0)
np.random.seed(1); torch.manual_seed(1)


def step():
    ops.invalidate_weight_cache()
    for p in model.parameters():
        p.grad = None
    loss, ld, _ = model(data, noun, verb, AllGather_multi.apply, 1, args, {'loss': {'type': 'EgoNCE'}}, EgoNCE(), 0, task_names='EgoNCE_MLM_ITM')
    loss.backward()


def apply(s):
    saved = {}
    for kv in s.split():
        k, v = kv.split('=', 1)
        saved[k] = os.environ.get(k)
        os.environ[k] = v
    return saved


def restore(saved):
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


for _ in range(4):
    step()
torch.cuda.synchronize()
res = {s: [] for s in settings}
for r in range(R):
    for s in settings:
        saved = apply(s)
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            step()
        torch.cuda.synchronize()
        res[s].append((time.perf_counter() - t0) / K * 1e3)
        restore(saved)
for s in settings:
    v = res[s]
    print(f"{s or '(default)':50s} median {statistics.median(v):7.2f}  min {min(v):7.2f}  max {max(v):7.2f} ms/step")
