"""BASELINE.json configs[3] style inference point: infer('EgoNCE') forward only, B=16, 32 x 224^2
frames, 77 tokens, bf16, eval mode, no_grad.  Prints clips/s on one GPU (side measurement; bench.py stays the contract)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egovlpv2_amd.config import PathConfig
from egovlpv2_amd.synthetic import make_state_dict, make_batch
from egovlpv2_amd.model.model import FrozenInTime
B, F, L = 16, 32, 77
cfg = PathConfig(frames=F)
m = FrozenInTime({'model': 'SpaceTimeTransformer', 'num_frames': F, 'pretrained': True}, {'model': 'roberta-base', 'pretrained': True, 'input': 'text'},
                 path_config=cfg, task_names='EgoNCE', compute_dtype=torch.bfloat16)
m.load_state_dict(make_state_dict(cfg, 0, 'EgoNCE'), strict=True)
m = m.cuda().eval()
data, _, _ = make_batch(cfg, B, L, 5, mlm=False)
cu = {'video': data['video'].cuda(), 'text': {k: v.cuda() for k, v in data['text'].items()}}
with torch.no_grad():
    for _ in range(2):
        m.infer(cu, task_names='EgoNCE')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        r = m.infer(cu, task_names='EgoNCE')
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
fl = 1495.3e9 * B
print(f"infer EgoNCE B={B} F={F} L={L}: {dt*1e3:.1f} ms/batch = {B/dt:.1f} clips/s, {fl/dt/1e12:.0f} TFLOP/s (1495.3 GF/pair forward)")
