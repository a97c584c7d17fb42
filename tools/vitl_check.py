"""ViT-L/14 + RoBERTa-large geometry (BASELINE.json configs[4] without the fp8 weights: d = 1024, 16 heads, 14 x 14 patches -> 256
patches per frame, S = 1 + F*256) at reduced depth against the CPU oracle: forward embeddings, three losses, gradient of the whole
step.  usage (GPU box): python tools/vitl_check.py [frames] [depth]"""
import math, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from egovlpv2_amd.config import PathConfig
from egovlpv2_amd.synthetic import make_state_dict, make_batch
from egovlpv2_amd.model.model import FrozenInTime
from egovlpv2_amd.model.loss import EgoNCE
from egovlpv2_amd.trainer.trainer_egoclip import AllGather_multi
from oracle import ref_model as O

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = PathConfig(depth=depth, n_fuse=depth // 2, img=224, patch=14, frames=frames, dim=1024, heads=16, proj_dim=1024)
B, L = 2, 16
sd = make_state_dict(cfg, 7)
data, noun, verb = make_batch(cfg, B, L, 77)
args = types.SimpleNamespace(world_size=1, rank=0)
for v in sd.values():
    if v.is_floating_point():
        v.requires_grad_(True)
np.random.seed(5); torch.manual_seed(5)
oloss, old, _ = O.forward_losses(sd, data, noun, verb, O.make_cfg(**cfg.as_dict()), 'EgoNCE_MLM_ITM')
oloss.backward()
dev = {'video': data['video'].cuda(), 'text': {k: v.cuda() for k, v in data['text'].items()}, 'text_mlm_ids': data['text_mlm_ids'].cuda(),
       'text_mlm_labels': data['text_mlm_labels'].cuda()}
for dt in (torch.float32, torch.bfloat16):
    m = FrozenInTime({'model': 'SpaceTimeTransformer', 'num_frames': frames, 'pretrained': True}, {'model': 'roberta-base', 'pretrained': True, 'input': 'text'},
                     path_config=cfg, compute_dtype=dt)
    m.load_state_dict({k: v.detach() for k, v in sd.items()}, strict=True)
    m = m.cuda().eval()
    np.random.seed(5); torch.manual_seed(5)
    loss, ld, _ = m(dev, noun.cuda(), verb.cuda(), AllGather_multi.apply, 1, args, {'loss': {'type': 'EgoNCE'}}, EgoNCE(), 0, task_names='EgoNCE_MLM_ITM')
    loss.backward()
    num = den = dot = gg = 0.0
    for k, p in m.named_parameters():
        a, b = p.grad.detach().double().cpu(), sd[k].grad.double()
        num += float((a - b).pow(2).sum()); den += float(b.pow(2).sum()); dot += float((a * b).sum()); gg += float(a.pow(2).sum())
    print(f"{dt}: S={cfg.seq} losses " + ", ".join(f"{k} {float(ld[k].detach()):.5f} (oracle {float(old[k]):.5f})" for k in ('EgoNCE', 'loss_mlm', 'loss_itm')) +
          f" | whole-gradient rel L2 {math.sqrt(num / den):.2e}, cosine {dot / math.sqrt(gg * den):.6f}", flush=True)
