"""actual bf16-mode errors against the reference's golden outputs (tests keep a margin above these)"""
import os, sys, types
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np, torch
from helpers import load_golden
from egovlpv2_amd.synthetic import make_state_dict, make_batch
from egovlpv2_amd.model.model import FrozenInTime
from egovlpv2_amd.model.loss import EgoNCE
from egovlpv2_amd.trainer.trainer_egoclip import AllGather_multi
def rel(a, b): return float(np.linalg.norm(a - b) / np.linalg.norm(b))
for name in ('base_f4', 'base_f16'):
    g, cfg, B, L, wseed, bseed = load_golden(name)
    for dtype in (torch.bfloat16,):
        sd = make_state_dict(cfg, wseed)
        data, noun, verb = make_batch(cfg, B, L, bseed)
        m = FrozenInTime({'model': 'SpaceTimeTransformer', 'num_frames': cfg.frames, 'pretrained': True}, {'model': 'roberta-base', 'pretrained': True, 'input': 'text'},
                         path_config=cfg, task_names='EgoNCE_MLM_ITM', compute_dtype=dtype)
        m.load_state_dict(sd, strict=True); m = m.cuda().eval()
        cu = {'video': data['video'].cuda(), 'text': {k: v.cuda() for k, v in data['text'].items()}, 'text_mlm_ids': data['text_mlm_ids'].cuda(), 'text_mlm_labels': data['text_mlm_labels'].cuda()}
        with torch.no_grad():
            r = m.infer(cu, task_names='EgoNCE')
        te, ve = rel(r['text_embeds'].float().cpu().numpy(), g['text_embeds']), rel(r['video_embeds'].float().cpu().numpy(), g['video_embeds'])
        np.random.seed(17); torch.manual_seed(17)
        loss, ld, ret = m(cu, noun.cuda(), verb.cuda(), AllGather_multi.apply, 1, types.SimpleNamespace(world_size=1, rank=0), {'loss': {'type': 'EgoNCE'}}, EgoNCE(), 0, task_names='EgoNCE_MLM_ITM')
        le = {k: abs(float(ld[k]) - float(g['loss_' + k])) / abs(float(g['loss_' + k])) for k in ('EgoNCE', 'loss_mlm', 'loss_itm', 'loss_total')}
        print(name, dtype, f"text_embeds {te:.2e} video_embeds {ve:.2e}", {k: f"{v:.2e}" for k, v in le.items()})
