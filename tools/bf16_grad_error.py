"""Per-tensor gradient errors of this build's bf16 mode against the oracle's fp32 gradients, next to the reference-under-autocast's own
(tests/golden/autocast_grad_error.json, oracle/ref_autocast_error.py --grads).  usage: python tools/bf16_grad_error.py [base_f4 base_f16] > out.json"""
import os, sys, types, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np, torch
from helpers import load_golden, oracle_setup
from oracle import ref_model as O
from egovlpv2_amd.model.model import FrozenInTime
from egovlpv2_amd.model.loss import EgoNCE
from egovlpv2_amd.trainer.trainer_egoclip import AllGather_multi

ref_all = json.load(open(os.path.join(REPO, 'tests', 'golden', 'autocast_grad_error.json')))
res = {}
for name in (sys.argv[1:] or ['base_f4']):
    g, cfg, B, L, wseed, bseed = load_golden(name)
    sd, data, noun, verb, oc = oracle_setup(cfg, B, L, wseed, bseed, requires_grad=True)
    m = FrozenInTime({'model': 'SpaceTimeTransformer', 'num_frames': cfg.frames, 'pretrained': True}, {'model': 'roberta-base', 'pretrained': True, 'input': 'text'},
                     path_config=cfg, task_names='EgoNCE_MLM_ITM', compute_dtype=torch.bfloat16)
    m.load_state_dict({k: v.detach() for k, v in sd.items()}, strict=True); m = m.cuda()
    cu = {'video': data['video'].cuda(), 'text': {k: v.cuda() for k, v in data['text'].items()}, 'text_mlm_ids': data['text_mlm_ids'].cuda(), 'text_mlm_labels': data['text_mlm_labels'].cuda()}
    np.random.seed(17); torch.manual_seed(17)
    loss, ld, ret = m(cu, noun.cuda(), verb.cuda(), AllGather_multi.apply, 1, types.SimpleNamespace(world_size=1, rank=0), {'loss': {'type': 'EgoNCE'}}, EgoNCE(), 0, task_names='EgoNCE_MLM_ITM')
    loss.backward()
    np.random.seed(17); torch.manual_seed(17)
    oloss, _, _ = O.forward_losses(sd, data, noun, verb, oc, 'EgoNCE_MLM_ITM')
    oloss.backward()
    out = {}
    for n, p in m.named_parameters():
        a, r = p.grad.double().cpu().reshape(-1), sd[n].grad.double().reshape(-1)
        out[n] = [float((a - r).norm()), float(r.norm()), int(r.numel())]
    res[name] = out
    ref = ref_all[name]['grad_err']
    names = [n for n in out if not n.endswith('.key.bias')]
    ours = np.array([out[n][0] / (out[n][1] + 1e-30) for n in names]); theirs = np.array([ref[n][0] / (ref[n][1] + 1e-30) for n in names])
    tot = (sum(out[n][0] ** 2 for n in names) / sum(out[n][1] ** 2 for n in names)) ** 0.5
    rtot = (sum(ref[n][0] ** 2 for n in names) / sum(ref[n][1] ** 2 for n in names)) ** 0.5
    ratio = ours / theirs
    print(name, f'whole gradient: ours {tot:.3e} reference-under-autocast {rtot:.3e}; per-tensor ratio ours/ref: median {np.median(ratio):.2f} p90 {np.percentile(ratio, 90):.2f} max {ratio.max():.2f}', file=sys.stderr)
    for i in np.argsort(-ratio)[:25]:
        print(f'   {names[i]:70s} ours {ours[i]:.3e} ref {theirs[i]:.3e} ratio {ratio[i]:.2f} numel {out[names[i]][2]}', file=sys.stderr)
json.dump(res, sys.stdout)
