import os, sys, re, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import test_model_parity as T
from helpers import load_golden, oracle_setup
from egovlpv2_amd.synthetic import make_state_dict, make_batch
from oracle import ref_model as O
from collections import defaultdict
for name in ('base_f4', 'base_f16'):
    g, cfg, B, L, wseed, bseed = load_golden(name)
    sd0 = make_state_dict(cfg, wseed)
    data, noun, verb = make_batch(cfg, B, L, bseed)
    m = T._build(cfg, sd0, torch.bfloat16)
    np.random.seed(17); torch.manual_seed(17)
    loss, ld, ret = T._forward(m, data, noun, verb, 'EgoNCE_MLM_ITM')
    loss.backward()
    ref = json.load(open('/root/repo/tests/golden/autocast_grad_error.json'))[name]
    sd, data, noun, verb, oc = oracle_setup(cfg, B, L, wseed, bseed, requires_grad=True)
    np.random.seed(17); torch.manual_seed(17)
    oloss, _, _ = O.forward_losses(sd, data, noun, verb, oc, 'EgoNCE_MLM_ITM')
    oloss.backward()
    cls = lambda n: re.sub(r'\.\d+\.', '.*.', n)
    rows = []
    for n, p in m.named_parameters():
        if n.endswith('.key.bias'): continue
        a, r = p.grad.double().cpu().reshape(-1), sd[n].grad.double().reshape(-1)
        ea, gn = float((a - r).norm()), float(r.norm())
        ra, rn, numel = ref['grad_err'][n]
        rows.append((n, cls(n), ea, gn, ra, numel))
    rtot = (sum(r[4] ** 2 for r in rows) / sum(r[3] ** 2 for r in rows)) ** 0.5
    by = defaultdict(list)
    for r in rows: by[r[1]].append(r)
    crms = {c: (sum((r[4] / (r[3] + 1e-30)) ** 2 for r in rs) / len(rs)) ** 0.5 for c, rs in by.items()}
    big, small = [], []
    for n, c, ea, gn, ra, numel in rows:
        if numel == 1: continue
        lim = max(ra / (gn + 1e-30), crms[c], rtot if numel < 4096 else 0.0)
        ratio = (ea / (gn + 1e-30) - 5e-3) / lim
        (big if numel >= 4096 else small).append((ratio, n))
    big.sort(reverse=True); small.sort(reverse=True)
    print(name, 'big tensors: max ratios', [(round(r, 3), n) for r, n in big[:6]], 'p90', round(np.percentile([r for r, _ in big], 90), 3))
    print(name, 'small tensors: max ratios', [(round(r, 3), n) for r, n in small[:6]], 'p90', round(np.percentile([r for r, _ in small], 90), 3))
    del m
