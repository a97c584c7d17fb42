"""How far is the REFERENCE ITSELF from its fp32 values when it runs the way its trainer runs it -- under autocast
(trainer_egoclip.py:143; bf16 here, CPU autocast: matmuls / convolutions in bf16, LayerNorm, softmax, residual sums and losses
in fp32)?  Puts the bf16 mode of this build (activations stored in bf16) in context.  Build container only
(imports /root/reference through oracle/gen_golden.py).  usage: python oracle/ref_autocast_error.py [base_f4|base_f16|tiny] [--grads] [--write]"""
import os, sys, types
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G                                               # noqa: E402
from egovlpv2_amd.synthetic import make_state_dict, make_batch      # noqa: E402


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).norm() / b.norm()).item()


def main(name, grads=False):
    c = G.CASES[name]
    cfg, B, L = c['cfg'], c['B'], c['L']
    R = G.import_reference()
    torch.manual_seed(0)
    m = G.build_reference(R, cfg).eval()
    m.load_state_dict(make_state_dict(cfg, c['wseed']), strict=True)
    data, noun, verb = make_batch(cfg, B, L, c['bseed'])
    g = np.load(os.path.join(G.REPO, 'tests', 'golden', name + '.npz'))
    G.enable_cpu_forward()
    args = types.SimpleNamespace(world_size=1, rank=0)
    with torch.no_grad(), torch.autocast('cpu', dtype=torch.bfloat16):
        r = m.infer(data, task_names='EgoNCE', ret={})
    out = {'text_embeds': rel(r['text_embeds'].float(), g['text_embeds']), 'video_embeds': rel(r['video_embeds'].float(), g['video_embeds'])}
    print(f"{name}: reference under bf16 autocast vs its fp32 values: text_embeds {out['text_embeds']:.2e}, "
          f"video_embeds {out['video_embeds']:.2e}")
    np.random.seed(17)
    torch.manual_seed(17)
    with torch.autocast('cpu', dtype=torch.bfloat16):
        loss, ld, ret = m(data, noun, verb, R.AllGather_multi.apply, 1, args, {'loss': {'type': 'EgoNCE'}}, R.ml.EgoNCE(), 0,
                          task_names='EgoNCE_MLM_ITM')
    for k in ('EgoNCE', 'loss_mlm', 'loss_itm', 'loss_total'):
        ref = float(g['loss_' + k])
        out[k] = abs(float(ld[k]) - ref) / abs(ref)
        print(f"   {k}: {float(ld[k]):.6f} vs fp32 {ref:.6f}  (rel {out[k]:.2e})")
    if grads:
        # per-tensor gradient error of the reference under autocast against ITS OWN fp32 gradients (same weights, batch and pinned
        # ITM draws; trainer/trainer_egoclip.py:143-149 runs forward under autocast and backward on the scaled loss): relative L2 per
        # parameter tensor.  The yardstick of the bf16 gradient bounds in tests/test_model_parity.py.
        loss.backward()
        g16 = {n: p.grad.detach().double().clone() for n, p in m.named_parameters() if p.grad is not None}
        m.zero_grad(set_to_none=True)
        np.random.seed(17)
        torch.manual_seed(17)
        loss32, _, _ = m(data, noun, verb, R.AllGather_multi.apply, 1, args, {'loss': {'type': 'EgoNCE'}}, R.ml.EgoNCE(), 0,
                         task_names='EgoNCE_MLM_ITM')
        loss32.backward()
        names = [str(x) for x in g['param_names']]
        pd = dict(m.named_parameters())
        gn = np.array([pd[k].grad.norm().item() for k in names])
        assert np.allclose(gn, g['grad_norms'], rtol=1e-4, atol=1e-7), 'fp32 gradients differ from the committed fixture'
        gerr = {}
        for n in names:
            r = pd[n].grad.detach().double()
            gerr[n] = [float((g16[n] - r).norm()), float(r.norm()), int(r.numel())]
        out['grad_err'] = gerr                                       # name -> [|g_autocast - g_fp32|, |g_fp32|, numel]
        tot = (sum(v[0] ** 2 for v in gerr.values()) / sum(v[1] ** 2 for v in gerr.values())) ** 0.5
        out['grad_total'] = tot
        print(f"   whole gradient: rel L2 {tot:.3e}")
    return out


if __name__ == '__main__':
    # `--write`: (re)generate tests/golden/autocast_error.json -- the reference's own mixed-precision distance from its fp32
    # values, the yardstick of the bf16 acceptance tests (tests/test_model_parity.py)
    import json
    names = [a for a in sys.argv[1:] if not a.startswith('--')] or ['base_f4']
    res = {n: main(n, grads='--grads' in sys.argv) for n in names}
    if '--write' in sys.argv:
        path = os.path.join(G.REPO, 'tests', 'golden', 'autocast_error.json')
        old = json.load(open(path)) if os.path.exists(path) else {}
        gpath = os.path.join(G.REPO, 'tests', 'golden', 'autocast_grad_error.json')
        gold = json.load(open(gpath)) if os.path.exists(gpath) else {}
        for n, r in res.items():
            if 'grad_err' in r:                                      # per-tensor gradient errors live in their own file
                gold[n] = {'grad_err': r.pop('grad_err'), 'grad_total': r.pop('grad_total')}
        if gold:
            json.dump(gold, open(gpath, 'w'), sort_keys=True)
            print('wrote', gpath)
        for n, r in res.items():
            old.setdefault(n, {}).update(r)
        json.dump(old, open(path, 'w'), indent=1, sort_keys=True)
        print('wrote', path)
