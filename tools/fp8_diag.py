"""fp8 video path: deviations of the HIP model and of the dequantised oracle from the fp32 oracle (embeddings, losses, gradients)"""
import sys, math, types
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import ref_model as O, mx_quant as MX
from egovlpv2_amd.config import PathConfig
from egovlpv2_amd.synthetic import make_state_dict, make_batch
from helpers import rel_err
import test_model_parity as T

cfg = PathConfig(depth=2, n_fuse=1, img=112, frames=4, dim=512, heads=8, proj_dim=512)
B, L = 2, 16
def run_oracle(mx):
    sd = make_state_dict(cfg, 12)
    data, noun, verb = make_batch(cfg, B, L, 34)
    for v in sd.values():
        if v.is_floating_point(): v.requires_grad_(True)
    oc = O.make_cfg(**cfg.as_dict())
    import contextlib
    with (MX.mx_video_linears(O) if mx else contextlib.nullcontext()):
        np.random.seed(5); torch.manual_seed(5)
        loss, ld, _ = O.forward_losses(sd, data, noun, verb, oc, 'EgoNCE_MLM_ITM')
        loss.backward()
        with torch.no_grad():
            ov = O.compute_video(sd, data['video'], oc)
    return sd, data, noun, verb, {k: float(v) for k, v in ld.items() if torch.is_tensor(v) and v.numel() == 1}, ov
sd0, data, noun, verb, l0, v0 = run_oracle(False)
sd1, _, _, _, l1, v1 = run_oracle(True)
def grads(sd): return torch.cat([v.grad.double().reshape(-1) for k, v in sd.items() if v.is_floating_point() and v.grad is not None])
g0, g1 = grads(sd0), grads(sd1)
print('oracle mx vs fp32: embeds', rel_err(v1, v0), 'grad rel', float((g1 - g0).norm() / g0.norm()), 'losses', {k: (l1[k] - l0[k]) / l0[k] for k in l0})
for fp8 in (True, False):
    m = T._build(cfg, sd0, torch.bfloat16, video_fp8=fp8).eval()
    with torch.no_grad():
        r = m.infer(T._to_cuda(data), task_names='EgoNCE')
    np.random.seed(5); torch.manual_seed(5)
    loss, ld, _ = T._forward(m, data, noun, verb, 'EgoNCE_MLM_ITM')
    loss.backward()
    names = [k for k, v in sd0.items() if v.is_floating_point() and v.grad is not None]
    P = dict(m.named_parameters())
    g = torch.cat([P[k].grad.double().cpu().reshape(-1) for k in names])
    print('HIP fp8' if fp8 else 'HIP bf16', 'embeds vs fp32', rel_err(r['video_embeds'].float(), v0), 'vs mx', rel_err(r['video_embeds'].float(), v1),
          'grad vs fp32', float((g - g0).norm() / g0.norm()), 'vs mx', float((g - g1).norm() / g1.norm()),
          'cos fp32', float(torch.dot(g, g0) / g.norm() / g0.norm()), 'cos mx', float(torch.dot(g, g1) / g.norm() / g1.norm()),
          'losses vs fp32', {k: (float(ld[k]) - l0[k]) / l0[k] for k in ('EgoNCE', 'loss_mlm', 'loss_itm')},
          'vs mx', {k: (float(ld[k]) - l1[k]) / l1[k] for k in ('EgoNCE', 'loss_mlm', 'loss_itm')})
