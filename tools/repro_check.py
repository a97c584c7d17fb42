"""which gradients differ between identical training steps?  (debug aid for tests/test_model_parity.py::test_training_step_is_bitwise_reproducible)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from egovlpv2_amd.config import PathConfig
from egovlpv2_amd.synthetic import make_state_dict, make_batch
import test_model_parity as T
cfg = PathConfig(frames=16, depth=4, n_fuse=2, drop_rate=0.1)
B, L = 8, 32
sd = make_state_dict(cfg, 21)
data, noun, verb = make_batch(cfg, B, L, 2024)
m = T._build(cfg, sd, torch.bfloat16).train()
runs = []
from egovlpv2_amd import hipops as _ops
_orig_wgrad = _ops.wgrad
REC = []
def _wgrad(dy, x, M, N, K, **kw):
    if M <= 16 and N == 4096 and K == 4096 and os.environ.get('REC', '1') == '1':
        a, b_ = dy.clone(), x.clone()
        out = _orig_wgrad(dy, x, M, N, K, **kw)
        REC[-1].append((a, b_, dy.clone(), x.clone(), (out[0] if isinstance(out, tuple) else out)))
        return out
    return _orig_wgrad(dy, x, M, N, K, **kw)
_ops.wgrad = _wgrad
for _ in range(int(os.environ.get('RUNS', '4'))):
    m.zero_grad(set_to_none=True)
    REC.append([])
    m.seed_dropout(123)
    np.random.seed(3); torch.manual_seed(3)
    loss, ld, _ = T._forward(m, data, noun, verb, 'EgoNCE_MLM_ITM')
    loss.backward()
    torch.cuda.synchronize()
    runs.append((float(loss.detach()), {n: p.grad.clone() for n, p in m.named_parameters()}))
names = list(runs[0][1])
print('first params:', names[:3])
for i, r in enumerate(runs[1:], 1):
    bad = [(n, float((runs[0][1][n] - r[1][n]).abs().max()), float(runs[0][1][n].abs().max())) for n in names if not torch.equal(runs[0][1][n], r[1][n])]
    for n, _, _ in bad[:3]:
        dif = (runs[0][1][n] != r[1][n]).flatten()
        ix = dif.nonzero().flatten()
        a, b_ = runs[0][1][n].flatten(), r[1][n].flatten()
        print(f'   {n}: {int(dif.sum())} of {dif.numel()} elements differ, flat index range [{int(ix[0])}, {int(ix[-1])}], rows {int(ix[0]) // runs[0][1][n].shape[-1]}..{int(ix[-1]) // runs[0][1][n].shape[-1]}; sample run0 {a[ix[:4]].tolist()} run{i} {b_[ix[:4]].tolist()}')
    print(f'run {i} vs 0: loss equal {runs[0][0] == r[0]}, {len(bad)} of {len(names)} tensors differ', bad[:12])

for i in range(1, len(REC)):
    for j, (r0, ri) in enumerate(zip(REC[0], REC[i])):
        eq = [bool(torch.equal(u, v)) for u, v in zip(r0, ri)]
        if not all(eq):
            print(f'run {i} small-M call {j}: dy before {eq[0]}, x before {eq[1]}, dy after {eq[2]}, x after {eq[3]}, dW {eq[4]}')
