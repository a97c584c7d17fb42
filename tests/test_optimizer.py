"""Optimiser row: parameter grouping pinned against the reference's own set_optim_schedule.py
(tests/golden/optim_groups.json, produced by oracle/gen_golden_optim.py), fused HIP AdamW + schedule against the oracle's
restatement of transformers-4.30 AdamW."""
import json
import os

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_grouping_matches_reference():
    from egovlpv2_amd.set_optim_schedule import group_parameters
    from egovlpv2_amd.synthetic import param_shapes
    from egovlpv2_amd.config import PathConfig
    gold = json.load(open(os.path.join(REPO, 'tests', 'golden', 'optim_groups.json')))
    named = [(n, torch.nn.Parameter(torch.zeros(1))) for n in param_shapes(PathConfig(frames=4)) if not n.endswith('position_ids')]
    groups = group_parameters(named, 3e-5, 0.01, 4, 4)
    assert len(groups) == 6
    for g, r in zip(groups, gold['groups']):
        assert g['names'] == r['names']
        assert g['weight_decay'] == r['weight_decay'] and abs(g['lr'] - r['lr']) < 1e-12
    assert gold['kw'] == {'lr': 3e-5, 'eps': 1e-8, 'betas': [0.9, 0.98]}
    # the quirks the reference's substring rules produce
    decayed = set(groups[0]['names']) | set(groups[4]['names'])
    assert 'video_model.blocks.0.norm3.weight' in decayed and 'video_model.blocks.0.norm1.weight' not in decayed
    assert 'video_model.blocks.11.attn.norm_i2t_i.weight' in set(groups[4]['names'])


def test_schedule_lambdas():
    from egovlpv2_amd.set_optim_schedule import cosine_lambda
    from oracle.ref_optim import cosine_with_warmup
    f = cosine_lambda(10, 100)
    for s in (0, 1, 9, 10, 11, 55, 99, 100):
        assert abs(f(s) - cosine_with_warmup(s, 10, 100)) < 1e-12
    assert f(0) == 0.0 and abs(f(10) - 1.0) < 1e-12 and f(100) < 1e-12


@pytest.mark.gpu
def test_fused_adamw_matches_oracle():
    from egovlpv2_amd.set_optim_schedule import set_schedule
    from oracle.ref_optim import adamw_step, cosine_with_warmup
    torch.manual_seed(0)

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = torch.nn.Linear(300, 77)                    # decay + no-decay (bias), lr
            self.norm3 = torch.nn.LayerNorm(77)                         # norm3.weight decays (reference quirk), bias does not
            self.txt_proj = torch.nn.Linear(77, 20011)                  # head group (lr x 4); > 16384 elements -> several chunks
            self.cross_modal_x = torch.nn.Linear(33, 5)                 # cross-modal group
    m = M().cuda()
    cfg = {"optimizer": {"type": "AdamW", "args": {"lr": 3e-3, "weight_decay": 0.01, "lr_mult_head": 4, "lr_mult_cross_modal": 4}}}
    opt, sched = set_schedule(m, cfg, {"decay_power": "cosine", "end_lr": 1e-7}, 20, 3)
    ref = {n: p.detach().double().cpu().clone() for n, p in m.named_parameters()}
    st = {n: (torch.zeros_like(v), torch.zeros_like(v)) for n, v in ref.items()}
    groups_of = {}
    from egovlpv2_amd.set_optim_schedule import group_parameters
    for g in group_parameters(m.named_parameters(), 3e-3, 0.01, 4, 4):
        for n in g['names']:
            groups_of[n] = (g['lr'], g['weight_decay'])
    for step in range(1, 6):
        gen = torch.Generator().manual_seed(step)
        for n, p in m.named_parameters():
            g = torch.randn(p.shape, generator=gen)
            p.grad = g.cuda()
            lam = cosine_with_warmup(step - 1, 3, 20)                   # LambdaLR: lr used at step t is lambda(t-1 steps taken)
            base_lr, wd = groups_of[n]
            adamw_step(ref[n], g.double(), st[n][0], st[n][1], step, base_lr * lam, (0.9, 0.98), 1e-8, wd)
        opt.step()
        sched.step()
    for n, p in m.named_parameters():
        err = (p.detach().double().cpu() - ref[n]).abs().max().item()
        assert err < 2e-6, (n, err)


@pytest.mark.gpu
def test_fused_adamw_keeps_a_step_count_per_parameter():
    """HF AdamW semantics: 'step' (and with it the bias correction) advances only for parameters that had a gradient.  Two
    tensors of one group get gradients on different subsets of the steps; both must follow the oracle update with their own t.
    Also: a 1-element tensor in front of a larger one inside one flat buffer (a DDP bucket view) makes the second tensor's
    gradient 4-byte aligned only -- the kernel must take its scalar path for it."""
    from egovlpv2_amd.set_optim_schedule import FusedAdamW
    from oracle.ref_optim import adamw_step
    torch.manual_seed(1)
    a = torch.nn.Parameter(torch.randn(1000, device='cuda'))
    b = torch.nn.Parameter(torch.randn(3, 700, device='cuda'))
    flat = torch.zeros(1 + 2100, device='cuda')                          # gradient "bucket": [1 pad element | b.grad]
    opt = FusedAdamW([a, b], lr=1e-2, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
    ref = {n: p.detach().double().cpu().clone() for n, p in (('a', a), ('b', b))}
    st = {n: [torch.zeros_like(v), torch.zeros_like(v), 0] for n, v in ref.items()}
    for step in range(1, 8):
        gen = torch.Generator().manual_seed(100 + step)
        ga, gb = torch.randn(1000, generator=gen), torch.randn(3, 700, generator=gen)
        a.grad = ga.cuda() if step % 2 == 1 else None                    # a: steps 1, 3, 5, 7
        if step != 2:                                                    # b: every step but the second
            flat[1:].copy_(gb.reshape(-1))
            b.grad = flat[1:].view(3, 700)
            assert b.grad.data_ptr() % 16 != 0
        else:
            b.grad = None
        for n, g, has in (('a', ga, step % 2 == 1), ('b', gb, step != 2)):
            if has:
                st[n][2] += 1
                adamw_step(ref[n], g.double(), st[n][0], st[n][1], st[n][2], 1e-2, (0.9, 0.98), 1e-8, 0.01)
        opt.step()
    assert opt.state[a]['step'] == 4 and opt.state[b]['step'] == 6
    for n, p in (('a', a), ('b', b)):
        err = (p.detach().double().cpu() - ref[n]).abs().max().item()
        assert err < 2e-6, (n, err)


@pytest.mark.gpu
def test_checkpoint_resume_continues_fused_adamw(tmp_path):
    """base_trainer.py:412-495 format with the HIP optimiser: save after 2 steps, resume into a fresh model/optimiser,
    one more step on both -> identical parameters (state keys 'step'/'exp_avg'/'exp_avg_sq' as in HF AdamW)."""
    from egovlpv2_amd.set_optim_schedule import set_schedule
    from egovlpv2_amd.utils.checkpoint import save_checkpoint, resume_checkpoint

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(16, 8)
            self.norm = torch.nn.LayerNorm(8)
            self.itm_score = torch.nn.Linear(8, 2)
    ocfg = {"optimizer": {"type": "AdamW", "args": {"lr": 3e-3, "weight_decay": 0.01, "lr_mult_head": 4, "lr_mult_cross_modal": 4}}}
    ycfg = {"decay_power": "cosine", "end_lr": 1e-7}

    def grads(m, step):
        gen = torch.Generator().manual_seed(step)
        for p in m.parameters():
            p.grad = torch.randn(p.shape, generator=gen).cuda()
    torch.manual_seed(0)
    m1 = Tiny().cuda()
    o1, s1 = set_schedule(m1, ocfg, ycfg, 20, 3)
    for step in (1, 2):
        grads(m1, step); o1.step(); s1.step()
    path = str(tmp_path / 'ck.pth')
    save_checkpoint(path, m1, o1, s1, epoch=3, monitor_best=1.5, config=ocfg)
    ck = torch.load(path, map_location='cpu', weights_only=False)
    st0 = next(iter(ck['optimizer']['state'].values()))
    assert set(st0.keys()) == {'step', 'exp_avg', 'exp_avg_sq'}
    m2 = Tiny().cuda()
    o2, s2 = set_schedule(m2, ocfg, ycfg, 20, 3)
    start, best = resume_checkpoint(path, m2, o2, s2, config=ocfg, map_location='cuda')
    assert (start, best) == (4, 1.5)
    for m, o, s in ((m1, o1, s1), (m2, o2, s2)):
        grads(m, 3); o.step(); s.step()
    torch.cuda.synchronize()
    for a, b in zip(m1.parameters(), m2.parameters()):
        assert torch.equal(a, b)


def test_oracle_adamw_is_pinned_to_torch_adam_and_a_hand_computed_vector():
    """oracle/ref_optim.adamw_step (the restatement of transformers-4.30 AdamW that the fused HIP kernel is tested against) pinned two ways:
    (a) at weight_decay = 0 and eps = 0 the HF update lr sqrt(1 - b2^t) / (1 - b1^t) m / sqrt(v) IS torch.optim.Adam's
    lr (m / (1 - b1^t)) / sqrt(v / (1 - b2^t)) -- five steps on random gradients against torch's own implementation; (b) one step with
    eps and DECOUPLED decay applied AFTER the Adam update with the plain learning rate (p -= lr wd p: the HF order, not torch.optim.AdamW's
    decay-first), against arithmetic written out by hand."""
    import math
    from oracle.ref_optim import adamw_step
    torch.manual_seed(3)
    p0 = torch.randn(257, dtype=torch.float64)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=3e-3, betas=(0.9, 0.98), eps=0.0, weight_decay=0.0)
    p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    for t in range(1, 6):
        g = torch.randn(257, dtype=torch.float64) + 0.1
        ref.grad = g.clone()
        opt.step()
        adamw_step(p, g, m, v, t, 3e-3, betas=(0.9, 0.98), eps=0.0, weight_decay=0.0)
        assert torch.allclose(p, ref.detach(), rtol=1e-12, atol=1e-14), t
    # (b) by hand: p = 1, g = 0.5, lr = 0.1, betas (0.9, 0.98), eps = 1e-8, wd = 0.01, first step
    m1 = 0.1 * 0.5
    v1 = 0.02 * 0.25
    step_size = 0.1 * math.sqrt(1.0 - 0.98) / (1.0 - 0.9)
    p1 = 1.0 - step_size * m1 / (math.sqrt(v1) + 1e-8)
    p1 = p1 - 0.1 * 0.01 * p1
    pt, mt, vt = torch.tensor([1.0], dtype=torch.float64), torch.zeros(1, dtype=torch.float64), torch.zeros(1, dtype=torch.float64)
    adamw_step(pt, torch.tensor([0.5], dtype=torch.float64), mt, vt, 1, 0.1, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
    assert abs(float(pt) - p1) < 1e-15 and abs(p1 - 0.8991000141) < 1e-9, (float(pt), p1)
    assert abs(float(mt) - m1) < 1e-15 and abs(float(vt) - v1) < 1e-15


@pytest.mark.gpu
def test_grad_scaler_around_fused_adamw_on_flat_block_gradients():
    """The reference's AMP step (trainer/trainer_egoclip.py:143-149, base/base_trainer.py:334): scaler.scale(loss).backward();
    scaler.step(optimizer); scaler.update() -- around the fused AdamW, on gradients that are VIEWS of the block executor's flat buffers.
    (bf16 needs no loss scaling; the wrap must still behave: the reference's trainer always has it.)  (a) a scaled step equals the
    unscaled step (unscale_ divides the flat views in place; scale 2^10 is exact in fp32); (b) an inf in one gradient skips the step --
    parameters and optimiser state untouched -- and halves the scale."""
    import types
    import numpy as np
    from egovlpv2_amd.config import tiny_config
    from egovlpv2_amd.synthetic import make_state_dict, make_batch
    from egovlpv2_amd.model.model import FrozenInTime
    from egovlpv2_amd.model.loss import EgoNCE
    from egovlpv2_amd.trainer.trainer_egoclip import AllGather_multi
    from egovlpv2_amd.set_optim_schedule import set_schedule
    cfg = tiny_config()
    data, noun, verb = make_batch(cfg, 2, 16, 77)
    dev = {'video': data['video'].cuda(), 'text': {k: v.cuda() for k, v in data['text'].items()},
           'text_mlm_ids': data['text_mlm_ids'].cuda(), 'text_mlm_labels': data['text_mlm_labels'].cuda()}
    args = types.SimpleNamespace(world_size=1, rank=0)
    ocfg = {"optimizer": {"type": "AdamW", "args": {"lr": 1e-3, "weight_decay": 0.01, "lr_mult_head": 4, "lr_mult_cross_modal": 4}}}

    def build():
        m = FrozenInTime({'model': 'SpaceTimeTransformer', 'num_frames': cfg.frames, 'pretrained': True},
                         {'model': 'roberta-base', 'pretrained': True, 'input': 'text'}, path_config=cfg,
                         task_names='EgoNCE_MLM_ITM', compute_dtype=torch.float32)
        m.load_state_dict(make_state_dict(cfg, 0), strict=True)
        m = m.cuda()
        opt, _ = set_schedule(m, ocfg, {"decay_power": "cosine", "end_lr": 1e-7}, 20, 3)
        return m, opt

    def fwd(m):
        np.random.seed(5)
        torch.manual_seed(5)
        loss, _, _ = m(dev, noun.cuda(), verb.cuda(), AllGather_multi.apply, 1, args, {'loss': {'type': 'EgoNCE'}}, EgoNCE(), 0,
                       task_names='EgoNCE_MLM_ITM')
        return loss

    m0, o0 = build()
    o0.zero_grad(set_to_none=True)
    fwd(m0).backward()
    o0.step()
    m1, o1 = build()
    scaler = torch.amp.GradScaler('cuda', init_scale=2.0 ** 10, growth_interval=1000)
    o1.zero_grad(set_to_none=True)
    scaler.scale(fwd(m1)).backward()
    scaler.step(o1)
    scaler.update()
    torch.cuda.synchronize()
    worst = 0.0
    for (n, a), (_, b) in zip(m0.named_parameters(), m1.named_parameters()):
        worst = max(worst, ((a - b).norm() / (a.norm() + 1e-12)).item())
    assert worst < 1e-6, worst                      # (not bitwise: a scaled loss moves the roundings inside backward by a few ulp)
    assert scaler.get_scale() == 2.0 ** 10
    # (b) inf-skip
    before = {n: p.detach().clone() for n, p in m1.named_parameters()}
    steps_before = {n: int(o1.state[p]['step']) for n, p in m1.named_parameters() if p in o1.state and 'step' in o1.state[p]}
    o1.zero_grad(set_to_none=True)
    scaler.scale(fwd(m1)).backward()
    victim = dict(m1.named_parameters())['video_model.blocks.0.mlp.fc1.weight']
    victim.grad.view(-1)[7] = float('inf')           # a view of the block's flat gradient buffer
    scaler.step(o1)
    scaler.update()
    torch.cuda.synchronize()
    for n, p in m1.named_parameters():
        assert torch.equal(p.detach(), before[n]), n
        if n in steps_before:
            assert int(o1.state[p]['step']) == steps_before[n], n
    assert scaler.get_scale() == 2.0 ** 9
