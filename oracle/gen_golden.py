"""Generate golden vectors by IMPORTING THE REFERENCE (build container only).

Run:  python oracle/gen_golden.py [tiny|base_f4|base_f16|dual_tiny|dual_base_f4|all]

This is the only file in the repo that touches /root/reference at run time.  It applies the import
shims of SURVEY.md §8(c) (missing third-party packages, transformers 5.x API drift, hard-coded
checkpoint paths, hard-coded ``.cuda()``), builds the *untouched* reference ``FrozenInTime`` for a
named config, loads the seeded synthetic weights of ``egovlpv2_amd.synthetic.make_state_dict`` into
it, runs ``infer`` / ``forward`` + ``backward`` on the seeded synthetic batch and stores inputs'
recipe + outputs in ``tests/golden/<name>.npz``.  Only outputs are stored: weights and inputs are
re-derivable from (config, seed).  The reference's Python never travels: tests read the .npz only.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/EgoVLPv2'
sys.path.insert(0, REPO)

from egovlpv2_amd.config import PathConfig, tiny_config        # noqa: E402
from egovlpv2_amd.synthetic import make_state_dict, make_batch, make_relation  # noqa: E402


def import_reference():
    """SURVEY.md §8(c) shims 1-8, in order."""
    os.chdir(REF)                                   # the yml is opened relative to cwd
    import transformers, transformers.modeling_utils as mu, transformers.models.bert.modeling_bert  # noqa

    def stub(name, **kw):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        sys.modules[name] = m
        return m

    class DropPath(nn.Module):                      # timm 0.4.12 semantics; identity at p == 0 / eval
        def __init__(self, p=0.):
            super().__init__()
            self.p = p

        def forward(self, x):
            if self.p == 0. or not self.training:
                return x
            keep = 1 - self.p
            return x.div(keep) * x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)

    t = stub('timm')
    t.models = stub('timm.models')
    t.models.layers = stub('timm.models.layers', DropPath=DropPath,
                           to_2tuple=lambda x: x if isinstance(x, tuple) else (x, x),
                           trunc_normal_=torch.nn.init.trunc_normal_)
    for n in ('humanize', 'av', 'cv2', 'ffmpeg'):
        stub(n)
    stub('decord').bridge = types.SimpleNamespace(set_bridge=lambda *a, **k: None)
    stub('torchvision').transforms = stub('torchvision.transforms')
    mu.find_pruneable_heads_and_indices = mu.prune_linear_layer = lambda *a, **k: None
    sys.path.insert(0, REF)
    import model.model as mm
    from model import roberta as rb, video_transformer as vt
    import model.loss as ml
    import model.model_epic_charades as mec
    from trainer.trainer_egoclip import AllGather_multi
    rb.RobertaModel.init_weights = lambda self: self.apply(self._init_weights)
    rb.RobertaModel.get_extended_attention_mask = lambda self, m, shape, device=None, dtype=None: \
        (1.0 - m[:, None, None, :].to(torch.float32)) * torch.finfo(torch.float32).min
    rb.RobertaModel.get_head_mask = lambda self, hm, n, *a, **k: [None] * n
    vt.config_yaml['use_checkpoint'] = False          # re-entrant recompute only; numerics unchanged
    rb.config_yaml['use_checkpoint'] = False
    mec.config['use_checkpoint'] = False
    return types.SimpleNamespace(mm=mm, mec=mec, rb=rb, vt=vt, ml=ml, AllGather_multi=AllGather_multi)


def build_reference(R, cfg: PathConfig, dual=False):
    """dual=False: model/model.py FrozenInTime; dual=True: the fine-tune variant model/model_epic_charades.py (task 'Dual')"""
    M = R.mec if dual else R.mm
    from transformers import RobertaConfig
    depth, n_fuse = cfg.depth, cfg.n_fuse

    def from_pretrained(name, *a, **k):
        R.rb.NUM_FUSE_BLOCK = 12 - (depth - n_fuse)          # RobertaLayer hard-codes `12 - NUM_FUSE_BLOCK`
        return R.rb.RobertaModel(RobertaConfig(
            vocab_size=cfg.vocab, hidden_size=cfg.dim, num_hidden_layers=depth, num_attention_heads=cfg.heads,
            intermediate_size=cfg.dim * cfg.mlp_ratio, max_position_embeddings=cfg.max_pos, type_vocab_size=1,
            layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=0, eos_token_id=2,
            hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0))
    R.rb.RobertaModel.from_pretrained = staticmethod(from_pretrained)
    orig_load = torch.load
    torch.load = lambda p, *a, **k: {'cls_token': torch.zeros(1, 1, cfg.dim)} if 'jx_vit' in str(p) else orig_load(p, *a, **k)

    def stt(**kw):                                            # SpaceTimeTransformer hard-codes `i < 6`
        net = R.vt.SpaceTimeTransformer(img_size=cfg.img, depth=depth, **kw)
        for i in range(depth):
            net.blocks[i] = R.vt.SpaceTimeBlock(dim=cfg.dim, num_heads=cfg.heads, mlp_ratio=4., qkv_bias=True,
                                                time_init=kw.get('time_init', 'zeros'),
                                                dim_text=cfg.dim if i >= depth - n_fuse else None)
        return net
    M.SpaceTimeTransformer = stt
    ycfg = dict(M.config, use_checkpoint=False, num_layers=depth, num_fuse_block=n_fuse, drop_rate=0.0)
    try:
        kw = dict(task_names='Dual') if dual else {}
        m = M.FrozenInTime(video_params={'model': 'SpaceTimeTransformer', 'num_frames': cfg.frames, 'pretrained': True,
                                         'drop_path_rate': 0.0},
                           text_params={'model': 'roberta-base', 'pretrained': True, 'input': 'text'},
                           projection='minimal', projection_dim=cfg.proj_dim, load_checkpoint="", config=ycfg, **kw)
    finally:
        torch.load = orig_load
    return m


def enable_cpu_forward():
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    if not dist.is_initialized():
        dist.init_process_group('gloo', rank=0, world_size=1)
    torch.Tensor.cuda = lambda self, *a, **k: self        # loss.py:41, model.py:436 hard-code .cuda()


def sl(t, n=32):
    return t.detach().reshape(-1)[:n].float().numpy().copy()


def stats(t):
    t = t.detach().float()
    return np.array([t.mean().item(), t.abs().mean().item(), t.pow(2).mean().sqrt().item()], dtype=np.float64)


def run_case(R, name, cfg, B, L, wseed, bseed, out_dir):
    torch.manual_seed(0)
    m = build_reference(R, cfg).eval()
    sd = make_state_dict(cfg, wseed)
    ref_sd = m.state_dict()
    missing = sorted(set(ref_sd) - set(sd))
    extra = sorted(set(sd) - set(ref_sd))
    assert not missing and not extra, (missing[:5], extra[:5])
    for k, v in sd.items():
        assert tuple(ref_sd[k].shape) == tuple(v.shape), (k, ref_sd[k].shape, v.shape)
    m.load_state_dict(sd, strict=True)
    names = [k for k, _ in m.named_parameters()]
    data, noun, verb = make_batch(cfg, B, L, bseed)
    out = {'meta_cfg': np.array([cfg.depth, cfg.n_fuse, cfg.img, cfg.frames, B, L, wseed, bseed]),
           'param_names': np.array(names)}

    # ---- per-block traces of the dual (unfused) pass and the fused pass via hooks
    tr = {}
    hooks = []
    for i, blk in enumerate(m.video_model.blocks):
        hooks.append(blk.register_forward_hook(lambda mod, inp, o, i=i: tr.__setitem__(f'v{i}', o)))
    for i, lyr in enumerate(m.text_model.encoder.layer):
        hooks.append(lyr.register_forward_hook(lambda mod, inp, o, i=i: tr.__setitem__(f't{i}', o[0])))
    with torch.no_grad():
        r = m.infer(data, task_names='EgoNCE', ret={})
        out['text_embeds'] = r['text_embeds'].numpy()
        out['video_embeds'] = r['video_embeds'].numpy()
        for k, v in tr.items():
            out[f'dual_{k}_stats'] = stats(v)
            out[f'dual_{k}_slice'] = sl(v[:, -1])          # last token of every sample
        tr.clear()
        r = m.infer(data, task_names='ITM', ret={})
        out['itm_logits_plain'] = r['cross_attn_itm_logits'].numpy()
        for k, v in tr.items():
            out[f'fused_{k}_stats'] = stats(v)
            out[f'fused_{k}_slice'] = sl(v[:, -1])
        tr.clear()
        r = m.infer(dict(data), task_names='MLM', ret={})
        lg = r['cross_attn_mlm_logits']
        out['mlm_logits_slice'] = lg[..., :48].numpy()
        out['mlm_logits_lse'] = torch.logsumexp(lg, -1).numpy()
    for h in hooks:
        h.remove()

    # ---- full three-loss forward + backward with pinned RNG consumption
    enable_cpu_forward()
    rng_log = {'randperm': [], 'rand': [], 'multinomial': []}
    o_randperm, o_multinomial, o_rand = torch.randperm, torch.multinomial, np.random.rand
    torch.randperm = lambda *a, **k: (lambda x: (rng_log['randperm'].append(x.clone()), x)[1])(o_randperm(*a, **k))
    torch.multinomial = lambda *a, **k: (lambda x: (rng_log['multinomial'].append(int(x.item())), x)[1])(o_multinomial(*a, **k))
    np.random.rand = lambda *a: (lambda x: (rng_log['rand'].append(float(x)), x)[1])(o_rand(*a))
    try:
        np.random.seed(17)
        torch.manual_seed(17)
        args = types.SimpleNamespace(world_size=1, rank=0)
        m.zero_grad()
        loss, ld, ret = m(data, noun, verb, R.AllGather_multi.apply, 1, args, {'loss': {'type': 'EgoNCE'}},
                          R.ml.EgoNCE(), 0, task_names='EgoNCE_MLM_ITM')
        loss.backward()
    finally:
        torch.randperm, torch.multinomial, np.random.rand = o_randperm, o_multinomial, o_rand
    for k, v in ld.items():
        out['loss_' + k] = np.array(float(v.detach()), dtype=np.float64)
    full_ld = {k: float(v.detach()) for k, v in ld.items()}
    out['sim_v2t'] = ret['sim_v2t'].detach().numpy()
    out['itm_logits'] = ret['cross_attn_itm_logits'].detach().numpy()
    out['rng_randperm'] = rng_log['randperm'][0].numpy()
    out['rng_rand'] = np.array(rng_log['rand'])
    out['rng_multinomial'] = np.array(rng_log['multinomial'])
    gn = []
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        gn.append(p.grad.norm().item())
    out['grad_norms'] = np.array(gn, dtype=np.float64)
    pd = dict(m.named_parameters())
    fi = cfg.depth - 1
    for k in ('cls_token', 'video_model.cls_token', 'video_model.temporal_embed', 'video_model.patch_embed.proj.bias',
              f'video_model.blocks.{fi}.attn.alpha_i2t', f'text_model.encoder.layer.{fi}.alpha_t2i',
              f'video_model.blocks.0.timeattn.qkv.weight', f'video_model.blocks.{fi}.attn.qkv_text_i2t.weight',
              'text_model.embeddings.LayerNorm.weight', 'mlm_score.transform.LayerNorm.bias', 'itm_score.fc.weight',
              'txt_proj.0.weight', 'vid_proj.4.bias'):
        out['grad_slice::' + k] = sl(pd[k].grad, 64)
    # word-embedding rows that were actually looked up
    ids = torch.unique(torch.cat([data['text']['input_ids'].reshape(-1), data['text_mlm_ids'].reshape(-1)]))[:8]
    out['grad_word_rows_ids'] = ids.numpy()
    out['grad_word_rows'] = pd['text_model.embeddings.word_embeddings.weight'].grad[ids, :16].numpy()

    # ---- EgoNCE-only step (BASELINE.json configs[0] / [1])
    m.zero_grad()
    loss, ld, ret = m(data, noun, verb, R.AllGather_multi.apply, 1, args, {'loss': {'type': 'EgoNCE'}},
                      R.ml.EgoNCE(), 0, task_names='EgoNCE')
    loss.backward()
    out['egonce_only_loss'] = np.array(float(loss), dtype=np.float64)
    out['egonce_only_grad_norms'] = np.array([(p.grad.norm().item() if p.grad is not None else -1.0)
                                              for _, p in m.named_parameters()], dtype=np.float64)

    # ---- temporal-embed inflation 4 -> 16 (model.py:532-563)
    te = sd['video_model.temporal_embed']
    m.video_params['num_frames'] = 16 if cfg.frames != 16 else 32
    infl = m._inflate_positional_embeds({'video_model.temporal_embed': te.clone()})['video_model.temporal_embed']
    m.video_params['num_frames'] = cfg.frames
    out['inflate_frames'] = np.array(infl.shape[1])
    out['inflate_slice'] = infl[0, :, :8].numpy()

    os.makedirs(out_dir, exist_ok=True)
    np.savez_compressed(os.path.join(out_dir, name + '.npz'), **out)
    print(name, full_ld, 'egonce_only', float(loss.detach()), 'saved', len(out), 'arrays')


def run_dual_case(R, name, cfg, B, L, wseed, bseed, out_dir):
    """The fine-tune variant (model_epic_charades.py:410-444): Dual forward + backward with the loss of configs/ft/epic.json
    (AdaptiveMaxMarginRankingLoss on data['relation']) and of configs/ft/charades.json (NormSoftmaxLoss)."""
    torch.manual_seed(0)
    m = build_reference(R, cfg, dual=True).eval()
    sd = make_state_dict(cfg, wseed, tasks='Dual')
    ref_sd = m.state_dict()
    assert sorted(ref_sd) == sorted(sd), (sorted(set(ref_sd) - set(sd))[:5], sorted(set(sd) - set(ref_sd))[:5])
    m.load_state_dict(sd, strict=True)
    names = [k for k, _ in m.named_parameters()]
    data, _, _ = make_batch(cfg, B, L, bseed)
    data['relation'] = make_relation(B, bseed)
    out = {'meta_cfg': np.array([cfg.depth, cfg.n_fuse, cfg.img, cfg.frames, B, L, wseed, bseed]), 'param_names': np.array(names),
           'relation': data['relation'].numpy()}
    enable_cpu_forward()
    args = types.SimpleNamespace(world_size=1, rank=0)
    for ds, loss_fn in (('epic', R.ml.AdaptiveMaxMarginRankingLoss(margin=0.2)), ('charades', R.ml.NormSoftmaxLoss())):
        m.zero_grad()
        loss, ld, ret = m(data, R.AllGather_multi.apply, 1, args, {}, loss_fn, 0, task_names='Dual', dataset_name=ds)
        loss.backward()
        out[f'{ds}_loss'] = np.array(float(loss.detach()), dtype=np.float64)
        out[f'{ds}_sim_v2t'] = ret['sim_v2t'].detach().numpy()
        out[f'{ds}_grad_norms'] = np.array([(p.grad.norm().item() if p.grad is not None else -1.0) for _, p in m.named_parameters()],
                                           dtype=np.float64)
        pd = dict(m.named_parameters())
        for k in ('txt_proj.1.weight', 'vid_proj.0.bias', 'video_model.cls_token', 'text_model.embeddings.LayerNorm.weight'):
            out[f'{ds}_grad_slice::' + k] = sl(pd[k].grad, 64)
        print(name, ds, float(loss.detach()))
    with torch.no_grad():
        r = m.infer(data, task_names='Dual', ret={})
    out['text_embeds'] = r['text_embeds'].numpy()
    out['video_embeds'] = r['video_embeds'].numpy()
    os.makedirs(out_dir, exist_ok=True)
    np.savez_compressed(os.path.join(out_dir, name + '.npz'), **out)
    print(name, 'saved', len(out), 'arrays')


CASES = {
    # BASELINE.json configs[0] shapes (all three losses so that the fusion path is pinned too)
    'tiny': dict(cfg=tiny_config(), B=2, L=16, wseed=0, bseed=1234),
    # full-depth ViT-B/16 + RoBERTa-base at the reference's own 4-frame pre-training shape
    'base_f4': dict(cfg=PathConfig(frames=4), B=2, L=16, wseed=1, bseed=4321),
    # the full geometry of BASELINE.json configs[2] (12 + 12 layers, 16 x 224^2 frames, 32 tokens) at B = 2
    'base_f16': dict(cfg=PathConfig(frames=16), B=2, L=32, wseed=2, bseed=777),
}
DUAL_CASES = {
    # fine-tune variant (256-d heads, Dual task): tiny depth and the full ViT-B/16 + RoBERTa-base towers at 4 frames
    'dual_tiny': dict(cfg=tiny_config(proj_dim=256, proj_style='linear'), B=3, L=16, wseed=3, bseed=99),
    'dual_base_f4': dict(cfg=PathConfig(frames=4, proj_dim=256, proj_style='linear'), B=3, L=16, wseed=4, bseed=98),
}


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    out_dir = os.path.join(REPO, 'tests', 'golden')
    R = import_reference()
    for nm, c in CASES.items():
        if which in ('all', nm):
            run_case(R, nm, c['cfg'], c['B'], c['L'], c['wseed'], c['bseed'], out_dir)
    for nm, c in DUAL_CASES.items():
        if which in ('all', nm):
            run_dual_case(R, nm, c['cfg'], c['B'], c['L'], c['wseed'], c['bseed'], out_dir)
