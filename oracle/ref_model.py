"""CPU oracle for the EgoVLPv2 pre-training hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a from-scratch fp32 restatement (plain torch on CPU, autograd for the
backward) of the algorithm the reference implements in
``/root/reference/EgoVLPv2/model/{model,video_transformer,roberta,heads,loss}.py``.
It exists only so that ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` have something to check the HIP path against
(and to time beside it).  Nothing under ``egovlpv2_amd/`` imports it.

Parity pin: the reference has NO tests/golden vectors for this path,
so this restatement is pinned against the reference itself, imported in the build
container by ``oracle/gen_golden.py`` (which needs /root/reference and therefore
never runs on the GPU box).  The outputs of that run are committed under
``tests/golden/*.npz`` and ``tests/test_oracle_golden.py`` re-checks this file
against them everywhere.

The model state is a flat ``dict[str, Tensor]`` that uses the reference's parameter
names, so a reference ``state_dict()`` can be fed in unchanged.
Every function cites the reference lines it restates (paths relative to
/root/reference/EgoVLPv2/).
"""
from __future__ import annotations

import copy
import math
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

F32_MIN = torch.finfo(torch.float32).min


def make_cfg(**kw):
    """Shape/config record.  Defaults = the reference architecture (ViT-B/16 + RoBERTa-base,
    model/model.py:73-83, EgoNCE_MLM_ITM_Config.yml)."""
    c = dict(depth=12, n_fuse=6, img=224, patch=16, frames=16, dim=768, heads=12, mlp_ratio=4,
             vocab=50265, max_pos=514, proj_dim=4096, proj_style='mlp', pad_id=1,
             eps_video=1e-5, eps_text=1e-5, eps_model_norm=1e-6, eps_mlm=1e-12)
    c.update(kw)
    c = SimpleNamespace(**c)
    c.n_patches = (c.img // c.patch) ** 2
    c.seq = 1 + c.frames * c.n_patches
    c.head_dim = c.dim // c.heads
    return c


# --------------------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------------------
def _ln(x, sd, prefix, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + '.weight'], sd[prefix + '.bias'], eps)


def _lin(x, sd, prefix, bias=True):
    return F.linear(x, sd[prefix + '.weight'], sd[prefix + '.bias'] if bias else None)


def _gelu(x):
    # exact erf GELU: nn.GELU (video_transformer.py:43), ACT2FN['gelu'] (roberta.py:402)
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def _softmax_attend(q, k, v):
    """video_transformer.py:35-39 -- plain softmax(q k^T) v on (problems, n, d) tensors."""
    p = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
    return p @ v


# --------------------------------------------------------------------------------------
# video side
# --------------------------------------------------------------------------------------
def patch_tokens(sd, video, cfg, cls_name):
    """Patch embedding + CLS + positional/temporal embedding.

    video_transformer.py:78-83 (Conv2d k=s=16 over every frame), :356-357 (flatten, frame-major
    token order), :360-371 (cls concat; pos_embed[1:] tiled per frame, temporal_embed repeated per
    patch, CLS gets pos_embed[0] only).  ``cls_name`` selects ``video_model.cls_token``
    (forward_features, :360) or the model-level ``cls_token`` (model.py:217, :301).
    The conv is restated as an unfold + matmul with the (c, ph, pw) weight flattening.
    """
    B, Fr, C, H, W = video.shape
    P = cfg.patch
    assert Fr == cfg.frames, (Fr, cfg.frames)          # video_transformer.py:80
    gh, gw = H // P, W // P
    x = video.reshape(B * Fr, C, gh, P, gw, P).permute(0, 2, 4, 1, 3, 5).reshape(B * Fr * gh * gw, C * P * P)
    w = sd['video_model.patch_embed.proj.weight'].reshape(cfg.dim, C * P * P)
    x = x @ w.t() + sd['video_model.patch_embed.proj.bias']
    x = x.reshape(B, Fr * gh * gw, cfg.dim)
    pos = sd['video_model.pos_embed']                      # (1, 1+N, D)
    tem = sd['video_model.temporal_embed']                 # (1, F, D)
    N = gh * gw
    body = pos[:, 1:, :].unsqueeze(1) + tem.unsqueeze(2)   # (1, F, N, D)
    body = body.reshape(1, Fr * N, cfg.dim)
    x = x + body
    cls = (sd[cls_name] + pos[:, :1, :]).expand(B, 1, cfg.dim)
    return torch.cat([cls, x], dim=1)


def divided_attention(xn, sd, prefix, cfg, mode):
    """VarAttention.forward without the text branch (video_transformer.py:117-152).

    q is pre-scaled by head_dim**-0.5 (:123, before the CLS split); the CLS query attends to all
    S keys (:129); every patch query attends to [CLS key ; its own frame (mode='space') or its own
    patch column across frames (mode='time')] (:131-141); heads are merged and projected (:150-152).
    """
    B, S, D = xn.shape
    h, dh, Fr, N = cfg.heads, cfg.head_dim, cfg.frames, cfg.n_patches
    qkv = _lin(xn, sd, prefix + '.qkv').reshape(B, S, 3, h, dh).permute(2, 0, 3, 1, 4)   # (3,B,h,S,dh)
    q, k, v = qkv[0] * (dh ** -0.5), qkv[1], qkv[2]
    cls_out = _softmax_attend(q[:, :, :1], k, v)                                          # (B,h,1,dh)

    def group(t):                       # (B,h,F*N,dh) -> (B,h,G,n,dh)
        t = t.reshape(B, h, Fr, N, dh)
        return t if mode == 'space' else t.transpose(2, 3)
    qg, kg, vg = group(q[:, :, 1:]), group(k[:, :, 1:]), group(v[:, :, 1:])
    G = qg.shape[2]
    kc = k[:, :, :1].unsqueeze(2).expand(B, h, G, 1, dh)
    vc = v[:, :, :1].unsqueeze(2).expand(B, h, G, 1, dh)
    og = _softmax_attend(qg, torch.cat([kc, kg], 3), torch.cat([vc, vg], 3))              # (B,h,G,n,dh)
    if mode != 'space':
        og = og.transpose(2, 3)
    out = torch.cat([cls_out, og.reshape(B, h, Fr * N, dh)], dim=2)                       # (B,h,S,dh)
    out = out.permute(0, 2, 1, 3).reshape(B, S, D)
    return _lin(out, sd, prefix + '.proj')


def i2t_cross(x, y, y_mask_add, sd, prefix, cfg):
    """Image->text gated cross attention (video_transformer.py:155-185).

    x is the *projected* space-attention output.  kv = qkv_text_i2t(y) laid out [2][h][dh] (:159-164),
    q = qkv_i2t(norm_i2t_i(x)) * dh**-0.5 (:166-172), additive mask (B,1,1,L) (:176-178), softmax over
    L, proj_i2t, x + alpha_i2t * y (:180-185).
    """
    B, S, D = x.shape
    L = y.shape[1]
    h, dh = cfg.heads, cfg.head_dim
    kv = _lin(y, sd, prefix + '.qkv_text_i2t').reshape(B, L, 2, h, dh).permute(2, 0, 3, 1, 4)
    q = _lin(_ln(x, sd, prefix + '.norm_i2t_i', cfg.eps_video), sd, prefix + '.qkv_i2t')
    q = q.reshape(B, S, h, dh).permute(0, 2, 1, 3) * (dh ** -0.5)
    s = q @ kv[0].transpose(-1, -2)
    if y_mask_add is not None:
        s = s + y_mask_add.view(B, 1, 1, L)
    o = (torch.softmax(s, -1) @ kv[1]).transpose(1, 2).reshape(B, S, D)
    return x + sd[prefix + '.alpha_i2t'] * _lin(o, sd, prefix + '.proj_i2t')


def video_block(x, i, sd, cfg, y=None, y_mask_add=None):
    """SpaceTimeBlock.forward (video_transformer.py:214-228): the space residual starts from x,
    NOT from the time residual (:222)."""
    p = f'video_model.blocks.{i}'
    t = divided_attention(_ln(x, sd, p + '.norm3', cfg.eps_video), sd, p + '.timeattn', cfg, 'time')
    tr = x + t
    s = divided_attention(_ln(tr, sd, p + '.norm1', cfg.eps_video), sd, p + '.attn', cfg, 'space')
    if y is not None:
        s = i2t_cross(s, y, y_mask_add, sd, p + '.attn', cfg)
    sr = x + s
    hdn = _gelu(_lin(_ln(sr, sd, p + '.norm2', cfg.eps_video), sd, p + '.mlp.fc1'))
    return sr + _lin(hdn, sd, p + '.mlp.fc2')


def video_features(sd, video, cfg):
    """SpaceTimeTransformer.forward_features (video_transformer.py:353-394): all blocks unfused,
    video_model.cls_token, video_model.norm (eps 1e-5), CLS row."""
    x = patch_tokens(sd, video, cfg, 'video_model.cls_token')
    for i in range(cfg.depth):
        x = video_block(x, i, sd, cfg)
    return _ln(x, sd, 'video_model.norm', cfg.eps_video)[:, 0]


# --------------------------------------------------------------------------------------
# text side
# --------------------------------------------------------------------------------------
def position_ids(input_ids, pad_id):
    """roberta.py:881-892: cumsum(mask)*mask + pad_id."""
    m = input_ids.ne(pad_id).to(torch.int64)
    return torch.cumsum(m, dim=1) * m + pad_id


def text_embeddings(sd, input_ids, cfg):
    """RobertaEmbeddings.forward (roberta.py:174-204): word + token_type(0) + position -> LayerNorm
    (dropout is identity in the parity setting p=0)."""
    p = 'text_model.embeddings'
    e = (sd[p + '.word_embeddings.weight'][input_ids]
         + sd[p + '.token_type_embeddings.weight'][0]
         + sd[p + '.position_embeddings.weight'][position_ids(input_ids, cfg.pad_id)])
    return _ln(e, sd, p + '.LayerNorm', cfg.eps_text)


def extended_mask(attention_mask):
    """get_extended_attention_mask, transformers 4.30 semantics (call sites roberta.py:826,
    model.py:251,338): (1 - m) * finfo(fp32).min broadcast as (B,1,1,L)."""
    return (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * F32_MIN


def _mha(hq, hkv, sd, prefix, cfg, mask_add):
    """RobertaSelfAttention.forward (roberta.py:257-327): separate q/k/v Linears, scores / sqrt(dh)
    + additive mask, softmax, context (heads merged)."""
    B, Lq, D = hq.shape
    Lk = hkv.shape[1]
    h, dh = cfg.heads, cfg.head_dim
    q = _lin(hq, sd, prefix + '.query').reshape(B, Lq, h, dh).transpose(1, 2)
    k = _lin(hkv, sd, prefix + '.key').reshape(B, Lk, h, dh).transpose(1, 2)
    v = _lin(hkv, sd, prefix + '.value').reshape(B, Lk, h, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    if mask_add is not None:
        s = s + mask_add
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, Lq, D)


def text_layer(hid, mask_add, i, sd, cfg, enc=None):
    """RobertaLayer.forward (roberta.py:444-505).  SelfOutput is dense only (:339-343).  With video
    states ``enc``: crossattention_t2i(attention_output, keys/values = enc, NO key mask (:274-277,
    encoder_attention_mask=None)), gated by alpha_t2i (:486); then attention.output.LayerNorm(a + h)
    (:488); FFN with exact GELU and output.LayerNorm(ffn + a) (:420-426, last_norm=True)."""
    p = f'text_model.encoder.layer.{i}'
    a = _lin(_mha(hid, hid, sd, p + '.attention.self', cfg, mask_add), sd, p + '.attention.output.dense')
    if enc is not None:
        c = _lin(_mha(a, enc, sd, p + '.crossattention_t2i.self', cfg, None), sd,
                 p + '.crossattention_t2i.output.dense')
        a = sd[p + '.alpha_t2i'] * c + a
    a = _ln(a + hid, sd, p + '.attention.output.LayerNorm', cfg.eps_text)
    f = _lin(_gelu(_lin(a, sd, p + '.intermediate.dense')), sd, p + '.output.dense')
    return _ln(f + a, sd, p + '.output.LayerNorm', cfg.eps_text)


def text_features(sd, input_ids, attention_mask, cfg):
    """RobertaModel.forward (roberta.py:761-878) -> last_hidden_state (B,L,D); all layers unfused."""
    hid = text_embeddings(sd, input_ids, cfg)
    m = extended_mask(attention_mask)
    for i in range(cfg.depth):
        hid = text_layer(hid, m, i, sd, cfg)
    return hid


# --------------------------------------------------------------------------------------
# model API (model/model.py)
# --------------------------------------------------------------------------------------
def _proj_mlp(x, sd, prefix):
    """txt_proj / vid_proj (model.py:105-115): Linear(no bias)-ReLU-Linear-ReLU-Linear."""
    x = torch.relu(F.linear(x, sd[prefix + '.0.weight']))
    x = torch.relu(_lin(x, sd, prefix + '.2'))
    return _lin(x, sd, prefix + '.4')


def _proj(x, sd, prefix, cfg):
    """projection heads: the pre-training MLP, or the fine-tune variant's txt ReLU-Linear / vid Linear
    (model_epic_charades.py:116-119)"""
    if getattr(cfg, 'proj_style', 'mlp') == 'linear':
        return _lin(torch.relu(x), sd, 'txt_proj.1') if prefix == 'txt_proj' else _lin(x, sd, 'vid_proj.0')
    return _proj_mlp(x, sd, prefix)


def compute_text(sd, text, cfg):
    """model.py:491-505 (model_epic_charades.py:447-460)."""
    return _proj(text_features(sd, text['input_ids'], text['attention_mask'], cfg)[:, 0], sd, 'txt_proj', cfg)


def compute_video(sd, video, cfg):
    """model.py:524-530 (model_epic_charades.py:481-487)."""
    return _proj(video_features(sd, video, cfg), sd, 'vid_proj', cfg)


def fused_stack(sd, video, input_ids, attention_mask, cfg, trace=None):
    """The fusion-in-backbone pass shared by the ITM and MLM branches of FrozenInTime.infer
    (model.py:209-271 / :293-357): model-level cls_token, ``depth - n_fuse`` unfused video blocks and
    text layers, then ``n_fuse`` fused steps in which BOTH sides read the other modality's state from
    before the step (:268-271)."""
    v = patch_tokens(sd, video, cfg, 'cls_token')
    t = text_embeddings(sd, input_ids, cfg)
    m = extended_mask(attention_mask)
    n_plain = cfg.depth - cfg.n_fuse
    for i in range(n_plain):
        v = video_block(v, i, sd, cfg)
        if trace is not None:
            trace[f'v{i}'] = v
    for i in range(n_plain):
        t = text_layer(t, m, i, sd, cfg)
        if trace is not None:
            trace[f't{i}'] = t
    for i in range(n_plain, cfg.depth):
        v_new = video_block(v, i, sd, cfg, y=t, y_mask_add=m)
        t = text_layer(t, m, i, sd, cfg, enc=v)
        v = v_new
        if trace is not None:
            trace[f'v{i}'] = v
            trace[f't{i}'] = t
    return v, t


def itm_logits(sd, video, input_ids, attention_mask, cfg):
    """ITM tail (model.py:275-290; heads.py:15-35): model-level norm (eps 1e-6) CLS row, t[:,0],
    cross-modal transforms, tanh poolers, cat([text, video]), Linear(2D, 2)."""
    v, t = fused_stack(sd, video, input_ids, attention_mask, cfg)
    vf = _ln(v, sd, 'norm', cfg.eps_model_norm)[:, 0]
    tf = _lin(t[:, 0], sd, 'cross_modal_text_transform')
    vf = _lin(vf, sd, 'cross_modal_video_transform')
    ct = torch.tanh(_lin(tf, sd, 'cross_modal_text_pooler.dense'))
    cv = torch.tanh(_lin(vf, sd, 'cross_modal_video_pooler.dense'))
    return _lin(torch.cat([ct, cv], -1), sd, 'itm_score.fc')


def mlm_logits(sd, video, mlm_ids, attention_mask, cfg):
    """MLM tail (model.py:360-365; heads.py:38-50): all L tokens -> cross_modal_text_transform ->
    BertPredictionHeadTransform (dense, GELU, LayerNorm eps 1e-12) -> decoder (no bias) + bias."""
    _, t = fused_stack(sd, video, mlm_ids, attention_mask, cfg)
    t = _lin(t, sd, 'cross_modal_text_transform')
    t = _ln(_gelu(_lin(t, sd, 'mlm_score.transform.dense')), sd, 'mlm_score.transform.LayerNorm', cfg.eps_mlm)
    return F.linear(t, sd['mlm_score.decoder.weight']) + sd['mlm_score.bias']


def sim_matrix(a, b, eps=1e-8):
    """model.py:576-584."""
    an = a / torch.clamp(a.norm(dim=1, keepdim=True), min=eps)
    bn = b / torch.clamp(b.norm(dim=1, keepdim=True), min=eps)
    return an @ bn.t()


def egonce(x, sim_v, sim_n, temperature=0.05, noun=True, verb=True):
    """EgoNCE.forward (loss.py:40-61).  Returns (loss, mask_bool, temperature)."""
    eye = torch.eye(x.shape[0], dtype=x.dtype)
    if noun and verb:
        mask = sim_v * sim_n + eye
    elif noun:
        mask = sim_n + eye
    elif verb:
        mask = sim_v + eye
    else:
        mask = eye
    mb = mask > 0
    i_sm = torch.softmax(x / temperature, dim=1)
    j_sm = torch.softmax(x.t() / temperature, dim=1)
    li = torch.log((i_sm * mb).sum(1)).mean()
    lj = torch.log((j_sm * mb).sum(1)).mean()
    return -li - lj, mb, temperature


def norm_softmax_loss(x, temperature=0.05):
    """NormSoftmaxLoss.forward (loss.py:19-31): returns (loss, temperature)."""
    li = torch.diag(torch.log_softmax(x / temperature, dim=1)).mean()
    lj = torch.diag(torch.log_softmax(x.t() / temperature, dim=1)).mean()
    return -li - lj, temperature


def max_margin_ranking_loss(x, weight=None, margin=0.2, adaptive=False):
    """MaxMarginRankingLoss / AdaptiveMaxMarginRankingLoss with fix_norm=True (loss.py:73-99 / :110-143): the mean over all
    off-diagonal (i, j) of relu(m_i - (x_ii - x_ij)) and of relu(m_i - (x_ii - x_ji)), m_i = margin (* weight_i)."""
    n = x.shape[0]
    d = torch.diag(x).unsqueeze(1)
    m = margin * weight.to(x.dtype).unsqueeze(1) if adaptive else margin
    off = ~torch.eye(n, dtype=torch.bool)
    rows = torch.relu(m - (d - x))[off]
    cols = torch.relu(m - (d - x.t()))[off]
    return torch.cat([rows, cols]).mean()


def dual_forward_loss(sd, data, cfg, dataset_name='charades', margin=0.2, temperature=0.05):
    """FrozenInTime.forward of the fine-tune variant, single rank (model_epic_charades.py:410-444) with the loss the reference
    configs pair with it: epic -> AdaptiveMaxMarginRankingLoss(margin) on data['relation'] (configs/ft/epic.json:57-62),
    charades -> NormSoftmaxLoss (configs/ft/charades.json:57-61).  Returns (loss, sim, text_embeds, video_embeds)."""
    te = compute_text(sd, data['text'], cfg)
    ve = compute_video(sd, data['video'], cfg)
    x = sim_matrix(te, ve)
    if dataset_name == 'epic':
        loss = max_margin_ranking_loss(x, data['relation'], margin, adaptive=True)
    elif dataset_name == 'charades':
        loss, _ = norm_softmax_loss(x, temperature)
    else:
        raise NameError(dataset_name)
    return loss, x, te, ve


def itm_sample(data, itm_labels_perm, weights_v2t, weights_t2v, all_video, all_ids, all_masks, rank):
    """Hard-negative assembly of model.py:449-468.  RNG consumption order is the contract:
    per negative one ``np.random.rand()`` then one ``torch.multinomial(w + 1e-9, 1)``.
    Returns (data_itm, neg_log) where neg_log lists (idx, kind, neg_idx)."""
    bsz = len(itm_labels_perm)
    d = copy.deepcopy(data)
    log = []
    for idx in range(bsz):
        own = rank * bsz + idx
        if itm_labels_perm[idx] == 1:
            d['video'][idx] = all_video[own]
            d['text']['input_ids'][idx] = all_ids[own]
            d['text']['attention_mask'][idx] = all_masks[own]
        elif np.random.rand() > 0.5:
            j = torch.multinomial(weights_t2v[idx] + 1e-9, 1).item()
            d['video'][idx] = all_video[j]
            d['text']['input_ids'][idx] = all_ids[own]
            d['text']['attention_mask'][idx] = all_masks[own]
            log.append((idx, 'video', j))
        else:
            j = torch.multinomial(weights_v2t[idx] + 1e-9, 1).item()
            d['video'][idx] = all_video[own]
            d['text']['input_ids'][idx] = all_ids[j]
            d['text']['attention_mask'][idx] = all_masks[j]
            log.append((idx, 'text', j))
    return d, log


def forward_losses(sd, data, n_embeds, v_embeds, cfg, task_names='EgoNCE_MLM_ITM', world=None):
    """FrozenInTime.forward (model.py:370-487) for one rank.

    ``world`` = None means world_size 1 (all-gathers are identities).  Otherwise it is a dict with
    'rank' and 'gather' (callable tensor -> concatenation over ranks, differentiable w.r.t. the local
    slice like AllGather_multi, trainer/trainer_egoclip.py:25-41); used by the multi-rank parity tests.
    Returns (loss, loss_dict, ret) exactly as the reference does.
    """
    rank = 0 if world is None else world['rank']
    gather = (lambda t: t) if world is None else world['gather']
    ret, loss_dict = {}, {}
    loss = None
    if 'EgoNCE' in task_names:                                           # model.py:380-400
        te = compute_text(sd, data['text'], cfg)
        ve = compute_video(sd, data['video'], cfg)
        ret.update(text_embeds=te, video_embeds=ve)
        ve_all, te_all = gather(ve), gather(te)
        n_all, v_all = gather(n_embeds), gather(v_embeds)
        out = sim_matrix(te_all, ve_all)
        loss, mask_bool, temp = egonce(out, sim_matrix(v_all, v_all), sim_matrix(n_all, n_all))
        ret.update(sim_v2t=out, sim_t2v=out.t())
        loss_dict['EgoNCE'] = loss
    if 'MLM' in task_names:                                              # model.py:404-422
        lg = mlm_logits(sd, data['video'], data['text_mlm_ids'], data['text']['attention_mask'], cfg)
        ret['cross_attn_mlm_logits'] = lg
        lg_all = gather(lg.reshape(-1, cfg.vocab))
        lab_all = gather(data['text_mlm_labels'].reshape(-1))
        loss_mlm = F.cross_entropy(lg_all, lab_all, ignore_index=-100)
        loss = loss + loss_mlm
        loss_dict['loss_mlm'] = loss_mlm
    if 'ITM' in task_names:                                              # model.py:426-483
        all_video = gather(data['video'])
        all_ids = gather(data['text']['input_ids'])
        all_masks = gather(data['text']['attention_mask'])
        bsz = data['video'].size(0)
        pos_len = bsz // 2
        labels = torch.cat([torch.ones(pos_len), torch.zeros(bsz - pos_len)])
        labels = labels[torch.randperm(labels.size(0))]
        with torch.no_grad():
            sl = slice(bsz * rank, bsz * (rank + 1))
            w_v2t = torch.softmax(ret['sim_v2t'][sl] / temp, dim=1).masked_fill(mask_bool[sl], 0)
            w_t2v = torch.softmax(ret['sim_t2v'][sl] / temp, dim=1).masked_fill(mask_bool[sl], 0)
        d_itm, neg_log = itm_sample(data, labels, w_v2t, w_t2v, all_video, all_ids, all_masks, rank)
        lg = itm_logits(sd, d_itm['video'], d_itm['text']['input_ids'], d_itm['text']['attention_mask'], cfg)
        ret['cross_attn_itm_logits'] = lg
        ret['_itm_labels'] = labels
        ret['_itm_neg_log'] = neg_log
        loss_itm = F.cross_entropy(gather(lg), gather(labels).long())
        loss = loss + 2 * loss_itm
        loss_dict['loss_itm'] = loss_itm
    loss_dict['loss_total'] = loss
    return loss, loss_dict, ret


def inflate_temporal_embed(load_embed, curr_frames, mode='bilinear'):
    """_inflate_positional_embeds (model.py:532-563): truncate, zero-pad or bilinear(align_corners)
    resize of (1, F_load, D) -> (1, F_curr, D)."""
    Fl = load_embed.shape[1]
    if Fl == curr_frames:
        return load_embed
    if Fl > curr_frames:
        return load_embed[:, :curr_frames]
    if mode == 'zeros':
        out = torch.zeros(load_embed.shape[0], curr_frames, load_embed.shape[2])
        out[:, :Fl] = load_embed
        return out
    m = 'bilinear' if mode == 'bilinear' else 'nearest'
    kw = dict(align_corners=True) if m == 'bilinear' else {}
    return F.interpolate(load_embed.unsqueeze(0), (curr_frames, load_embed.shape[2]), mode=m, **kw).squeeze(0)
