"""Seeded synthetic weights and EgoClip-shaped batches.

There is no network for checkpoints or datasets, so weights are random-init of the reference
architecture and batches are synthetic clips/captions of the named shape.  Everything is derived
from (name, seed) so fixtures only need to store outputs, never tensors.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import torch

from .config import PathConfig


def param_shapes(cfg: PathConfig, tasks: str = 'EgoNCE_MLM_ITM') -> "OrderedDict[str, tuple]":
    """Names/shapes of the reference state dict (model.py:47-184,
    video_transformer.py:284-304, roberta.py:147-172,:223-257,:331-443, heads.py:15-46)."""
    D, H = cfg.dim, cfg.dim * cfg.mlp_ratio
    P = cfg.proj_dim
    s: "OrderedDict[str, tuple]" = OrderedDict()
    fused = ('MLM' in tasks) or ('ITM' in tasks)
    t = 'text_model.'
    s[t + 'embeddings.position_ids'] = (1, cfg.max_pos)
    s[t + 'embeddings.word_embeddings.weight'] = (cfg.vocab, D)
    s[t + 'embeddings.position_embeddings.weight'] = (cfg.max_pos, D)
    s[t + 'embeddings.token_type_embeddings.weight'] = (1, D)
    s[t + 'embeddings.LayerNorm.weight'] = (D,)
    s[t + 'embeddings.LayerNorm.bias'] = (D,)
    for i in range(cfg.depth):
        p = f'{t}encoder.layer.{i}.'
        for nm in ('query', 'key', 'value'):
            s[p + f'attention.self.{nm}.weight'] = (D, D)
            s[p + f'attention.self.{nm}.bias'] = (D,)
        s[p + 'attention.output.dense.weight'] = (D, D)
        s[p + 'attention.output.dense.bias'] = (D,)
        s[p + 'attention.output.LayerNorm.weight'] = (D,)
        s[p + 'attention.output.LayerNorm.bias'] = (D,)
        if i >= cfg.depth - cfg.n_fuse:
            for nm in ('query', 'key', 'value'):
                s[p + f'crossattention_t2i.self.{nm}.weight'] = (D, D)
                s[p + f'crossattention_t2i.self.{nm}.bias'] = (D,)
            s[p + 'crossattention_t2i.output.dense.weight'] = (D, D)
            s[p + 'crossattention_t2i.output.dense.bias'] = (D,)
            s[p + 'alpha_t2i'] = (1,)
        s[p + 'intermediate.dense.weight'] = (H, D)
        s[p + 'intermediate.dense.bias'] = (H,)
        s[p + 'output.dense.weight'] = (D, H)
        s[p + 'output.dense.bias'] = (D,)
        s[p + 'output.LayerNorm.weight'] = (D,)
        s[p + 'output.LayerNorm.bias'] = (D,)
    v = 'video_model.'
    s[v + 'cls_token'] = (1, 1, D)
    s[v + 'pos_embed'] = (1, cfg.n_patches + 1, D)
    s[v + 'temporal_embed'] = (1, cfg.frames, D)
    s[v + 'patch_embed.proj.weight'] = (D, 3, cfg.patch, cfg.patch)
    s[v + 'patch_embed.proj.bias'] = (D,)
    for i in range(cfg.depth):
        p = f'{v}blocks.{i}.'
        for nm in ('norm1', 'norm2', 'norm3'):
            s[p + nm + '.weight'] = (D,)
            s[p + nm + '.bias'] = (D,)
        for a in ('attn', 'timeattn'):
            s[p + a + '.qkv.weight'] = (3 * D, D)
            s[p + a + '.qkv.bias'] = (3 * D,)
            s[p + a + '.proj.weight'] = (D, D)
            s[p + a + '.proj.bias'] = (D,)
        if i >= cfg.depth - cfg.n_fuse:
            a = p + 'attn.'
            s[a + 'alpha_i2t'] = (1,)
            s[a + 'qkv_text_i2t.weight'] = (2 * D, D)
            s[a + 'qkv_text_i2t.bias'] = (2 * D,)
            s[a + 'qkv_i2t.weight'] = (D, D)
            s[a + 'qkv_i2t.bias'] = (D,)
            s[a + 'proj_i2t.weight'] = (D, D)
            s[a + 'proj_i2t.bias'] = (D,)
            s[a + 'norm_i2t_i.weight'] = (D,)
            s[a + 'norm_i2t_i.bias'] = (D,)
        s[p + 'mlp.fc1.weight'] = (H, D)
        s[p + 'mlp.fc1.bias'] = (H,)
        s[p + 'mlp.fc2.weight'] = (D, H)
        s[p + 'mlp.fc2.bias'] = (D,)
    s[v + 'norm.weight'] = (D,)
    s[v + 'norm.bias'] = (D,)
    if getattr(cfg, 'proj_style', 'mlp') == 'linear':         # model_epic_charades.py:116-119
        s['txt_proj.1.weight'] = (P, D)
        s['txt_proj.1.bias'] = (P,)
        s['vid_proj.0.weight'] = (P, D)
        s['vid_proj.0.bias'] = (P,)
    else:
        for nm in ('txt_proj', 'vid_proj'):
            s[nm + '.0.weight'] = (P, D)
            s[nm + '.2.weight'] = (P, P)
            s[nm + '.2.bias'] = (P,)
            s[nm + '.4.weight'] = (P, P)
            s[nm + '.4.bias'] = (P,)
    if fused:
        for nm in ('cross_modal_text_transform', 'cross_modal_video_transform',
                   'cross_modal_video_pooler.dense', 'cross_modal_text_pooler.dense'):
            s[nm + '.weight'] = (D, D)
            s[nm + '.bias'] = (D,)
        s['cls_token'] = (1, 1, D)
        s['norm.weight'] = (D,)
        s['norm.bias'] = (D,)
    if 'MLM' in tasks:
        s['mlm_score.bias'] = (cfg.vocab,)
        s['mlm_score.transform.dense.weight'] = (D, D)
        s['mlm_score.transform.dense.bias'] = (D,)
        s['mlm_score.transform.LayerNorm.weight'] = (D,)
        s['mlm_score.transform.LayerNorm.bias'] = (D,)
        s['mlm_score.decoder.weight'] = (cfg.vocab, D)
    if 'ITM' in tasks:
        s['itm_score.fc.weight'] = (2, 2 * D)
        s['itm_score.fc.bias'] = (2,)
    return s


def _gen(name: str, seed: int) -> torch.Generator:
    return torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


def make_state_dict(cfg: PathConfig, seed: int = 0, tasks: str = 'EgoNCE_MLM_ITM') -> "OrderedDict[str, torch.Tensor]":
    """De-degenerated random weights keyed by reference parameter name.

    The reference zero-initialises alpha_i2t/alpha_t2i, timeattn.qkv, the model cls_token and
    temporal_embed and sets timeattn.proj to ones (video_transformer.py:96-102,114,293; roberta.py:440;
    model.py:150), which would hide indexing bugs, so every tensor here is random: gates ~0.5, LayerNorm
    weights ~1, attention q/k projections with a larger std so that softmaxes are not flat.
    """
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in param_shapes(cfg, tasks).items():
        g = _gen(name, seed)
        if name.endswith('position_ids'):
            sd[name] = torch.arange(cfg.max_pos).expand(1, -1).clone()
            continue
        r = torch.randn(shape, generator=g, dtype=torch.float32)
        last = name.rsplit('.', 1)[-1]
        if 'alpha_' in name:
            t = 0.5 + 0.1 * r
        elif ('LayerNorm.weight' in name) or (last == 'weight' and len(shape) == 1):
            t = 1.0 + 0.05 * r
        elif last == 'bias' and 'mlm_score.bias' != name:
            t = 0.02 * r
        elif any(k in name for k in ('.qkv.weight', 'query.weight', 'key.weight', 'qkv_i2t.weight',
                                      'qkv_text_i2t.weight')):
            t = 0.05 * r
        else:
            t = 0.02 * r
        sd[name] = t
    return sd


def make_batch(cfg: PathConfig, batch: int, text_len: int, seed: int = 1234, mlm: bool = True):
    """One synthetic EgoClip-shaped batch: video ~ N(0,1) f32 (B,F,3,R,R); RoBERTa-style
    ids (<s>=0 ... </s>=2, pad=1) with per-sample lengths; 15 % MLM masking (80/10/10); multi-hot
    noun (582) / verb (118) vectors with at least one hot entry."""
    g = torch.Generator().manual_seed(seed)
    B, L = batch, text_len
    video = torch.randn(B, cfg.frames, 3, cfg.img, cfg.img, generator=g)
    ids = torch.full((B, L), cfg.pad_id, dtype=torch.int64)
    lens = torch.randint(min(8, L), L + 1, (B,), generator=g)
    hi = min(50260, cfg.vocab - 5)
    for b in range(B):
        n = int(lens[b])
        ids[b, 0] = 0
        ids[b, 1:n - 1] = torch.randint(3, hi, (n - 2,), generator=g)
        ids[b, n - 1] = 2
    mask = (ids != cfg.pad_id).to(torch.int64)
    data = {'video': video, 'text': {'input_ids': ids, 'attention_mask': mask}}
    if mlm:
        special = (ids == 0) | (ids == 2) | (ids == cfg.pad_id)
        pick = (torch.rand(B, L, generator=g) < 0.15) & ~special
        for b in range(B):                      # make sure every caption contributes >=1 label
            if not pick[b].any():
                pick[b, 1] = True
        labels = torch.where(pick, ids, torch.full_like(ids, -100))
        u = torch.rand(B, L, generator=g)
        mlm_ids = ids.clone()
        mlm_ids[pick & (u < 0.8)] = cfg.vocab - 1                      # <mask> = 50264
        rnd = torch.randint(3, hi, (B, L), generator=g)
        sel = pick & (u >= 0.8) & (u < 0.9)
        mlm_ids[sel] = rnd[sel]
        data['text_mlm_ids'] = mlm_ids
        data['text_mlm_labels'] = labels
    noun = (torch.rand(B, 582, generator=g) < 0.01).float()
    verb = (torch.rand(B, 118, generator=g) < 0.02).float()
    # share some tags across samples so that EgoNCE's positive mask is not just the diagonal
    noun[:, 7] = (torch.arange(B) % 2 == 0).float()
    verb[:, 3] = (torch.arange(B) % 2 == 0).float()
    noun[:, 11] = 1.0 - noun[:, 7]
    verb[:, 5] = 1.0 - verb[:, 3]
    return data, noun, verb


def make_relation(B: int, seed: int) -> torch.Tensor:
    """per-pair caption relevancy in (0.1, 1] (the `relation` field of an EPIC-Kitchens MIR sample,
    EpicKitchens_MIR_dataset.py:110,176): weights of AdaptiveMaxMarginRankingLoss in the fine-tune variant"""
    return 0.1 + 0.9 * torch.rand(B, generator=_gen('relation', seed), dtype=torch.float32)
