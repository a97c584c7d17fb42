"""Input side of the EgoClip pre-training step: the sample format of the dataset and the per-step batch
assembly of the trainer, restated on plain tensors so that a loader can hand batches to FrozenInTime.forward unchanged.

  * ``egoclip_sample``   -- data_loader/EgoClip_EgoMCQ_dataset.py:95-130 (``_get_caption`` multi-hot tags, ``_get_train_item``
                            dict with the scene-aware negative);
  * ``collate_samples``  -- what torch's default collate makes of a list of such samples (tensors stacked, captions listed);
  * ``assemble_train_batch`` -- trainer/trainer_egoclip.py:112-139: positives and negatives concatenated (text list, video, noun /
                            verb vectors), the tokenizer call ``tokenizer(text, return_tensors='pt', padding='max_length',
                            max_length=15, truncation=True)``, the MLM collation of every row, device placement.

The pretrained RoBERTa vocabulary is not available offline; any object with the Hugging Face tokenizer call signature plugs in.
``HashTokenizer`` is a deterministic stand-in with RoBERTa's special ids for synthetic runs and tests -- it is NOT the
pretrained vocabulary.  Video clips may stay uint8 (F, 3, H, W): normalisation then happens inside the patchify kernel
(hipops.PatchTokensFn), which removes the 77 MB fp32 host-to-device copy of a configs[2] batch."""
from __future__ import annotations

import zlib

import torch

from .collate import mlm_collate, ROBERTA_VOCAB

NOUN_DIM, VERB_DIM = 582, 118            # EgoClip_EgoMCQ_dataset.py:30-31


def tag_vectors(noun_idx, verb_idx, noun_dim: int = NOUN_DIM, verb_dim: int = VERB_DIM):
    """multi-hot tag vectors of one narration (EgoClip_EgoMCQ_dataset.py:92-103: ``eval(sample['tag_noun'])`` index lists)"""
    noun_vec, verb_vec = torch.zeros(noun_dim), torch.zeros(verb_dim)
    for i in noun_idx:
        noun_vec[i] = 1
    for i in verb_idx:
        verb_vec[i] = 1
    return noun_vec, verb_vec


def egoclip_sample(video, caption, noun_idx, verb_idx, path='', dataset_name='EgoClip', neg=None):
    """one training sample in the reference's format (EgoClip_EgoMCQ_dataset.py:105-130).  video: (F, 3, H, W) float32 normalised
    (or uint8 raw); neg: optional (video, caption, noun_idx, verb_idx) of the scene-aware negative (same segment_id, :113-118)."""
    noun_vec, verb_vec = tag_vectors(noun_idx, verb_idx)
    out = {'video': video, 'text': caption, 'meta': {'raw_captions': caption, 'paths': path, 'dataset': dataset_name},
           'noun_vec': noun_vec, 'verb_vec': verb_vec}
    if neg is not None:
        nv, nc, nn_, nvb = neg
        noun_neg, verb_neg = tag_vectors(nn_, nvb)
        out.update({'video_neg': nv, 'text_neg': nc, 'noun_vec_neg': noun_neg, 'verb_vec_neg': verb_neg})
    return out


def collate_samples(samples):
    """torch.utils.data default collation of a list of ``egoclip_sample`` dicts: tensors stacked on a new batch axis, strings
    gathered into lists, the meta dict collated field by field"""
    out = {}
    for k in samples[0]:
        vals = [s[k] for s in samples]
        if torch.is_tensor(vals[0]):
            out[k] = torch.stack(vals, 0)
        elif isinstance(vals[0], dict):
            out[k] = {kk: [v[kk] for v in vals] for kk in vals[0]}
        else:
            out[k] = list(vals)
    return out


class HashTokenizer:
    """Deterministic word-level stand-in with RoBERTa's conventions (<s>=0, <pad>=1, </s>=2, ids 3..50259 for words) and the
    Hugging Face call signature the trainer uses.  NOT the pretrained roberta-base vocabulary."""

    bos, pad, eos = 0, 1, 2

    def __call__(self, text, return_tensors='pt', padding='max_length', max_length=15, truncation=True):
        assert return_tensors == 'pt'
        if isinstance(text, str):
            text = [text]
        rows = []
        for t in text:
            ids = [3 + zlib.crc32(w.lower().encode()) % (ROBERTA_VOCAB - 8) for w in t.split()]
            if truncation:
                ids = ids[:max_length - 2]
            rows.append([self.bos] + ids + [self.eos])
        width = max_length if padding == 'max_length' else max(len(r) for r in rows)
        input_ids = torch.full((len(rows), width), self.pad, dtype=torch.int64)
        for i, r in enumerate(rows):
            input_ids[i, :len(r)] = torch.tensor(r, dtype=torch.int64)
        return {'input_ids': input_ids, 'attention_mask': (input_ids != self.pad).to(torch.int64)}


def assemble_train_batch(data, tokenizer, device=None, mlm=True, max_length=15, generator=None):
    """trainer/trainer_egoclip.py:112-139 on one collated loader batch.  Returns (data, n_embeds, v_embeds) exactly as the step
    passes them to ``model(data, n_embeds, v_embeds, ...)``: with B loader samples and scene-aware negatives the model sees
    2B clips / captions (positives first, negatives after, :113-116)."""
    data = dict(data)
    if 'video_neg' in data:                                                     # :112-116
        data['text'] = list(data['text']) + list(data['text_neg'])
        data['video'] = torch.cat((data['video'], data['video_neg']), dim=0)
        data['noun_vec'] = torch.cat((data['noun_vec'], data['noun_vec_neg']), dim=0)
        data['verb_vec'] = torch.cat((data['verb_vec'], data['verb_vec_neg']), dim=0)
    if tokenizer is not None:                                                   # :119-121
        tok = tokenizer(data['text'], return_tensors='pt', padding='max_length', max_length=max_length, truncation=True)
        data['text'] = {k: tok[k] for k in ('input_ids', 'attention_mask')}
    if mlm:                                                                     # :123-133 (row-by-row list -> collator -> stacked)
        m = mlm_collate(data['text']['input_ids'], generator=generator)
        data['text_mlm_ids'], data['text_mlm_labels'] = m['input_ids'], m['labels']
    if device is not None:                                                      # :131-138
        nb = dict(non_blocking=True)
        data['text'] = {k: v.to(device, **nb) for k, v in data['text'].items()}
        data['video'] = data['video'].to(device, **nb)
        for k in ('text_mlm_ids', 'text_mlm_labels'):
            if k in data:
                data[k] = data[k].to(device, **nb)
    n_embeds = data['noun_vec'].to(device, non_blocking=True) if device is not None else data['noun_vec']
    v_embeds = data['verb_vec'].to(device, non_blocking=True) if device is not None else data['verb_vec']
    return data, n_embeds, v_embeds
