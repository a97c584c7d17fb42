"""CPU tests of the host-side "next" rows of SURVEY.md §8(f): EgoMCQ accuracy, checkpoint format / resume, MLM collation."""
import os
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def test_egomcq_accuracy_matches_loop_definition():
    from egovlpv2_amd.model.metric import egomcq_accuracy_metrics_ensemble, egomcq_accuracy_metrics_vtm
    g = torch.Generator().manual_seed(0)
    preds = torch.randn(200, 5, generator=g)
    labels = torch.randint(0, 5, (200,), generator=g)
    types = torch.randint(1, 3, (200,), generator=g)            # 1 = inter-video, 2 = intra-video
    # the reference's definition (metric.py:225-241), spelled out sample by sample
    want = {}
    for t, name in zip(sorted(set(types.tolist())), ["Inter-video", "Intra-video"]):
        idx = [i for i in range(200) if types[i] == t]
        want[name] = 100.0 * sum(int(preds[i].argmax() == labels[i]) for i in idx) / len(idx)
    for fn in (egomcq_accuracy_metrics_ensemble, egomcq_accuracy_metrics_vtm):
        got = fn(preds, labels, types)
        assert set(got) == set(want)
        for k in want:
            assert abs(got[k] - want[k]) < 1e-4


def test_checkpoint_round_trip_and_data_parallel_prefix(tmp_path):
    from egovlpv2_amd.utils.checkpoint import save_checkpoint, resume_checkpoint, match_data_parallel_keys
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    opt = torch.optim.AdamW(net.parameters(), lr=1e-2)
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0 / (1 + s))
    for _ in range(3):
        net(torch.randn(5, 4)).sum().backward()
        opt.step(); sch.step(); opt.zero_grad()
    cfg = {'arch': {'type': 'FrozenInTime'}, 'optimizer': {'type': 'AdamW'}}
    path = tmp_path / 'checkpoint-epoch7.pth'
    state = save_checkpoint(str(path), net, opt, sch, epoch=7, monitor_best=0.25, config=cfg)
    assert list(state.keys()) == ['arch', 'epoch', 'state_dict', 'optimizer', 'scheduler', 'monitor_best', 'config']   # base_trainer.py:421-429
    net2 = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    opt2 = torch.optim.AdamW(net2.parameters(), lr=1e-2)
    sch2 = torch.optim.lr_scheduler.LambdaLR(opt2, lambda s: 1.0 / (1 + s))
    start, best = resume_checkpoint(str(path), net2, opt2, sch2, config=cfg)
    assert (start, best) == (8, 0.25)
    for a, b in zip(net.parameters(), net2.parameters()):
        assert torch.equal(a, b)
    assert sch2.last_epoch == sch.last_epoch
    s1, s2 = opt.state_dict()['state'], opt2.state_dict()['state']
    assert all(torch.equal(s1[k]['exp_avg'], s2[k]['exp_avg']) for k in s1)
    # a checkpoint written from a DDP-wrapped model ('module.' keys) loads into a bare model and the other way round
    dp = {'module.' + k: v for k, v in net.state_dict().items()}
    assert list(match_data_parallel_keys(dp, list(net.state_dict().keys())).keys()) == list(net.state_dict().keys())
    assert list(match_data_parallel_keys(net.state_dict(), list(dp.keys())).keys()) == list(dp.keys())
    # a different optimiser type in the current config: weights load, optimiser state is left alone (base_trainer.py:486-491)
    opt3 = torch.optim.AdamW(net2.parameters(), lr=1e-2)
    resume_checkpoint(str(path), net2, opt3, None, config={'arch': cfg['arch'], 'optimizer': {'type': 'SGD'}})
    assert len(opt3.state_dict()['state']) == 0


def test_mlm_collate_matches_installed_transformers_and_statistics():
    from egovlpv2_amd.trainer.collate import mlm_collate, ROBERTA_MASK_ID, ROBERTA_VOCAB
    g = torch.Generator().manual_seed(3)
    B, L = 512, 15
    ids = torch.randint(3, 50000, (B, L), generator=g)
    ids[:, 0] = 0
    n = torch.randint(4, L + 1, (B,), generator=g)
    for b in range(B):
        ids[b, n[b] - 1] = 2
        ids[b, n[b]:] = 1
    out = mlm_collate(ids, generator=torch.Generator().manual_seed(11))
    lab, mi = out['labels'], out['input_ids']
    special = (ids == 0) | (ids == 1) | (ids == 2)
    picked = lab != -100
    assert not (picked & special).any()
    assert torch.equal(lab[picked], ids[picked]) and torch.equal(mi[~picked], ids[~picked])
    rate = picked.sum().item() / (~special).sum().item()
    assert abs(rate - 0.15) < 0.02
    frac_mask = (mi[picked] == ROBERTA_MASK_ID).float().mean().item()
    frac_keep = (mi[picked] == ids[picked]).float().mean().item()
    assert abs(frac_mask - 0.8) < 0.05 and abs(frac_keep - 0.1) < 0.04
    # same draws as transformers.DataCollatorForLanguageModeling under the same torch seed
    try:
        from transformers import DataCollatorForLanguageModeling
    except Exception:
        pytest.skip("transformers not importable")

    class Tok:                                                     # the four things the collator asks a tokenizer for
        mask_token = '<mask>'
        pad_token = '<pad>'
        pad_token_id = 1
        padding_side = 'right'

        def __len__(self):
            return ROBERTA_VOCAB

        def convert_tokens_to_ids(self, t):
            return ROBERTA_MASK_ID

        def get_special_tokens_mask(self, val, already_has_special_tokens=True):
            return [1 if int(v) in (0, 1, 2) else 0 for v in val]
    try:
        coll = DataCollatorForLanguageModeling(Tok(), mlm=True, mlm_probability=0.15)
        torch.manual_seed(5)
        ref = coll([ids[i] for i in range(64)])
    except Exception as e:                                        # stub tokenizer rejected by this transformers version
        pytest.skip(f"installed transformers collator not drivable with a stub tokenizer: {e}")
    torch.manual_seed(5)
    mine = mlm_collate(ids[:64])
    assert torch.equal(ref['input_ids'], mine['input_ids']) and torch.equal(ref['labels'], mine['labels'])


def test_egoclip_sample_format_and_batch_assembly():
    """sample dict of EgoClip_EgoMCQ_dataset.py:105-130 and the trainer's per-step assembly
    (trainer/trainer_egoclip.py:112-139): negatives appended after the positives, the tokenizer call shape, MLM collation."""
    import torch
    from egovlpv2_amd.trainer.batch import egoclip_sample, collate_samples, assemble_train_batch, HashTokenizer, tag_vectors
    from egovlpv2_amd.trainer.collate import mlm_collate
    g = torch.Generator().manual_seed(0)
    F, R = 4, 32
    caps = ['#C C opens the drawer', '#C C picks a knife from the table and cuts the onion on the chopping board slowly', '#O man X walks']
    negs = ['#C C closes the drawer', '#C C puts the knife down', '#C C looks around']
    samples = []
    for i in range(3):
        v = torch.randn(F, 3, R, R, generator=g)
        vn = torch.randn(F, 3, R, R, generator=g)
        samples.append(egoclip_sample(v, caps[i], [i, 5 + i], [i], path=f'v{i}.mp4', neg=(vn, negs[i], [7], [2, 3])))
    s0 = samples[0]
    assert set(s0) == {'video', 'text', 'video_neg', 'text_neg', 'meta', 'noun_vec', 'verb_vec', 'noun_vec_neg', 'verb_vec_neg'}
    assert s0['noun_vec'].shape == (582,) and s0['verb_vec'].shape == (118,) and s0['noun_vec'].sum() == 2 and s0['verb_vec_neg'].sum() == 2
    nv, vv = tag_vectors([1, 1, 3], [])
    assert nv.sum() == 2 and vv.sum() == 0                         # multi-hot, duplicates collapse
    batch = collate_samples(samples)
    assert batch['video'].shape == (3, F, 3, R, R) and batch['text'] == caps and batch['meta']['paths'] == ['v0.mp4', 'v1.mp4', 'v2.mp4']

    calls = []

    class Rec(HashTokenizer):
        def __call__(self, text, **kw):
            calls.append((list(text), kw))
            return super().__call__(text, **kw)
    gm = torch.Generator().manual_seed(5)
    data, n_emb, v_emb = assemble_train_batch(batch, Rec(), generator=gm)
    # positives first, negatives after (trainer_egoclip.py:113-116)
    assert calls == [(caps + negs, dict(return_tensors='pt', padding='max_length', max_length=15, truncation=True))]
    assert torch.equal(data['video'], torch.cat([batch['video'], batch['video_neg']], 0))
    assert torch.equal(n_emb, torch.cat([batch['noun_vec'], batch['noun_vec_neg']], 0)) and v_emb.shape == (6, 118)
    ids, am = data['text']['input_ids'], data['text']['attention_mask']
    assert ids.shape == (6, 15) and ids.dtype == torch.int64 and torch.equal(am, (ids != 1).long())
    assert (ids[:, 0] == 0).all() and ids[1, -1] == 2               # truncated to 15 with </s> kept
    assert ((ids == 2).sum(1) == 1).all()
    ref = mlm_collate(ids, generator=torch.Generator().manual_seed(5))
    assert torch.equal(data['text_mlm_ids'], ref['input_ids']) and torch.equal(data['text_mlm_labels'], ref['labels'])
    # without negatives the batch passes through unchanged
    plain = collate_samples([egoclip_sample(torch.zeros(F, 3, R, R), 'a b', [1], [1])])
    d2, n2, _ = assemble_train_batch(plain, HashTokenizer(), mlm=False)
    assert d2['video'].shape[0] == 1 and 'text_mlm_ids' not in d2 and n2.shape == (1, 582)


def test_ranking_losses_of_the_finetune_variant_match_the_reference_values():
    """NormSoftmaxLoss / (Adaptive)MaxMarginRankingLoss (reference model/loss.py:13-31,:65-143) on the similarity matrices the
    imported reference produced (tests/golden/dual_*.npz) reproduce its loss values; the fix_norm=False form equals the plain
    2 n^2-term mean."""
    import os
    import numpy as np
    import torch
    from egovlpv2_amd.model.loss import NormSoftmaxLoss, MaxMarginRankingLoss, AdaptiveMaxMarginRankingLoss
    for name in ('dual_tiny', 'dual_base_f4'):
        g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', name + '.npz'))
        x = torch.tensor(g['epic_sim_v2t'])
        w = torch.tensor(g['relation'])
        assert abs(float(AdaptiveMaxMarginRankingLoss(margin=0.2)(x, w)) - float(g['epic_loss'])) < 1e-6
        loss, temp = NormSoftmaxLoss()(torch.tensor(g['charades_sim_v2t']))
        assert abs(float(loss) - float(g['charades_loss'])) < 1e-5 and temp == 0.05
        n = x.shape[0]
        d = torch.diag(x)
        terms = [max(0.0, 0.2 - float(d[i] - x[i, j])) for i in range(n) for j in range(n)] + \
                [max(0.0, 0.2 - float(d[i] - x[j, i])) for i in range(n) for j in range(n)]
        assert abs(float(MaxMarginRankingLoss(0.2, fix_norm=False)(x)) - sum(terms) / len(terms)) < 1e-6
        off = [t for k, t in enumerate(terms) if (k % (n * n)) // n != (k % (n * n)) % n]
        assert abs(float(MaxMarginRankingLoss(0.2)(x)) - sum(off) / len(off)) < 1e-6


class _OpaqueConfig:
    """stands in for the ConfigParser object the reference trainer pickles under 'config' (base_trainer.py:412-436)"""

    def __init__(self):
        self.name = 'EgoClip_4f'


def test_load_checkpoint_roundtrip_with_pickled_config_module_prefix_and_frame_inflation(tmp_path):
    """FrozenInTime(load_checkpoint=...) (model.py:158-176): a checkpoint written the way the reference trainer writes it --
    'config' is an arbitrary pickled object (torch >= 2.6 needs weights_only=False for it), keys carry DistributedDataParallel's
    'module.' prefix, and the temporal embedding has 2 frames while the model is built for 4 (bilinear inflation, :532-563;
    values checked against oracle.inflate_temporal_embed, itself pinned by tests/golden)."""
    import torch
    from egovlpv2_amd.config import tiny_config
    from egovlpv2_amd.model.model import FrozenInTime
    from egovlpv2_amd.synthetic import make_state_dict
    from oracle import ref_model as O
    cfg2 = tiny_config(frames=2)
    sd2 = make_state_dict(cfg2, 5)
    path = str(tmp_path / 'ckpt.pth')
    torch.save({'arch': 'FrozenInTime', 'epoch': 3, 'config': _OpaqueConfig(),
                'state_dict': {'module.' + k: v for k, v in sd2.items()}}, path)
    cfg4 = tiny_config(frames=4)
    vp = {'model': 'SpaceTimeTransformer', 'num_frames': 4, 'pretrained': True}
    tp = {'model': 'roberta-base', 'pretrained': True, 'input': 'text'}
    m = FrozenInTime(vp, tp, path_config=cfg4, load_checkpoint=path, compute_dtype=torch.float32)
    got = m.state_dict()
    for k, v in sd2.items():
        if k == 'video_model.temporal_embed':
            want = O.inflate_temporal_embed(v, 4, 'bilinear')
            assert got[k].shape == (1, 4, cfg4.dim) and torch.allclose(got[k], want, atol=1e-6)
        else:
            assert torch.equal(got[k], v), k
    # 'zeros' inflation and truncation (load_f > curr_f)
    m0 = FrozenInTime(vp, tp, path_config=cfg4, load_checkpoint=path, load_temporal_fix='zeros', compute_dtype=torch.float32)
    te = m0.state_dict()['video_model.temporal_embed']
    assert torch.equal(te[:, :2], sd2['video_model.temporal_embed']) and float(te[:, 2:].abs().max()) == 0.0
    torch.save({'config': _OpaqueConfig(), 'state_dict': make_state_dict(cfg4, 6)}, path)
    m1 = FrozenInTime({**vp, 'num_frames': 2}, tp, path_config=cfg2, load_checkpoint=path, compute_dtype=torch.float32)
    assert torch.equal(m1.state_dict()['video_model.temporal_embed'], make_state_dict(cfg4, 6)['video_model.temporal_embed'][:, :2])


def test_pretrained_tower_ingestion(tmp_path):
    """model.py:69 / :80-94: a RoBERTa checkpoint (HF names, 'roberta.' prefix, lm_head and pooler present) goes into
    text_model.*, a timm ViT checkpoint ('module.'-prefixed here, with a classifier head) into video_model.* with strict=False
    semantics: matching names are loaded, the rest keeps its init, unknown names are ignored, a wrong shape is an error."""
    import pytest
    import torch
    from egovlpv2_amd.config import tiny_config
    from egovlpv2_amd.model.model import FrozenInTime
    from egovlpv2_amd.synthetic import make_state_dict
    cfg = tiny_config()
    src = make_state_dict(cfg, 11)
    rob = {'roberta.' + k[len('text_model.'):]: v for k, v in src.items() if k.startswith('text_model.') and 'crossattention' not in k and 'alpha' not in k}
    rob['lm_head.dense.weight'] = torch.zeros(4, 4)
    rob['roberta.pooler.dense.weight'] = torch.zeros(cfg.dim, cfg.dim)
    vit_names = [k for k in src if k.startswith('video_model.') and not any(t in k for t in ('timeattn', 'temporal_embed', 'norm3', 'i2t'))]
    vit = {'module.' + k[len('video_model.'):]: src[k] for k in vit_names}
    vit['module.head.weight'] = torch.zeros(10, cfg.dim)
    torch.save(rob, tmp_path / 'roberta.bin')
    torch.save(vit, tmp_path / 'vit.pth')
    vp = {'model': 'SpaceTimeTransformer', 'num_frames': cfg.frames, 'pretrained': True, 'pretrained_path': str(tmp_path / 'vit.pth')}
    tp = {'model': 'roberta-base', 'pretrained': True, 'input': 'text', 'pretrained_path': str(tmp_path / 'roberta.bin')}
    base = FrozenInTime({k: v for k, v in vp.items() if k != 'pretrained_path'}, {k: v for k, v in tp.items() if k != 'pretrained_path'},
                        path_config=cfg, compute_dtype=torch.float32, init_seed=3).state_dict()
    got = FrozenInTime(vp, tp, path_config=cfg, compute_dtype=torch.float32, init_seed=3).state_dict()
    loaded = 0
    for k, v in got.items():
        from_ckpt = (k.startswith('text_model.') and 'roberta.' + k[len('text_model.'):] in rob and not k.endswith('position_ids')) or k in vit_names
        if from_ckpt:
            assert torch.equal(v, src[k]), k
            loaded += 1
        else:
            assert torch.equal(v, base[k]), k                # same seed -> same init as without checkpoints
    assert loaded > 40
    bad = dict(vit)
    bad['module.pos_embed'] = torch.zeros(1, 3, cfg.dim)
    torch.save(bad, tmp_path / 'bad.pth')
    with pytest.raises(RuntimeError, match='size mismatch'):
        FrozenInTime(dict(vp, pretrained_path=str(tmp_path / 'bad.pth')), tp, path_config=cfg, compute_dtype=torch.float32)
