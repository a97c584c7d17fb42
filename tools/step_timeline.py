"""GPU timeline of the training step WITHOUT a profiler attached: HIP event pairs around every block-level call (video block /
text layer, forward and backward) and at the phase boundaries of FrozenInTime.forward, on whatever stream the call runs on.
Prints, per steady-state step: time inside block calls per (kind, direction, stream), the gaps between consecutive calls on the
calling stream grouped by what sits between them, and the phase spans.  `python tools/step_timeline.py [--steps 4]`"""
import argparse
import collections
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from egovlpv2_amd import hipops as ops                                   # noqa: E402
from egovlpv2_amd.config import PathConfig                               # noqa: E402
from egovlpv2_amd.synthetic import make_state_dict, make_batch           # noqa: E402
from egovlpv2_amd.model.model import FrozenInTime                        # noqa: E402
from egovlpv2_amd.model.loss import EgoNCE                               # noqa: E402
from egovlpv2_amd.trainer.trainer_egoclip import AllGather_multi         # noqa: E402

LOG = []
ON = [False]


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def wrap(cls, name, tag):
    orig = getattr(cls, name)

    def f(ctx, *a):
        if not ON[0]:
            return orig(ctx, *a)
        e0 = ev()
        r = orig(ctx, *a)
        LOG.append((tag, torch.cuda.current_stream().cuda_stream, e0, ev()))
        return r
    setattr(cls, name, staticmethod(f))


def mark(tag):
    if ON[0]:
        e = ev()
        LOG.append((tag, torch.cuda.current_stream().cuda_stream, e, e))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--dump', action='store_true', help='chronological list of the calls of the last step (both streams)')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    cfg = PathConfig(frames=16, drop_rate=0.1)
    model = FrozenInTime({'model': 'SpaceTimeTransformer', 'num_frames': cfg.frames, 'pretrained': True},
                         {'model': 'roberta-base', 'pretrained': True, 'input': 'text'}, path_config=cfg,
                         task_names='EgoNCE_MLM_ITM', compute_dtype=torch.bfloat16)
    model.load_state_dict(make_state_dict(cfg, 0), strict=True)
    model = model.to(dev)
    data, noun, verb = make_batch(cfg, 8, 32, 1234)
    data = {'video': data['video'].to(dev), 'text': {k: v.to(dev) for k, v in data['text'].items()},
            'text_mlm_ids': data['text_mlm_ids'].to(dev), 'text_mlm_labels': data['text_mlm_labels'].to(dev)}
    noun, verb = noun.to(dev), verb.to(dev)
    args = types.SimpleNamespace(world_size=1, rank=0)
    loss_fn = EgoNCE()
    conf = {'loss': {'type': 'EgoNCE'}}
    np.random.seed(1)
    torch.manual_seed(1)

    wrap(ops.VideoBlockFn, 'forward', 'vf')
    wrap(ops.VideoBlockFn, 'backward', 'vb')
    wrap(ops.TextLayerFn, 'forward', 'tf')
    wrap(ops.TextLayerFn, 'backward', 'tb')
    for nm in ('PatchEmbedFn', 'PatchTokensFn', 'VocabLinearFn', 'CrossEntropySumFn', 'TextEmbedFn'):
        c = getattr(ops, nm, None)
        if c is not None:
            wrap(c, 'forward', nm + '.f')
            wrap(c, 'backward', nm + '.b')

    def step():
        ops.invalidate_weight_cache()
        mark('step')
        for p in model.parameters():
            p.grad = None
        loss, ld, _ = model(data, noun, verb, AllGather_multi.apply, 1, args, conf, loss_fn, 0, task_names='EgoNCE_MLM_ITM')
        mark('fwd_end')
        loss.backward()
        mark('bwd_end')

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    ON[0] = True
    for _ in range(a.steps):
        step()
    mark('step')
    torch.cuda.synchronize()
    ON[0] = False

    base = LOG[0][2]
    rows = [(tag, st, base.elapsed_time(e0), base.elapsed_time(e1)) for tag, st, e0, e1 in LOG]
    main_st = rows[0][1]
    steps = [r[2] for r in rows if r[0] == 'step']
    print(f"steps: {[round(steps[i + 1] - steps[i], 2) for i in range(len(steps) - 1)]} ms (GPU timeline, calling stream)")
    # last full step
    t0, t1 = steps[-2], steps[-1]
    cur = [r for r in rows if t0 <= r[2] < t1 + 1e-6]
    inside = collections.defaultdict(lambda: [0.0, 0])
    for tag, st, b, e in cur:
        if b != e:
            k = (tag, 'main' if st == main_st else 'side')
            inside[k][0] += e - b
            inside[k][1] += 1
    print("time inside calls (last step):")
    for k, (ms, n) in sorted(inside.items(), key=lambda kv: -kv[1][0]):
        print(f"  {k[0]:22s} {k[1]:5s} n={n:4d}  {ms:8.2f} ms  avg {ms / n * 1e3:8.1f} us")
    mrows = sorted([r for r in cur if r[1] == main_st], key=lambda r: r[2])
    gaps = collections.defaultdict(lambda: [0.0, 0])
    big = []
    for p, q in zip(mrows, mrows[1:]):
        g = q[2] - p[3]
        k = (p[0], q[0])
        gaps[k][0] += g
        gaps[k][1] += 1
        big.append((g, p[0], q[0], p[3] - t0))
    print("gaps between consecutive calls / marks on the calling stream (last step):")
    for k, (ms, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"  {k[0]:20s} -> {k[1]:20s} n={n:4d}  {ms:8.2f} ms  avg {ms / n * 1e3:8.1f} us")
    print("largest single gaps:")
    for g, p, q, at in sorted(big, reverse=True)[:15]:
        print(f"  {g:8.3f} ms after {p} before {q} at t={at:.2f} ms")
    if a.dump:
        print("calls of the last step (start, end, duration in ms; M = calling stream, S = companion):")
        for tag, st, b, e in sorted(cur, key=lambda r: r[2]):
            print(f"  {'M' if st == main_st else 'S'} {b - t0:8.3f} {e - t0:8.3f} {e - b:7.3f}  {tag}")
    for tag in ('fwd_end', 'bwd_end'):
        for r in cur:
            if r[0] == tag:
                print(f"{tag} at {r[2] - t0:.2f} ms")


if __name__ == '__main__':
    main()
