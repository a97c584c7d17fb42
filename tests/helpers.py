"""Shared helpers for the parity tests (oracle side)."""
import os

import numpy as np
import torch

from egovlpv2_amd.config import PathConfig
from egovlpv2_amd.synthetic import make_state_dict, make_batch, make_relation
from oracle import ref_model as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# The CPU oracle runs inside the parity tests (and inside their spawned workers, which import this module): on a many-core host torch's
# default -- one thread per logical CPU -- oversubscribes its GEMMs (EPYC 9575F, 256 logical CPUs: the oracle's full-size step takes 18 s on
# 64 threads and 3 x that on 128; tests/test_model_parity.py::test_base_f16_vs_golden[bf16] 230 s against 17 s).  64 at most.
torch.set_num_threads(max(1, min(torch.get_num_threads(), 64)))


def load_golden(name):
    g = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    depth, n_fuse, img, frames, B, L, wseed, bseed = [int(x) for x in g['meta_cfg']]
    cfg = PathConfig(depth=depth, n_fuse=n_fuse, img=img, frames=frames)
    return g, cfg, B, L, wseed, bseed


def load_golden_dual(name):
    """fixtures of the fine-tune variant (model_epic_charades.py): 256-d 'linear' heads, task Dual"""
    g = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    depth, n_fuse, img, frames, B, L, wseed, bseed = [int(x) for x in g['meta_cfg']]
    cfg = PathConfig(depth=depth, n_fuse=n_fuse, img=img, frames=frames, proj_dim=256, proj_style='linear')
    return g, cfg, B, L, wseed, bseed


def dual_batch(cfg, B, L, bseed):
    data, _, _ = make_batch(cfg, B, L, bseed)
    data['relation'] = make_relation(B, bseed)
    return data


def oracle_setup(cfg, B, L, wseed, bseed, requires_grad=False, tasks='EgoNCE_MLM_ITM'):
    sd = make_state_dict(cfg, wseed, tasks)
    if requires_grad:
        for k, v in sd.items():
            if v.is_floating_point():
                v.requires_grad_(True)
    data, noun, verb = make_batch(cfg, B, L, bseed)
    return sd, data, noun, verb, O.make_cfg(**cfg.as_dict())


def rel_err(a, b):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64) if not torch.is_tensor(a) else a.detach().double().cpu()
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64) if not torch.is_tensor(b) else b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
