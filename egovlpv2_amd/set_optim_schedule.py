"""Optimiser + LR schedule of the pre-training step (reference set_optim_schedule.py:16-129) on MI355X.

``set_schedule(model, config, config_yaml, max_steps, warmup_steps) -> (optimizer, scheduler)`` keeps the reference
signature.  Parameters are split into the reference's six groups by SUBSTRING match on their names
({decay, no-decay} x {backbone, head * lr_mult_head, cross-modal * lr_mult_cross_modal}; quirks preserved: ``norm3.weight``
and ``norm_i2t_i.weight`` are decayed because only ``norm.``/``norm1.``/``norm2.`` are listed, a name matching both a head
and a cross-modal pattern lands in no group).  The update is transformers==4.30.0 ``AdamW`` (the package is not installed
here and 5.x removed the class; algorithm restated from its published source) as ONE fused multi-tensor HIP kernel per
group (csrc/egv_optim.hip); schedules are the HF cosine / polynomial warm-up lambdas.
"""
from __future__ import annotations

import math

import torch

NO_DECAY = ["bias", "LayerNorm.bias", "LayerNorm.weight", "norm.bias", "norm.weight", "norm1.bias", "norm1.weight",
            "norm2.bias", "norm2.weight"]
HEAD_NAMES = ["mlm_score", "itm_score", "txt_proj", "vid_proj"]
CROSS_MODAL_NAMES = ["cross_modal", "i2t", "t2i"]


def group_parameters(named_parameters, lr, wd, lr_mult_head, lr_mult_cross_modal):
    """set_optim_schedule.py:37-103: six groups in the reference's order; returns a list of dicts with 'names' added."""
    named = list(named_parameters)

    def pick(nd, head, cross):
        return [(n, p) for n, p in named
                if any(x in n for x in NO_DECAY) == nd and any(x in n for x in HEAD_NAMES) == head
                and any(x in n for x in CROSS_MODAL_NAMES) == cross]
    spec = [(False, False, False, wd, lr), (True, False, False, 0.0, lr),
            (False, True, False, wd, lr * lr_mult_head), (True, True, False, 0.0, lr * lr_mult_head),
            (False, False, True, wd, lr * lr_mult_cross_modal), (True, False, True, 0.0, lr * lr_mult_cross_modal)]
    groups = []
    for nd, head, cross, gwd, glr in spec:
        sel = pick(nd, head, cross)
        groups.append({"params": [p for _, p in sel], "names": [n for n, _ in sel], "weight_decay": gwd, "lr": glr})
    return groups


def cosine_lambda(warmup_steps, max_steps, num_cycles=0.5):
    """transformers.get_cosine_schedule_with_warmup"""
    def f(step):
        if step < warmup_steps:
            return float(step) / float(max(1, warmup_steps))
        progress = float(step - warmup_steps) / float(max(1, max_steps - warmup_steps))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))
    return f


def polynomial_lambda(warmup_steps, max_steps, lr_init, lr_end, power):
    """transformers.get_polynomial_decay_schedule_with_warmup"""
    def f(step):
        if step < warmup_steps:
            return float(step) / float(max(1, warmup_steps))
        if step > max_steps:
            return lr_end / lr_init
        remaining = 1 - (step - warmup_steps) / (max_steps - warmup_steps)
        return ((lr_init - lr_end) * remaining ** power + lr_end) / lr_init
    return f


class FusedAdamW(torch.optim.Optimizer):
    """HF-4.30 AdamW semantics, one fused HIP launch per parameter group.  State keys match HF ('step', 'exp_avg',
    'exp_avg_sq') so optimiser checkpoints round-trip."""
    CHUNK = 16384

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias)
        super().__init__(params, defaults)
        self._tables = {}

    def _table(self, gi, plist):
        """device table (one 32-byte record per tensor) + chunk prefix for the tensors of one launch; rebuilt only when a
        pointer changed"""
        import numpy as np
        arr = np.zeros(len(plist), dtype=[('p', '<u8'), ('g', '<u8'), ('m', '<u8'), ('v', '<u8'), ('n', '<i4'), ('pad', '<i4')])
        arr['p'] = [p.data_ptr() for p in plist]
        arr['g'] = [p.grad.data_ptr() for p in plist]
        arr['m'] = [self.state[p]['exp_avg'].data_ptr() for p in plist]
        arr['v'] = [self.state[p]['exp_avg_sq'].data_ptr() for p in plist]
        arr['n'] = [p.numel() for p in plist]
        ent = self._tables.get(gi)
        if ent is not None and ent[0].shape == arr.shape and (ent[0] == arr).all():
            return ent[1], ent[2], len(plist), ent[3]
        for p in plist:
            if not (p.is_contiguous() and p.grad.is_contiguous() and p.dtype == torch.float32 and p.grad.dtype == torch.float32):
                raise RuntimeError("FusedAdamW needs contiguous fp32 parameters and gradients")
        chunks = (arr['n'].astype(np.int64) + self.CHUNK - 1) // self.CHUNK
        prefix = np.zeros(len(plist) + 1, dtype=np.int32)
        prefix[1:] = np.cumsum(chunks)
        dev = plist[0].device
        pin = torch.from_numpy(arr.view(np.uint8).copy()).pin_memory()
        table = pin.to(dev, non_blocking=True)
        pre = torch.from_numpy(prefix).pin_memory().to(dev, non_blocking=True)
        self._tables[gi] = (arr, table, pre, int(prefix[-1]), pin)
        return table, pre, len(plist), int(prefix[-1])

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        from . import hipops as ops
        from ._lib import lib, check
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            # 'step' is per parameter, as in HF AdamW: a tensor that had no gradient in some steps (task subsets, unused
            # fusion layers, a resumed optimiser state) keeps its own bias correction.  Tensors are launched together per
            # distinct step value -- one launch per group in the usual case where they all agree.
            by_step = {}
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if 'exp_avg' not in st:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                by_step.setdefault(int(st['step']), []).append(p)
            b1, b2 = group['betas']
            lr = group['lr']
            for si, (t0, plist) in enumerate(sorted(by_step.items())):
                table, prefix, nt, nchunks = self._table((gi, si), plist)
                t = t0 + 1
                for p in plist:
                    self.state[p]['step'] = t
                step_size = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t) if group['correct_bias'] else lr
                check(lib.egv_adamw_step(table.data_ptr(), prefix.data_ptr(), nt, nchunks, float(lr), float(step_size), float(b1), float(b2),
                                         float(group['eps']), float(group['weight_decay']), float(grad_scale),
                                         torch.cuda.current_stream().cuda_stream), 'egv_adamw_step')
        ops.invalidate_weight_cache()          # the fp32 masters changed: bf16 compute copies are stale
        return loss


def set_schedule(model, config, config_yaml, max_steps, warmup_steps):
    """reference set_optim_schedule.py:16-129"""
    a = config["optimizer"]["args"]
    lr, wd = a["lr"], a["weight_decay"]
    groups = group_parameters(model.named_parameters(), lr, wd, a["lr_mult_head"], a["lr_mult_cross_modal"])
    optim_type = config["optimizer"]["type"]
    if optim_type != "AdamW":
        raise NotImplementedError(f"optimizer type {optim_type!r}: only the pre-training config's AdamW is implemented on HIP")
    pg = [{k: v for k, v in g.items() if k != 'names'} for g in groups]
    optimizer = FusedAdamW(pg, lr=lr, eps=1e-8, betas=(0.9, 0.98))
    decay_power = config_yaml["decay_power"]
    if decay_power == "cosine":
        lam = cosine_lambda(warmup_steps, max_steps)
    else:
        lam = polynomial_lambda(warmup_steps, max_steps, lr, config_yaml["end_lr"], decay_power)
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lam)
    return optimizer, scheduler
