"""CPU oracle for the optimiser step -- TEST INFRASTRUCTURE (see oracle/ref_model.py header).

Restates transformers==4.30.0 ``AdamW.step`` and the cosine / polynomial warm-up schedules that the reference uses through
set_optim_schedule.py:8-13,108,114-127.  transformers 4.30.0 is a third-party dependency that is NOT in /root/reference and
is not installed here (the installed 5.x removed ``AdamW``), so this part of the oracle is pinned differently:
  * the parameter GROUPING is the reference's own code: oracle/gen_golden_optim.py imports set_optim_schedule.py with a
    recording stub in place of the missing AdamW and stores the six name lists (tests/golden/optim_groups.json);
  * the update arithmetic below is the published algorithm of that release -- no reference test or fixture exists for it ("parity
    unpinned" against the absent dependency itself); it is pinned instead against torch.optim.Adam (the identical update at
    weight_decay = 0, eps = 0) and one hand-computed decay-after-update vector
    (tests/test_optimizer.py::test_oracle_adamw_is_pinned_to_torch_adam_and_a_hand_computed_vector); the HIP kernel is checked against
    this restatement bit-for-bit-close in fp32.
"""
import math

import torch


def adamw_step(p, g, m, v, step, lr, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.0, correct_bias=True):
    """one HF-4.30 AdamW update of a single tensor (in place); returns nothing"""
    b1, b2 = betas
    m.mul_(b1).add_(g, alpha=1.0 - b1)
    v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
    denom = v.sqrt().add_(eps)
    step_size = lr
    if correct_bias:
        step_size = step_size * math.sqrt(1.0 - b2 ** step) / (1.0 - b1 ** step)
    p.addcdiv_(m, denom, value=-step_size)
    if weight_decay > 0.0:
        p.add_(p, alpha=-lr * weight_decay)


def cosine_with_warmup(step, warmup, total, num_cycles=0.5):
    if step < warmup:
        return float(step) / float(max(1, warmup))
    progress = float(step - warmup) / float(max(1, total - warmup))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))
