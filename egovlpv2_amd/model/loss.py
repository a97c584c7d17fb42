"""Losses of the model API: EgoNCE (reference model/loss.py:33-61) on the fused HIP kernel; the ranking losses of the fine-tune
variant (loss.py:13-31, :65-143) as a few tensor ops on the (n, n) similarity matrix (n = global batch: host-scale work)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hipops


class EgoNCE(nn.Module):
    def __init__(self, temperature=0.05, noun=True, verb=True):
        super().__init__()
        self.noun = noun
        self.verb = verb
        self.temperature = temperature

    def forward(self, x, mask_v, mask_n):
        """x: (n, n) text->video cosine similarities; mask_v / mask_n: verb / noun cosine matrices.
        Returns (loss, mask_bool, temperature) like the reference."""
        loss, mask_bool = hipops.egonce(x.float(), mask_v.float(), mask_n.float(), self.temperature, self.noun, self.verb)
        return loss, mask_bool, self.temperature


class NormSoftmaxLoss(nn.Module):
    """loss.py:13-31: symmetric InfoNCE on the diagonal; returns (loss, temperature)"""

    def __init__(self, temperature=0.05):
        super().__init__()
        self.temperature = temperature

    def forward(self, x):
        li = torch.diag(F.log_softmax(x / self.temperature, dim=1)).mean()
        lj = torch.diag(F.log_softmax(x.t() / self.temperature, dim=1)).mean()
        return -li - lj, self.temperature


def _max_margin(x, m, fix_norm):
    """mean of relu(m_i - (x_ii - x_ij)) and relu(m_i - (x_ii - x_ji)) over all (i, j), or over i != j with fix_norm
    (loss.py:73-99 builds the same 2 n^2 (2 n (n - 1)) terms with view/cat/index_select)"""
    n = x.shape[0]
    d = torch.diag(x).unsqueeze(1)
    rows = F.relu(m - (d - x))
    cols = F.relu(m - (d - x.t()))
    if fix_norm:
        off = ~torch.eye(n, dtype=torch.bool, device=x.device)
        return torch.cat([rows[off], cols[off]]).mean()
    return torch.cat([rows.reshape(-1), cols.reshape(-1)]).mean()


class MaxMarginRankingLoss(nn.Module):
    """loss.py:65-99"""

    def __init__(self, margin=0.2, fix_norm=True):
        super().__init__()
        self.margin = margin
        self.fix_norm = fix_norm

    def forward(self, x, weight=None):
        return _max_margin(x, self.margin, self.fix_norm)


class AdaptiveMaxMarginRankingLoss(nn.Module):
    """loss.py:102-143: the margin of row i is weight_i * margin"""

    def __init__(self, margin=0.4, fix_norm=True):
        super().__init__()
        self.margin = margin
        self.fix_norm = fix_norm

    def forward(self, x, weight=None):
        return _max_margin(x, self.margin * weight.to(x.dtype).unsqueeze(1), self.fix_norm)
