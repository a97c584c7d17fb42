"""EgoNCE (reference model/loss.py:33-61) on the fused HIP kernel."""
import torch.nn as nn

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
