"""EgoMCQ validation scoring (reference trainer/trainer_egoclip.py:_valid_epoch, :216-246): for b1 questions with b2 = 5
candidate clips each, the dual-encoder cosine score (VTC) and the fused-encoder match probability (VTM), and their sum
(the `ensemble` the reference reports), and the cross-rank accumulation of one validation pass (:250-291: all_gather of ground
truth, scores and question types per batch, concatenation, the two accuracy metrics).  Tokenisation stays with the caller."""
import torch
import torch.distributed as dist
import torch.nn.functional as F

from ..model.metric import egomcq_accuracy_metrics_ensemble, egomcq_accuracy_metrics_vtm
from ..model.model import sim_matrix_batch_val


@torch.no_grad()
def egomcq_scores(model, data):
    """data['video']: (b1, b2, F, 3, H, W); data['text']: {'input_ids', 'attention_mask'} of b1 questions.
    Returns dict(vtc (b1, b2), vtm (b1, b2), ensemble (b1, b2))."""
    b1, b2 = data['video'].shape[:2]
    video = data['video'].reshape(b1 * b2, *data['video'].shape[2:])
    text = data['text']
    ret = model.infer({'video': video, 'text': text}, return_embeds=True, task_names='EgoNCE', ret={})
    rep = {'input_ids': torch.repeat_interleave(text['input_ids'], b2, dim=0),
           'attention_mask': torch.repeat_interleave(text['attention_mask'], b2, dim=0)}
    ret = model.infer({'video': video, 'text': rep}, return_embeds=True, task_names='ITM', ret=ret)
    te = ret['text_embeds'].float().reshape(b1, 1, -1)
    ve = ret['video_embeds'].float().reshape(b1, b2, -1)
    vtc = sim_matrix_batch_val(te, ve).squeeze(1)
    vtm = F.softmax(ret['cross_attn_itm_logits'].float(), dim=1)[:, 1:].t().reshape(1, b1, b2)[0].contiguous()
    return {'vtc': vtc, 'vtm': vtm, 'ensemble': vtc + vtm}


def _gather_cat(t: torch.Tensor, group=None) -> torch.Tensor:
    """every rank's tensor of this batch, concatenated along dim 0 in rank order (trainer_egoclip.py:250-267); a process that runs
    without a process group returns its own tensor"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, t.contiguous(), group=group)
    return torch.cat(parts, dim=0)


class EgoMCQAccumulator:
    """One validation pass over an EgoMCQ loader (trainer_egoclip.py:_valid_epoch, :216-291): per batch the scores of
    `egomcq_scores`, the index of the correct clip and the question type (1 inter-video, 2 intra-video) are gathered from all
    ranks and kept on the host; `metrics()` concatenates them and evaluates the reference's two metrics
    (model/metric.py:225-259).  Like the reference, every rank ends up with the same arrays and the same numbers.

        acc = EgoMCQAccumulator()
        for data in loader:                      # the reference's loader: one question per rank and step
            acc.add(egomcq_scores(model, data), data['correct'], data['type'])
        res = acc.metrics()                      # {'egomcq_accuracy_metrics_ensemble': {...}, 'egomcq_accuracy_metrics_vtm': {...}}
    """

    def __init__(self, group=None):
        self.group = group
        self.gt, self.ensemble, self.vtm, self.types = [], [], [], []

    def add(self, scores: dict, correct: torch.Tensor, qtype: torch.Tensor):
        dev = scores['vtm'].device
        b1 = scores['vtm'].shape[0]
        gt = torch.as_tensor(correct).to(dev).reshape(-1)[:b1]
        ty = torch.as_tensor(qtype).to(dev).reshape(-1)[:b1]
        vtc = _gather_cat(scores['vtc'].float(), self.group)
        vtm = _gather_cat(scores['vtm'].float(), self.group)
        self.gt.append(_gather_cat(gt, self.group).cpu())
        self.ensemble.append((vtc + vtm).cpu())                    # the sum is formed after the gathers, as at :262
        self.vtm.append(vtm.cpu())
        self.types.append(_gather_cat(ty, self.group).cpu())

    def arrays(self):
        return {'gt': torch.cat(self.gt), 'ensemble': torch.cat(self.ensemble), 'vtm': torch.cat(self.vtm), 'type': torch.cat(self.types)}

    def metrics(self) -> dict:
        a = self.arrays()
        return {'egomcq_accuracy_metrics_ensemble': egomcq_accuracy_metrics_ensemble(a['ensemble'], a['gt'], a['type']),
                'egomcq_accuracy_metrics_vtm': egomcq_accuracy_metrics_vtm(a['vtm'], a['gt'], a['type'])}
