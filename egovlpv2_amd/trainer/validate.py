"""EgoMCQ validation scoring (reference trainer/trainer_egoclip.py:_valid_epoch, :216-246): for b1 questions with b2 = 5
candidate clips each, the dual-encoder cosine score (VTC) and the fused-encoder match probability (VTM), and their sum
(the `ensemble` the reference reports).  Tokenisation and the cross-rank gathers stay with the caller's trainer."""
import torch
import torch.nn.functional as F

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
