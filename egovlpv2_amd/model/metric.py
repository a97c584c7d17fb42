"""EgoMCQ accuracy (reference model/metric.py:225-259): argmax over the 5 candidate clips per question, reported per
question type.  The two reference functions (`..._ensemble`, `..._vtm`) are the same computation on different scores; the
per-sample Python loop with `.item()` host syncs is replaced by one vectorised comparison."""
import torch

GROUPS = ["Inter-video", "Intra-video"]


def _egomcq_accuracy(preds: torch.Tensor, labels: torch.Tensor, types: torch.Tensor) -> dict:
    preds, labels, types = preds.detach().cpu(), labels.detach().cpu().reshape(-1), types.detach().cpu().reshape(-1)
    hit = preds.argmax(dim=-1).reshape(-1) == labels
    metrics = {}
    for type_i, group_i in zip(torch.unique(types), GROUPS):          # sorted unique values, zipped like the reference
        sel = types == type_i
        metrics[group_i] = hit[sel].float().mean().item() * 100
    return metrics


def egomcq_accuracy_metrics_ensemble(preds, labels, types):
    return _egomcq_accuracy(preds, labels, types)


def egomcq_accuracy_metrics_vtm(preds, labels, types):
    return _egomcq_accuracy(preds, labels, types)
