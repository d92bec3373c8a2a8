"""Losses at the end of the prompt-tuning step (reference trainers/utils.py:21-23, 85-93); thin host-side
PyTorch over [B, 80] logits - not part of the kernel path."""
import torch


def norm_logits_BCEloss(pred, targets):
    return torch.nn.functional.binary_cross_entropy_with_logits(pred, targets)


def ranking_loss(y_pred, y_true, scale_=2.0, margin_=1):
    y_pred = y_pred * scale_
    y_true_ = y_true.float()
    tmp = margin_ - y_pred[:, None, :] + y_pred[:, :, None]
    loss = torch.clamp(tmp, min=0) * y_true_[:, None, :] * (1 - y_true_[:, :, None])
    return loss.sum(dim=-1).sum(dim=-1).mean()
