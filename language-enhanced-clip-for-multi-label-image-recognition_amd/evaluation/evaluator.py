"""mAP over the label set and the multi-label evaluator plug-in.

The metric is the reference's definition (dassl/evaluation/evaluator.py:137-175): per class, sort scores
descending with ``argsort()[::-1]``, sum precision@i over the positives, divide by (n_pos + 1e-8); mean over
classes x 100.  It is fp64 numpy on the host in the reference too (O(C N log N), off the device path).
Unlike the reference's ``MLClassification.evaluate`` - which returns a hard-coded ``mAP_score = 0`` because the
competition test set is unlabeled (:219, :232) - this evaluator computes the metric when labels are present.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from ..registry import EVALUATOR_REGISTRY


def average_precision(output: np.ndarray, target: np.ndarray) -> float:
    order = np.argsort(output)[::-1]
    positives = np.asarray(target)[order] == 1
    hits = np.cumsum(positives, dtype=np.float64)
    n_pos = hits[-1] if len(hits) else 0.0
    prec = hits / np.arange(1, len(hits) + 1, dtype=np.float64)
    return float(prec[positives].sum() / (n_pos + 1e-8))


def mAP(targs: np.ndarray, preds: np.ndarray) -> float:
    if np.size(preds) == 0:
        return 0
    ap = [average_precision(preds[:, k], targs[:, k]) for k in range(preds.shape[1])]
    return 100.0 * float(np.mean(ap))


@EVALUATOR_REGISTRY.register()
class MLClassification:
    """``process(mo, gt)`` accumulates scores/targets, ``evaluate()`` -> OrderedDict(mAP=...) (reference :178-233)."""

    def __init__(self, cfg=None, lab2cname=None, **kwargs):
        self.cfg = cfg
        self._lab2cname = lab2cname
        self.reset()

    def reset(self):
        self._scores, self._targets, self._scores_aux = [], [], []

    def _act(self, mo):
        act = getattr(getattr(self.cfg, "TEST", None), "EVALUATOR_ACT", "default") if self.cfg is not None else "default"
        mo = torch.as_tensor(mo).float()
        if act == "softmax":
            mo = torch.softmax(mo, dim=1)
        elif act == "sigmoid":
            mo = torch.sigmoid(mo)
        return mo.cpu().numpy()

    def process(self, mo, gt, mo_aux=None):
        """Reference :197-205: ``mo`` global scores, ``mo_aux`` the local-branch scores (or None)."""
        self._scores.append(self._act(mo))
        self._targets.append(torch.as_tensor(gt).cpu().numpy())
        if mo_aux is not None:
            self._scores_aux.append(self._act(mo_aux))

    def evaluate(self):
        scores = np.concatenate(self._scores, axis=0)
        targets = np.concatenate(self._targets, axis=0)
        if len(self._scores_aux) > 0:
            # reference :213-218: preds * GL_merge_rate + preds_aux * (1 - GL_merge_rate)
            rate = 0.5
            if self.cfg is not None:
                rate = float(getattr(getattr(getattr(self.cfg, "TRAINER", None), "Caption", None), "GL_merge_rate", 0.5))
            scores = scores * rate + np.concatenate(self._scores_aux, axis=0) * (1 - rate)
        self.merged_scores = scores
        score = mAP(targets, scores)
        print(f"=> result\n* total: {len(scores):,}\n* mAP: {score:.4f}")
        return OrderedDict(mAP=score)
