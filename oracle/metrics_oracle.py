"""CPU oracle (numpy, fp64): label/metric tail of the scoring path.

TEST INFRASTRUCTURE ONLY (see ``oracle/clip_oracle.py`` header for who may import it).
Pinned by ``tests/golden/metrics_kat.npz``, written by ``oracle/make_golden.py`` from the
reference's own ``dassl/evaluation/evaluator.py`` and ``trainers/utils.py`` run in the
build container.
"""
from __future__ import annotations

import numpy as np


def average_precision(output: np.ndarray, target: np.ndarray) -> float:
    """dassl/evaluation/evaluator.py:137-154.  Scores sorted descending via
    ``argsort()[::-1]`` (so ties come out in *reverse* stable order); precision@i summed
    over positives, divided by (n_pos + 1e-8)."""
    order = np.argsort(output)[::-1]
    hit = target[order] == 1
    rank = np.arange(1, len(output) + 1, dtype=np.float64)
    cum = np.cumsum(hit).astype(np.float64)
    n_pos = cum[-1]
    return float(np.sum(np.where(hit, cum / rank, 0.0)) / (n_pos + 1e-8))


def mAP(targs: np.ndarray, preds: np.ndarray) -> float:
    """dassl/evaluation/evaluator.py:157-175 - 100 x mean over classes of AP."""
    if np.size(preds) == 0:
        return 0
    return 100.0 * float(np.mean([average_precision(preds[:, k], targs[:, k]) for k in range(preds.shape[1])]))


def ranking_loss(y_pred: np.ndarray, y_true: np.ndarray, scale_: float = 2.0, margin_: float = 1.0) -> float:
    """trainers/utils.py:85-93: mean_b sum_{i,j} max(0, m - s*p[b,j] + s*p[b,i]) * y[b,j] * (1-y[b,i])."""
    p = np.asarray(y_pred, dtype=np.float64) * scale_
    y = np.asarray(y_true, dtype=np.float64)
    tmp = margin_ - p[:, None, :] + p[:, :, None]
    loss = np.maximum(tmp, 0.0) * y[:, None, :] * (1.0 - y[:, :, None])
    return float(loss.sum(axis=-1).sum(axis=-1).mean())


def norm_logits_bce(pred: np.ndarray, targets: np.ndarray) -> float:
    """trainers/utils.py:21-23: BCEWithLogitsLoss, mean reduction."""
    x = np.asarray(pred, dtype=np.float64)
    t = np.asarray(targets, dtype=np.float64)
    return float(np.mean(np.maximum(x, 0.0) - x * t + np.log1p(np.exp(-np.abs(x)))))


def window_aggregate(global_logits: np.ndarray, window_logits: np.ndarray, threshold: float = 0.3, weight: float = 1.4) -> np.ndarray:
    """trainers/Caption_distill_double.py:654-660: alpha = max over windows, beta = min over windows,
    gamma = alpha > threshold, s_ag = gamma*alpha + (1-gamma)*beta, final = 1.4*s_ag + global.  Pinned: make_golden.py
    executes those reference lines as a source slice on seeded scores (tests/golden/postprocess.npz)."""
    alpha, beta = window_logits.max(axis=1), window_logits.min(axis=1)
    gamma = (alpha > threshold).astype(window_logits.dtype)
    return weight * (gamma * alpha + (1 - gamma) * beta) + global_logits


def cooccurrence_adjust(p: np.ndarray, adj: np.ndarray, nums: np.ndarray, weight: float = 0.5) -> np.ndarray:
    """Caption_distill_double.py:614-618 + 632-636: M = adj / nums[:, None]; M /= M.sum(-1)[:, None]; p + weight * (p @ M).
    Pinned the same way, on the reference's real freq_stats.pkl (values stored in the fixture)."""
    m = adj / nums[:, None]
    m = m / m.sum(-1)[:, None]
    return p + weight * (p @ m)


def merge_global_local(preds: np.ndarray, preds_aux: np.ndarray, rate: float = 0.5) -> np.ndarray:
    """dassl/evaluation/evaluator.py:213-218 (activate_func 'default_merge_aux'): preds * rate + preds_aux * (1 - rate)
    with rate = cfg.TRAINER.Caption.GL_merge_rate."""
    return np.asarray(preds) * rate + np.asarray(preds_aux) * (1 - rate)
