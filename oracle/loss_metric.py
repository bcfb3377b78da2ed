"""Oracle: weighted per-pixel cross-entropy, argmax and mean-IoU.  TEST INFRASTRUCTURE ONLY.

Reference:
  fp_loss         src/deepCam/utils/losses.py:28-52
  argmax          torch.max(outputs, 1)[1]   train_hdf5_ddp.py:376,406,458
  compute_score   src/deepCam/utils/utils.py:32-60
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np
import torch


def class_weights(loss_pow: float = -0.125) -> list:
    """train_hdf5_ddp.py:204-206."""
    return [0.986267818390377 ** loss_pow, 0.0004578708870701058 ** loss_pow, 0.01327431072255291 ** loss_pow]


def weighted_ce_map(logit: torch.Tensor, target: torch.Tensor, weight: Sequence[float]) -> torch.Tensor:
    """Per-pixel w[y] * (logsumexp(logit) - logit[y]), shape [B,H,W].

    losses.py:35-36 builds nn.CrossEntropyLoss(weight, reduction='none').  The two
    "false positive" factors at :41-46 multiply by (eq & ne) == 0, i.e. they are identities.
    """
    w = torch.tensor(np.array(weight), dtype=torch.float64).float().to(logit.device)   # losses.py:35 goes through numpy float64 -> float32
    t = target.squeeze(1) if target.dim() == 4 else target
    t = t.long()
    lse = torch.logsumexp(logit, 1)
    picked = torch.gather(logit, 1, t.unsqueeze(1)).squeeze(1)
    return w[t] * (lse - picked)


def fp_loss(logit: torch.Tensor, target: torch.Tensor, weight: Sequence[float], fpw_1: float = 0, fpw_2: float = 0) -> torch.Tensor:
    """Plain mean over B*H*W of the weighted map (losses.py:50) -- NOT the weight-normalised mean."""
    return weighted_ce_map(logit, target, weight).mean()


def fp_loss_grad(logit: torch.Tensor, target: torch.Tensor, weight: Sequence[float]) -> torch.Tensor:
    """d fp_loss / d logit = w[y] * (softmax - onehot) / (B*H*W)."""
    w = torch.tensor(np.array(weight), dtype=torch.float64).float()
    t = (target.squeeze(1) if target.dim() == 4 else target).long()
    p = torch.softmax(logit, 1)
    onehot = torch.zeros_like(p).scatter_(1, t.unsqueeze(1), 1.0)
    n = t.numel()
    return (p - onehot) * (w[t] / n).unsqueeze(1)


def argmax_first(logit: torch.Tensor) -> np.ndarray:
    """First index of the maximum over the class axis (torch.max tie-break), int64 [B,H,W].

    Written with explicit strict comparisons so that the tie-break is part of the oracle,
    not an accident of a library call.
    """
    x = logit.detach().cpu().numpy()
    best = x[:, 0]
    idx = np.zeros(best.shape, dtype=np.int64)
    for k in range(1, x.shape[1]):
        better = x[:, k] > best
        idx = np.where(better, k, idx)
        best = np.where(better, x[:, k], best)
    return idx


def confusion_counts(pred: np.ndarray, gt: np.ndarray, num_classes: int = 3) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(tp, fp, fn) int64 vectors as utils.py:43-51 defines them."""
    pred = np.asarray(pred).astype(np.int64).ravel()
    gt = np.asarray(gt).astype(np.int64).ravel()
    tp = np.zeros(num_classes, np.int64)
    fp = np.zeros(num_classes, np.int64)
    fn = np.zeros(num_classes, np.int64)
    eq = pred == gt
    for j in range(num_classes):
        tp[j] = np.count_nonzero(eq & (gt == j))
        fp[j] = np.count_nonzero(~eq & (pred == j))
        fn[j] = np.count_nonzero(~eq & (gt == j))
    return tp, fp, fn


def iou_from_counts(tp, fp, fn) -> float:
    """mean_j tp/(tp+fp+fn), with IoU_j = 1 when the union is empty (utils.py:53-60); fp32 divide."""
    ious = []
    for a, b, c in zip(tp, fp, fn):
        union = int(a) + int(b) + int(c)
        ious.append(np.float32(1.0) if union == 0 else np.float32(a) / np.float32(union))
    return float(np.float32(sum(ious, np.float32(0.0))) / np.float32(len(ious)))


def compute_score(prediction, gt, num_classes: int = 3) -> float:
    pred = prediction.detach().cpu().numpy() if torch.is_tensor(prediction) else prediction
    g = gt.detach().cpu().numpy() if torch.is_tensor(gt) else gt
    return iou_from_counts(*confusion_counts(pred, g, num_classes))
