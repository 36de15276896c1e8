"""CPU restatement of the tracker's cost matrix — TEST INFRASTRUCTURE ONLY, never imported by the product.

Follows ``src/tracking/similarity.py`` of the reference: ``cosine_similarity`` (:42-60: fp32 dot of L2-normalised features,
clipped to [-1, 1]), ``iou`` (:75-118: xywh boxes, 0 when the intersection is empty or the union is not positive, clipped to
[0, 1]), ``compute_similarity`` (:133-172: weighted sum of the available terms divided by the sum of the weights used,
clipped to [0, 1]; a missing feature vector drops the appearance term), ``compute_similarity_matrix`` / ``compute_distance_matrix``
(:190-220: float32 matrices, distance = 1 - similarity).  Pinned to the reference's own class on seeded data
(``tests/golden/similarity.npz``, made by ``tools/gen_golden.py``).
"""

from __future__ import annotations

from typing import Optional

import numpy as np


def iou_xywh(b1, b2) -> float:
    x1, y1, w1, h1 = (float(v) for v in b1)
    x2, y2, w2, h2 = (float(v) for v in b2)
    ix0, iy0 = max(x1, x2), max(y1, y2)
    ix1, iy1 = min(x1 + w1, x2 + w2), min(y1 + h1, y2 + h2)
    if ix1 <= ix0 or iy1 <= iy0:
        return 0.0
    inter = (ix1 - ix0) * (iy1 - iy0)
    union = w1 * h1 + w2 * h2 - inter
    if union <= 0:
        return 0.0
    return float(np.clip(inter / union, 0.0, 1.0))


def similarity_matrix(f1: Optional[np.ndarray], b1: np.ndarray, has1: Optional[np.ndarray], f2: Optional[np.ndarray], b2: np.ndarray,
                      has2: Optional[np.ndarray], appearance_weight: float = 0.7, motion_weight: float = 0.3) -> np.ndarray:
    n1, n2 = len(b1), len(b2)
    out = np.zeros((n1, n2), np.float32)
    for i in range(n1):
        for j in range(n2):
            score, total = 0.0, 0.0
            ok = f1 is not None and f2 is not None and (has1 is None or has1[i]) and (has2 is None or has2[j])
            if ok:
                score += appearance_weight * float(np.clip(np.dot(f1[i], f2[j]), -1.0, 1.0))
                total += appearance_weight
            score += motion_weight * iou_xywh(b1[i], b2[j])
            total += motion_weight
            out[i, j] = float(np.clip(score / total if total > 0 else 0.0, 0.0, 1.0))
    return out
