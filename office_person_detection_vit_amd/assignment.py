"""Track-to-detection assignment on the host (the step after the device-side cost matrix).

``HungarianAlgorithm.solve`` keeps the contract of the reference's class (``src/tracking/hungarian.py:26-62``): an optimal
assignment of rows to columns of a cost matrix (rows = tracks, columns = detections, ``similarity.SimilarityCalculator.
compute_distance_matrix``), returned as one column index per row with -1 for "unassigned", plus the summed cost.  Infinite
entries mark forbidden pairs: the solver sees them as a large finite cost and a row that ends up on one stays unassigned.
Pinned by ``tests/golden/assignment.json`` (the reference class on seeded matrices, ``tools/gen_golden.py assignment``).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
from scipy.optimize import linear_sum_assignment

FORBIDDEN_COST = 1e9   # what an infinite entry costs inside the solver (the reference's constant)


class HungarianAlgorithm:
    def solve(self, cost_matrix: np.ndarray) -> Tuple[np.ndarray, float]:
        cost_matrix = np.asarray(cost_matrix)
        if cost_matrix.size == 0:
            return np.array([], dtype=np.int32), 0.0
        forbidden = np.isinf(cost_matrix)
        rows, cols = linear_sum_assignment(np.where(forbidden, FORBIDDEN_COST, cost_matrix))
        assignment = np.full(cost_matrix.shape[0], -1, dtype=np.int32)
        total = 0.0
        for r, c in zip(rows, cols):
            if not forbidden[r, c]:
                assignment[r] = c
                total += cost_matrix[r, c]
        return assignment, total


def assign_tracks(distance: np.ndarray, max_distance: float) -> Tuple[np.ndarray, float]:
    """Gate then solve: pairs whose distance exceeds ``max_distance`` are forbidden, the rest is assigned optimally."""
    d = np.asarray(distance, dtype=np.float64)
    return HungarianAlgorithm().solve(np.where(d > max_distance, np.inf, d))
