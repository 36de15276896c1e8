"""``detector.feature_extractor``: the attribute the reference's callers expect (``src/pipeline/phases/tracking.py:195-207``).

The detector pools appearance features on the device (``opd_detr_roi_features``, ``kernels_misc.hip::roi_features_kernel``).
This host class serves callers that hold an encoder map in numpy already.  Contract of the reference class
(``src/tracking/feature_extractor.py:39-88``): boxes are (x, y, w, h) in image pixels, mapped to the (h, w, C) map by
truncation, clamped to at least one cell, mean-pooled and L2-normalised with ``+1e-8``.  All boxes are pooled at once from
one summed-area table of the map (float64), not cell by cell."""

from __future__ import annotations

import numpy as np

_EPS = 1e-8


def roi_cells(bboxes: np.ndarray, map_hw, image_hw) -> np.ndarray:
    """(N, 4) xywh pixel boxes -> (N, 4) int cell ranges [x0, y0, x1, y1) on the feature map (truncate, then clamp so that
    every range holds at least one cell)."""
    h, w = map_hw
    img_h, img_w = image_hw
    b = np.asarray(bboxes, dtype=np.float64).reshape(-1, 4)
    lo = np.trunc(b[:, :2] / (img_w, img_h) * (w, h)).astype(np.int64)
    hi = np.trunc((b[:, :2] + b[:, 2:]) / (img_w, img_h) * (w, h)).astype(np.int64)
    lo = np.clip(lo, 0, (w - 1, h - 1))
    hi = np.maximum(lo + 1, np.minimum(hi, (w, h)))
    return np.concatenate([lo, hi], axis=1)


class FeatureExtractor:
    def normalize_features(self, features: np.ndarray) -> np.ndarray:
        if features.size == 0:
            return features
        return features / (np.sqrt((features * features).sum(axis=1, keepdims=True)) + _EPS)

    def extract_roi_features(self, encoder_features: np.ndarray, bboxes, image_shape) -> np.ndarray:
        if encoder_features.ndim != 3:
            raise ValueError(f"Expected 3D encoder features, got {encoder_features.ndim}D")
        h, w, c = encoder_features.shape
        if len(bboxes) == 0:
            return np.array([]).reshape(0, c)
        cells = roi_cells(np.asarray(bboxes), (h, w), image_shape)
        sat = np.zeros((h + 1, w + 1, c), dtype=np.float64)
        sat[1:, 1:] = encoder_features.astype(np.float64).cumsum(axis=0).cumsum(axis=1)
        x0, y0, x1, y1 = cells.T
        total = sat[y1, x1] - sat[y0, x1] - sat[y1, x0] + sat[y0, x0]
        pooled = total / ((x1 - x0) * (y1 - y0))[:, None]
        return self.normalize_features(pooled.astype(encoder_features.dtype if encoder_features.dtype.kind == "f" else np.float64))
