"""Host mirror of the reference's ``FeatureExtractor`` (``src/tracking/feature_extractor.py:14-88``).

Only what the detect path needs: L2 normalisation and ROI pooling on a DETR encoder map.  The detector itself pools
on the device (``opd_detr_roi_features``); this class keeps the attribute ``detector.feature_extractor`` and its two
methods available to callers that use them directly (e.g. ``src/pipeline/phases/tracking.py:195-207``)."""

from __future__ import annotations

import numpy as np


class FeatureExtractor:
    def normalize_features(self, features: np.ndarray) -> np.ndarray:
        if features.size == 0:
            return features
        norms = np.linalg.norm(features, axis=1, keepdims=True)
        return features / (norms + 1e-8)

    def extract_roi_features(self, encoder_features: np.ndarray, bboxes, image_shape) -> np.ndarray:
        if encoder_features.ndim != 3:
            raise ValueError(f"Expected 3D encoder features, got {encoder_features.ndim}D")
        h, w, feature_dim = encoder_features.shape
        img_h, img_w = image_shape
        rois = []
        for (x, y, width, height) in bboxes:
            x_min = int((x / img_w) * w)
            y_min = int((y / img_h) * h)
            x_max = int(((x + width) / img_w) * w)
            y_max = int(((y + height) / img_h) * h)
            x_min = max(0, min(x_min, w - 1))
            y_min = max(0, min(y_min, h - 1))
            x_max = max(x_min + 1, min(x_max, w))
            y_max = max(y_min + 1, min(y_max, h))
            rois.append(encoder_features[y_min:y_max, x_min:x_max, :].mean(axis=(0, 1)))
        if not rois:
            return np.array([]).reshape(0, feature_dim)
        return self.normalize_features(np.array(rois))
