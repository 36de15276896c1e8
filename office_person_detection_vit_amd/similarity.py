"""Tracker cost matrix on the device: the reference's ``SimilarityCalculator``
(``src/tracking/similarity.py:16-220``) with the same constructor, method names and error behaviour.  The scalar helpers
are the reference's formulas on the host (they are called per pair by ``Tracker``); the two matrix builders run on the
MI355X through ``opd_similarity_matrix`` and fail loudly when the HIP library is missing."""

from __future__ import annotations

import ctypes as C
import logging
from typing import List, Optional, Tuple

import numpy as np

from . import _capi
from .data_models import Detection

logger = logging.getLogger(__name__)


class SimilarityCalculator:
    def __init__(self, appearance_weight: float = 0.7, motion_weight: float = 0.3, device_ordinal: int = 0):
        if abs(appearance_weight + motion_weight - 1.0) > 1e-6:
            raise ValueError(f"appearance_weight ({appearance_weight}) + motion_weight ({motion_weight}) must equal 1.0")
        self.appearance_weight = appearance_weight
        self.motion_weight = motion_weight
        self.device_ordinal = device_ordinal
        logger.info(f"SimilarityCalculator initialized: appearance_weight={appearance_weight}, motion_weight={motion_weight}")

    # ---- scalar helpers (``similarity.py:42-131``) -------------------------------------------------------------------------
    def cosine_similarity(self, feat1: np.ndarray, feat2: np.ndarray) -> float:
        if feat1.shape != feat2.shape:
            raise ValueError(f"Feature shape mismatch: {feat1.shape} vs {feat2.shape}")
        return float(np.clip(np.dot(feat1, feat2), -1.0, 1.0))

    def cosine_distance(self, feat1: np.ndarray, feat2: np.ndarray) -> float:
        return 1.0 - self.cosine_similarity(feat1, feat2)

    def iou(self, bbox1: Tuple[float, float, float, float], bbox2: Tuple[float, float, float, float]) -> float:
        x1, y1, w1, h1 = bbox1
        x2, y2, w2, h2 = bbox2
        ix0, iy0 = max(x1, x2), max(y1, y2)
        ix1, iy1 = min(x1 + w1, x2 + w2), min(y1 + h1, y2 + h2)
        if ix1 <= ix0 or iy1 <= iy0:
            return 0.0
        inter = (ix1 - ix0) * (iy1 - iy0)
        union = w1 * h1 + w2 * h2 - inter
        if union <= 0:
            return 0.0
        return float(np.clip(inter / union, 0.0, 1.0))

    def iou_distance(self, bbox1, bbox2) -> float:
        return 1.0 - self.iou(bbox1, bbox2)

    def compute_similarity(self, det1: Detection, det2: Detection, use_appearance: bool = True, use_motion: bool = True) -> float:
        score, total = 0.0, 0.0
        if use_appearance and det1.features is not None and det2.features is not None:
            score += self.appearance_weight * self.cosine_similarity(det1.features, det2.features)
            total += self.appearance_weight
        elif use_appearance:
            logger.warning("Features not available, skipping appearance similarity")
        if use_motion:
            score += self.motion_weight * self.iou(det1.bbox, det2.bbox)
            total += self.motion_weight
        score = score / total if total > 0 else 0.0
        return float(np.clip(score, 0.0, 1.0))

    def compute_distance(self, det1: Detection, det2: Detection, use_appearance: bool = True, use_motion: bool = True) -> float:
        return 1.0 - self.compute_similarity(det1, det2, use_appearance, use_motion)

    # ---- matrices on the device (``similarity.py:190-220``) ------------------------------------------------------------------
    @staticmethod
    def _pack(dets: List[Detection]):
        boxes = np.ascontiguousarray([d.bbox for d in dets], dtype=np.float32).reshape(len(dets), 4)
        has = np.ascontiguousarray([d.features is not None for d in dets], dtype=np.uint8)
        dim = next((int(np.asarray(d.features).size) for d in dets if d.features is not None), 0)
        feats = None
        if dim:
            feats = np.zeros((len(dets), dim), np.float32)
            for i, d in enumerate(dets):
                if d.features is not None:
                    feats[i] = np.asarray(d.features, dtype=np.float32).reshape(-1)
        return feats, boxes, has, dim

    def _matrix(self, detections1: List[Detection], detections2: List[Detection], as_distance: bool) -> np.ndarray:
        n1, n2 = len(detections1), len(detections2)
        out = np.zeros((n1, n2), np.float32)
        if n1 == 0 or n2 == 0:
            return out
        f1, b1, h1, d1 = self._pack(detections1)
        f2, b2, h2, d2 = self._pack(detections2)
        if d1 and d2 and d1 != d2:
            raise ValueError(f"Feature shape mismatch: ({d1},) vs ({d2},)")
        p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        lib = _capi.load_library()
        rc = lib.opd_similarity_matrix(self.device_ordinal, p(f1), p(b1), p(h1), n1, p(f2), p(b2), p(h2), n2, max(d1, d2, 1),
                                       float(self.appearance_weight), float(self.motion_weight), int(as_distance), p(out))
        _capi.check(rc, "opd_similarity_matrix")
        return out

    def compute_similarity_matrix(self, detections1: List[Detection], detections2: List[Detection]) -> np.ndarray:
        return self._matrix(detections1, detections2, False)

    def compute_distance_matrix(self, detections1: List[Detection], detections2: List[Detection]) -> np.ndarray:
        return self._matrix(detections1, detections2, True)
