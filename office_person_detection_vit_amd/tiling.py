"""High-resolution tiled detection (BASELINE.json configs[4]: a 4K office frame -> 2 x 2 tiles of 1080p, batched through the
detector, per-tile detections gathered and merged on the orchestrator).

The reference has no tiling mode; this is the workload the benchmark contract names, built from the path's own pieces:
tiles are ordinary frames for ``detect_batch`` (device-side resize, frame sharding over ranks when the wrapped detector is a
``ShardedDetector``), tile boxes are shifted back into frame coordinates, and duplicates along the tile seams are removed by
the same greedy IoU-NMS the per-frame path uses (``opd_person_nms``)."""

from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np

from . import _capi
from .data_models import Detection


def tile_grid(height: int, width: int, rows: int = 2, cols: int = 2) -> List[Tuple[int, int, int, int]]:
    """(y0, x0, h, w) of a rows x cols grid covering the frame exactly; the last row / column takes the remainder."""
    if rows < 1 or cols < 1 or height < rows or width < cols:
        raise ValueError("tile grid does not fit the frame")
    th, tw = height // rows, width // cols
    out = []
    for r in range(rows):
        for c in range(cols):
            y0, x0 = r * th, c * tw
            out.append((y0, x0, height - y0 if r == rows - 1 else th, width - x0 if c == cols - 1 else tw))
    return out


def split_tiles(frame: np.ndarray, rows: int = 2, cols: int = 2):
    """Contiguous tile copies (the detector requires contiguous uint8 HxWx3 frames) and their (y0, x0) origins."""
    grid = tile_grid(frame.shape[0], frame.shape[1], rows, cols)
    return [np.ascontiguousarray(frame[y:y + h, x:x + w]) for y, x, h, w in grid], [(y, x) for y, x, _, _ in grid]


def merge_tile_detections(tile_dets: Sequence[Sequence[Detection]], origins: Sequence[Tuple[int, int]],
                          nms_threshold: float = 0.4) -> List[Detection]:
    """Tile-local detections -> frame coordinates, then one greedy IoU-NMS over the union (a person on a seam is seen by
    two tiles).  ``query_index`` is kept per tile; ``camera_coords`` (foot point) is recomputed in frame coordinates."""
    if len(tile_dets) != len(origins):
        raise ValueError("one origin per tile is required")
    merged: List[Detection] = []
    for dets, (y0, x0) in zip(tile_dets, origins):
        for d in dets:
            x, y, w, h = d.bbox
            bbox = (x + x0, y + y0, w, h)
            merged.append(Detection(bbox=bbox, confidence=d.confidence, class_id=d.class_id, class_name=d.class_name,
                                    camera_coords=(bbox[0] + bbox[2] / 2, bbox[1] + bbox[3]), features=d.features,
                                    query_index=d.query_index))
    if len(merged) < 2 or nms_threshold >= 1.0:
        return sorted(merged, key=lambda d: -d.confidence)
    recs = (_capi.OpdDet * len(merged))()
    for i, d in enumerate(merged):
        x, y, w, h = d.bbox
        recs[i].x1, recs[i].y1, recs[i].x2, recs[i].y2 = x, y, x + w, y + h
        recs[i].score, recs[i].label, recs[i].query_index, recs[i].frame = d.confidence, 1, i, 0
    kept = _capi.load_library().opd_person_nms(recs, len(merged), 1, float(nms_threshold))
    if kept < 0:
        _capi.check(kept, "opd_person_nms")
    return [merged[recs[k].query_index] for k in range(kept)]


class TiledDetector:
    """``detect`` / ``detect_batch`` on frames that are cut into ``rows x cols`` tiles first.  ``detector`` is anything with
    ``detect_batch`` (a loaded ``HipDetrDetector`` or a ``ShardedDetector``)."""

    def __init__(self, detector, rows: int = 2, cols: int = 2, nms_threshold: float = 0.4):
        self.detector, self.rows, self.cols, self.nms_threshold = detector, rows, cols, nms_threshold

    def detect_batch(self, frames: Sequence[np.ndarray]) -> List[List[Detection]]:
        tiles, origins = [], []
        for f in frames:
            t, o = split_tiles(f, self.rows, self.cols)
            tiles.extend(t)
            origins.append(o)
        per_tile = self.detector.detect_batch(tiles) if tiles else []
        n = self.rows * self.cols
        return [merge_tile_detections(per_tile[i * n:(i + 1) * n], origins[i], self.nms_threshold) for i in range(len(frames))]

    def detect(self, frame: np.ndarray) -> List[Detection]:
        return self.detect_batch([frame])[0]
