"""High-resolution tiled detection (BASELINE.json configs[4]: a 4K office frame -> 2 x 2 tiles of 1080p, batched through the
detector, per-tile detections gathered and merged on the orchestrator).

The reference has no tiling mode; this is the workload the benchmark contract names, built from the path's own pieces:
tiles are ordinary frames for ``detect_batch`` (device-side resize, frame sharding over ranks when the wrapped detector is a
``ShardedDetector``) and tile boxes are shifted back into frame coordinates.

Seams.  A body cut by a tile edge gives two partial boxes whose IoU is near zero, so IoU-NMS alone would count it twice.
Therefore (a) neighbouring tiles OVERLAP (``overlap`` = fraction of the base tile each interior edge is pushed into the
neighbour; default 1/8, i.e. 240 px of a 1920-px tile, more than a standing person's width in a 4K office view), so a body
narrower than the band is seen whole by at least one tile; and (b) the merge is a greedy non-maximum MERGE: in descending
score order a box is dropped when its IoU with a kept box exceeds ``nms_threshold`` (the per-frame path's rule), and when it
comes from a different tile than a kept box, one of the two is cut by an interior tile edge and their intersection covers more
than ``merge_threshold`` of the smaller box, the kept box grows to their union (the two parts of one body).  ``overlap=0``
with ``tile_sizes=None`` is the plain shift + IoU-NMS of round 1."""

from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from .data_models import Detection

DEFAULT_OVERLAP = 0.125


def tile_grid(height: int, width: int, rows: int = 2, cols: int = 2, overlap: float = 0.0) -> List[Tuple[int, int, int, int]]:
    """(y0, x0, h, w) of a rows x cols grid covering the frame; the last row / column of the base grid takes the remainder and
    every interior edge is moved ``overlap`` x (base tile size) into the neighbouring tile."""
    if rows < 1 or cols < 1 or height < rows or width < cols:
        raise ValueError("tile grid does not fit the frame")
    if not 0.0 <= overlap < 0.5:
        raise ValueError("overlap must be in [0, 0.5)")
    th, tw = height // rows, width // cols
    oy, ox = int(round(th * overlap)), int(round(tw * overlap))
    out = []
    for r in range(rows):
        for c in range(cols):
            y0, x0 = r * th, c * tw
            y1 = height if r == rows - 1 else y0 + th
            x1 = width if c == cols - 1 else x0 + tw
            y0, x0, y1, x1 = max(0, y0 - oy), max(0, x0 - ox), min(height, y1 + oy), min(width, x1 + ox)
            out.append((y0, x0, y1 - y0, x1 - x0))
    return out


def split_tiles(frame: np.ndarray, rows: int = 2, cols: int = 2, overlap: float = 0.0):
    """Contiguous tile copies (the detector requires contiguous uint8 HxWx3 frames) and their (y0, x0) origins."""
    grid = tile_grid(frame.shape[0], frame.shape[1], rows, cols, overlap)
    return [np.ascontiguousarray(frame[y:y + h, x:x + w]) for y, x, h, w in grid], [(y, x) for y, x, _, _ in grid]


def _iou_iomin(a, b) -> Tuple[float, float]:
    iw = min(a[2], b[2]) - max(a[0], b[0])
    ih = min(a[3], b[3]) - max(a[1], b[1])
    if iw <= 0 or ih <= 0:
        return 0.0, 0.0
    inter = iw * ih
    aa, ab = max(0.0, a[2] - a[0]) * max(0.0, a[3] - a[1]), max(0.0, b[2] - b[0]) * max(0.0, b[3] - b[1])
    union, small = aa + ab - inter, min(aa, ab)
    return (inter / union if union > 0 else 0.0), (inter / small if small > 0 else 0.0)


def merge_tile_detections(tile_dets: Sequence[Sequence[Detection]], origins: Sequence[Tuple[int, int]],
                          nms_threshold: float = 0.4, tile_sizes: Optional[Sequence[Tuple[int, int]]] = None,
                          frame_size: Optional[Tuple[int, int]] = None, merge_threshold: float = 0.5,
                          edge_tolerance: float = 0.01) -> List[Detection]:
    """Tile-local detections -> frame coordinates, then the greedy merge described in the module docstring.

    ``tile_sizes`` ((h, w) per tile) and ``frame_size`` ((H, W)) tell which box sides are cut by an INTERIOR tile edge (within
    ``edge_tolerance`` x tile size of it); without them no box counts as cut and the merge is plain IoU-NMS.
    ``query_index`` is kept per tile; ``camera_coords`` (foot point) is recomputed in frame coordinates."""
    if len(tile_dets) != len(origins) or (tile_sizes is not None and len(tile_sizes) != len(origins)):
        raise ValueError("one origin (and size) per tile is required")
    cands = []   # [xyxy, score, tile, cut, detection]
    for t, (dets, (y0, x0)) in enumerate(zip(tile_dets, origins)):
        for d in dets:
            x, y, w, h = d.bbox
            box = [x + x0, y + y0, x + x0 + w, y + y0 + h]
            cut = False
            if tile_sizes is not None and frame_size is not None:
                th, tw = tile_sizes[t]
                ty, tx = edge_tolerance * th, edge_tolerance * tw
                cut = ((x0 > 0 and x <= tx) or (x0 + tw < frame_size[1] and x + w >= tw - tx) or
                       (y0 > 0 and y <= ty) or (y0 + th < frame_size[0] and y + h >= th - ty))
            cands.append([box, float(d.confidence), t, cut, d])
    cands.sort(key=lambda c: -c[1])   # stable: ties keep tile / query order
    kept: List[list] = []
    for c in cands:
        absorbed = False
        if nms_threshold < 1.0:
            for k in kept:
                iou, iomin = _iou_iomin(c[0], k[0])
                if c[2] != k[2] and (c[3] or k[3]) and iomin > merge_threshold:   # two parts of one body: checked first, the
                    k[0] = [min(k[0][0], c[0][0]), min(k[0][1], c[0][1]), max(k[0][2], c[0][2]), max(k[0][3], c[0][3])]   # parts' IoU may be high too
                    k[3] = k[3] and c[3]   # a part that was not cut completes the body
                    absorbed = True
                elif iou > nms_threshold:
                    absorbed = True
                if absorbed:
                    break
        if not absorbed:
            kept.append(c)
    out = []
    for box, score, _, _, d in kept:
        bbox = (box[0], box[1], box[2] - box[0], box[3] - box[1])
        out.append(Detection(bbox=bbox, confidence=d.confidence, class_id=d.class_id, class_name=d.class_name,
                             camera_coords=(bbox[0] + bbox[2] / 2, bbox[1] + bbox[3]), features=d.features, query_index=d.query_index))
    return out


class TiledDetector:
    """``detect`` / ``detect_batch`` on frames that are cut into ``rows x cols`` overlapping tiles first.  ``detector`` is
    anything with ``detect_batch`` (a loaded ``HipDetrDetector`` or a ``ShardedDetector``)."""

    def __init__(self, detector, rows: int = 2, cols: int = 2, nms_threshold: float = 0.4, overlap: float = DEFAULT_OVERLAP,
                 merge_threshold: float = 0.5):
        self.detector, self.rows, self.cols, self.nms_threshold = detector, rows, cols, nms_threshold
        self.overlap, self.merge_threshold = overlap, merge_threshold

    def detect_batch(self, frames: Sequence[np.ndarray]) -> List[List[Detection]]:
        tiles, origins, sizes = [], [], []
        for f in frames:
            t, o = split_tiles(f, self.rows, self.cols, self.overlap)
            tiles.extend(t)
            origins.append(o)
            sizes.append([(x.shape[0], x.shape[1]) for x in t])
        per_tile = self.detector.detect_batch(tiles) if tiles else []
        n = self.rows * self.cols
        return [merge_tile_detections(per_tile[i * n:(i + 1) * n], origins[i], self.nms_threshold, sizes[i],
                                      (frames[i].shape[0], frames[i].shape[1]), self.merge_threshold) for i in range(len(frames))]

    def detect(self, frame: np.ndarray) -> List[Detection]:
        return self.detect_batch([frame])[0]
