"""Synthetic "office camera" frames (BGR uint8 HxWx3), numpy only, bit-stable from a seed.

The reference's own detector tests feed ``np.random.randint(0, 255, (720, 1280, 3))``
(``tests/test_yolov8_detector.py:17-20``).  White noise makes the stage-4 feature map spatially
uniform, which hides attention bugs, so parity tests use *structured* frames:
a bilinearly up-sampled coarse random field (32-px cells) plus a dozen flat rectangles.
"""

from __future__ import annotations

import numpy as np


def _bilinear_upsample(coarse: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """coarse [h,w,c] float32 -> [out_h,out_w,c], half-pixel-centre bilinear (align_corners=False)."""
    h, w, _ = coarse.shape
    ys = (np.arange(out_h, dtype=np.float64) + 0.5) * (h / out_h) - 0.5
    xs = (np.arange(out_w, dtype=np.float64) + 0.5) * (w / out_w) - 0.5
    ys = np.clip(ys, 0, h - 1)
    xs = np.clip(xs, 0, w - 1)
    y0 = np.floor(ys).astype(np.int64)
    x0 = np.floor(xs).astype(np.int64)
    y1 = np.minimum(y0 + 1, h - 1)
    x1 = np.minimum(x0 + 1, w - 1)
    wy = (ys - y0).astype(np.float32)[:, None, None]
    wx = (xs - x0).astype(np.float32)[None, :, None]
    top = coarse[y0][:, x0] * (1 - wx) + coarse[y0][:, x1] * wx
    bot = coarse[y1][:, x0] * (1 - wx) + coarse[y1][:, x1] * wx
    return top * (1 - wy) + bot * wy


def structured_frame(height: int, width: int, seed: int, cell: int = 32, rects: int = 12) -> np.ndarray:
    """One structured BGR uint8 frame."""
    rng = np.random.default_rng(seed)
    ch, cw = max(2, -(-height // cell)), max(2, -(-width // cell))
    field = rng.random((ch, cw, 3), dtype=np.float32)
    img = _bilinear_upsample(field, height, width)
    for _ in range(rects):
        rh = int(rng.integers(max(2, height // 16), max(3, height // 3)))
        rw = int(rng.integers(max(2, width // 24), max(3, width // 5)))
        y = int(rng.integers(0, max(1, height - rh)))
        x = int(rng.integers(0, max(1, width - rw)))
        img[y:y + rh, x:x + rw, :] = rng.random(3, dtype=np.float32)
    # C-contiguous like a decoded camera frame (the up-sampling above leaves a transposed view behind)
    return np.ascontiguousarray(np.clip(np.rint(img * 255.0), 0, 255).astype(np.uint8))


def structured_frames(n: int, height: int, width: int, seed: int = 1234) -> list:
    return [structured_frame(height, width, seed + i) for i in range(n)]


def noise_frame(height: int, width: int, seed: int) -> np.ndarray:
    """White-noise BGR frame — the shape the reference's tests use; fine for throughput runs."""
    return np.random.default_rng(seed).integers(0, 255, (height, width, 3), dtype=np.uint8)
