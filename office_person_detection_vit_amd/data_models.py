"""Output record of the detect path — field-for-field the reference's ``Detection`` dataclass
(``src/models/data_models.py:9-38``) so downstream tracking / transform code can consume it unchanged."""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np


@dataclass
class Detection:
    bbox: Tuple[float, float, float, float]  # (x, y, width, height), top-left + size, original-frame pixels
    confidence: float
    class_id: int
    class_name: str
    camera_coords: Tuple[float, float]  # foot point (x + w/2, y + h)
    floor_coords: Optional[Tuple[float, float]] = None
    floor_coords_mm: Optional[Tuple[float, float]] = None
    zone_ids: List[str] = field(default_factory=list)
    track_id: Optional[int] = None
    features: Optional[np.ndarray] = None
    appearance_score: Optional[float] = None
    query_index: Optional[int] = None
