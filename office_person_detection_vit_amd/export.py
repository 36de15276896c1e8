"""Detection export in the reference's COCO-style JSON.

Schema = what ``tools/detect_yolov8.py:41-97`` writes ("same format as DETR") and what
``src/evaluation/detection_benchmark.py:201-339`` (``_parse_predictions``, ``person_category_id = 0``) reads:
``{"images": [{id, file_name, width, height}], "categories": [{"id": 0, "name": "person"}],
"annotations": [{id, image_id, category_id, bbox [x, y, w, h], area, score, iscrowd}]}``.
"""

from __future__ import annotations

import json
import os
from typing import Any, Dict, List, Optional, Sequence, Tuple

from .data_models import Detection

PERSON_CATEGORY_ID = 0  # the evaluator's default (detection_benchmark.py:150), the exporter's category (detect_yolov8.py:43)


def detections_to_coco(detections: Sequence[Sequence[Detection]], image_sizes: Sequence[Tuple[int, int]],
                       file_names: Optional[Sequence[str]] = None, category_id: int = PERSON_CATEGORY_ID) -> Dict[str, Any]:
    """Per-frame ``Detection`` lists (``detect_batch`` output) -> COCO-style dict.  ``image_sizes`` = (height, width) of each
    ORIGINAL frame; boxes are already top-left + size in original pixels (``Detection.bbox``)."""
    if len(detections) != len(image_sizes):
        raise ValueError("one (height, width) per frame is required")
    if file_names is not None and len(file_names) != len(detections):
        raise ValueError("one file name per frame is required")
    out: Dict[str, Any] = {"images": [], "categories": [{"id": category_id, "name": "person"}], "annotations": []}
    ann_id = 0
    for image_id, (dets, (h, w)) in enumerate(zip(detections, image_sizes)):
        out["images"].append({"id": image_id, "file_name": file_names[image_id] if file_names is not None else f"frame_{image_id:06d}.jpg",
                              "width": int(w), "height": int(h)})
        for d in dets:
            x, y, bw, bh = (float(v) for v in d.bbox)
            out["annotations"].append({"id": ann_id, "image_id": image_id, "category_id": category_id, "bbox": [x, y, bw, bh],
                                       "area": bw * bh, "score": float(d.confidence), "iscrowd": 0})
            ann_id += 1
    return out


def write_coco(path: str, coco: Dict[str, Any]) -> None:
    """Same serialisation as the reference writer (``detect_yolov8.py:94-97``): UTF-8, indent 2, parents created."""
    parent = os.path.dirname(os.path.abspath(path))
    os.makedirs(parent, exist_ok=True)
    with open(path, "w", encoding="utf-8") as f:
        json.dump(coco, f, indent=2, ensure_ascii=False)
