"""Accuracy harness for the detector's COCO-style output.

TEST INFRASTRUCTURE (checker of tests/test_detector_gpu.py::test_export_scored_like_oracle), not shipped in the package.
Host-side restatement of the reference's detection evaluator, ``DetectionBenchmark.evaluate``
(``src/evaluation/detection_benchmark.py:201-503``): it scores what ``export.detections_to_coco`` writes against a COCO ground
truth and reports precision / recall / F1, AP@0.5, AP@0.75 and AP@[0.5:0.95].  The reference's conventions are kept exactly,
because the numbers it publishes come from them:

* only annotations of ``person_category_id`` count, on both sides (``:286-345``; the custom ``frames`` prediction layout with
  ``det`` / ``bb`` / ``conf`` keys is accepted too);
* predictions below ``confidence_threshold`` are dropped for precision / recall / AP@0.5 but NOT for AP@0.75 and the COCO
  average (``_calculate_ap_at_iou`` matches the unfiltered lists, ``:466-488``);
* greedy matching per image in descending score order, a prediction takes the free ground-truth box of strictly greatest
  IoU and is a true positive when that IoU >= the threshold (``:340-403``); boxes are (x, y, w, h);
* AP is the 11-point interpolation over the matched / unmatched predictions, with recall measured against the number of
  TRUE POSITIVES in the list, not the number of ground-truth boxes (``:431-464``) — so a missed box lowers recall but not AP;
* "AP@0.5" is taken from the main pass, i.e. at ``iou_threshold`` whatever its value (``:256``).

Pinned by ``tests/golden/evaluation.json``: the reference's own class run on seeded boxes (``tools/gen_golden.py``).
"""
from __future__ import annotations

from dataclasses import asdict, dataclass
from typing import Any, Dict, List, Tuple

import numpy as np


@dataclass
class DetectionMetrics:
    """Same fields as the reference's ``DetectionMetrics`` (``src/evaluation/detection_benchmark.py:21-75``)."""

    precision: float = 0.0
    recall: float = 0.0
    f1_score: float = 0.0
    ap_50: float = 0.0
    ap_75: float = 0.0
    ap: float = 0.0
    true_positives: int = 0
    false_positives: int = 0
    false_negatives: int = 0
    gt_count: int = 0
    pred_count: int = 0
    iou_threshold: float = 0.5
    confidence_threshold: float = 0.0
    num_images: int = 0

    def to_dict(self) -> Dict[str, Any]:
        return asdict(self)

    def summary(self) -> str:
        return (f"Precision: {self.precision:.2%}, Recall: {self.recall:.2%}, F1: {self.f1_score:.2%}, "
                f"AP@50: {self.ap_50:.2%}, mAP: {self.ap:.2%}")


def box_iou_xywh(a, b) -> float:
    """IoU of two (x, y, w, h) boxes; malformed boxes and empty unions give 0 (``:405-429``)."""
    if len(a) != 4 or len(b) != 4:
        return 0.0
    ax, ay, aw, ah = a
    bx, by, bw, bh = b
    iw = max(0, min(ax + aw, bx + bw) - max(ax, bx))
    ih = max(0, min(ay + ah, by + bh) - max(ay, by))
    inter = iw * ih
    union = aw * ah + bw * bh - inter
    return float(inter / union) if union > 0 else 0.0


def _ground_truth_by_image(data: Dict[str, Any], category: int) -> Dict[int, List[dict]]:
    out: Dict[int, List[dict]] = {}
    for ann in data.get("annotations", []):
        if ann.get("category_id") == category:
            out.setdefault(ann["image_id"], []).append({"bbox": ann["bbox"], "id": ann.get("id")})
    return out


def _predictions_by_image(data: Dict[str, Any], category: int) -> Dict[int, List[dict]]:
    out: Dict[int, List[dict]] = {}
    if "annotations" in data:
        for ann in data["annotations"]:
            if ann.get("category_id") == category:
                out.setdefault(ann["image_id"], []).append({"bbox": ann["bbox"], "score": ann.get("score", 1.0), "id": ann.get("id")})
    elif "frames" in data:
        for frame in data["frames"]:
            boxes = out.setdefault(frame.get("frame_idx", frame.get("idx", 0)), [])
            for det in frame.get("det", frame.get("detections", [])):
                boxes.append({"bbox": det.get("bb", det.get("bbox", [])), "score": det.get("conf", det.get("confidence", 1.0)),
                              "id": det.get("id")})
    return out


def match_image(gt: List[dict], pred: List[dict], iou_threshold: float) -> Tuple[List[Tuple[float, float]], List[float], int]:
    """One image: ([(score, iou) of each true positive], [scores of the false positives], missed ground-truth boxes)."""
    taken = set()
    tps: List[Tuple[float, float]] = []
    fps: List[float] = []
    for p in sorted(pred, key=lambda d: d.get("score", 0), reverse=True):   # stable: ties keep file order
        best, best_j = 0.0, -1
        for j, g in enumerate(gt):
            if j in taken:
                continue
            iou = box_iou_xywh(p["bbox"], g["bbox"])
            if iou > best:
                best, best_j = iou, j
        if best >= iou_threshold and best_j >= 0:
            taken.add(best_j)
            tps.append((p.get("score", 1.0), best))
        else:
            fps.append(p.get("score", 1.0))
    return tps, fps, len(gt) - len(taken)


def average_precision_11pt(scored: List[Tuple[float, bool]]) -> float:
    """11-point interpolated AP of (score, is_true_positive) pairs, recall relative to the true positives present."""
    if not scored:
        return 0.0
    ranked = sorted(scored, key=lambda s: s[0], reverse=True)
    positives = sum(1 for _, ok in ranked if ok)
    if positives == 0:
        return 0.0
    tp = fp = 0
    prec, rec = [], []
    for _, ok in ranked:
        tp += ok
        fp += not ok
        prec.append(tp / (tp + fp))
        rec.append(tp / positives)
    ap = 0.0
    for t in np.arange(0, 1.1, 0.1):   # the reference's grid, float steps included
        ap += max((p for p, r in zip(prec, rec) if r >= t), default=0.0) / 11
    return float(ap)


class DetectionEvaluator:
    """``DetectionBenchmark`` of the reference: ``evaluate(gt_data, pred_data) -> DetectionMetrics``."""

    def __init__(self, iou_threshold: float = 0.5, confidence_threshold: float = 0.0, person_category_id: int = 0):
        self.iou_threshold = iou_threshold
        self.confidence_threshold = confidence_threshold
        self.person_category_id = person_category_id

    def _scored(self, gt_by, pred_by, iou_threshold: float, min_score=None):
        scored: List[Tuple[float, bool]] = []
        tp = fp = fn = 0
        for image_id in set(gt_by.keys()) | set(pred_by.keys()):
            pred = pred_by.get(image_id, [])
            if min_score is not None:
                pred = [p for p in pred if p.get("score", 1.0) >= min_score]
            tps, fps, missed = match_image(gt_by.get(image_id, []), pred, iou_threshold)
            scored += [(s, True) for s, _ in tps] + [(s, False) for s in fps]
            tp, fp, fn = tp + len(tps), fp + len(fps), fn + missed
        return scored, tp, fp, fn

    def evaluate(self, gt_data: Dict[str, Any], pred_data: Dict[str, Any]) -> DetectionMetrics:
        gt_by = _ground_truth_by_image(gt_data, self.person_category_id)
        pred_by = _predictions_by_image(pred_data, self.person_category_id)
        scored, tp, fp, fn = self._scored(gt_by, pred_by, self.iou_threshold, self.confidence_threshold)
        precision = tp / (tp + fp) if tp + fp > 0 else 0.0
        recall = tp / (tp + fn) if tp + fn > 0 else 0.0
        f1 = 2 * precision * recall / (precision + recall) if precision + recall > 0 else 0.0
        at = lambda thr: average_precision_11pt(self._scored(gt_by, pred_by, thr)[0])
        coco = [at(t) for t in np.arange(0.5, 1.0, 0.05)]
        return DetectionMetrics(precision=precision, recall=recall, f1_score=f1, ap_50=average_precision_11pt(scored), ap_75=at(0.75),
                                ap=float(np.mean(coco)) if coco else 0.0, true_positives=tp, false_positives=fp, false_negatives=fn,
                                gt_count=sum(len(v) for v in gt_by.values()), pred_count=sum(len(v) for v in pred_by.values()),
                                iou_threshold=self.iou_threshold, confidence_threshold=self.confidence_threshold,
                                num_images=len(set(gt_by.keys()) | set(pred_by.keys())))


def evaluate_detections(gt_data: Dict[str, Any], pred_data: Dict[str, Any], iou_threshold: float = 0.5,
                        confidence_threshold: float = 0.0, person_category_id: int = 0) -> DetectionMetrics:
    return DetectionEvaluator(iou_threshold, confidence_threshold, person_category_id).evaluate(gt_data, pred_data)
