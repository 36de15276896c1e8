"""'Next' rows of SURVEY.md §8(f) on the host: the similarity oracle against the reference's own SimilarityCalculator
(golden made by tools/gen_golden.py from src/tracking/similarity.py), and the COCO exporter against the JSON the reference's
DetectionBenchmark evaluator accepted (precision / recall recorded at fixture time)."""

import json
import os

import numpy as np
import pytest

from office_person_detection_vit_amd.data_models import Detection
from office_person_detection_vit_amd.export import detections_to_coco, write_coco
from office_person_detection_vit_amd.similarity import SimilarityCalculator
from oracle import similarity_oracle as SO


def test_similarity_oracle_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "similarity.npz"))
    sim = SO.similarity_matrix(g["f1"], g["b1"], None, g["f2"], g["b2"], g["has2"], 0.7, 0.3)
    np.testing.assert_allclose(sim, g["similarity"], atol=1e-7)      # fp32 dot: BLAS summation order
    np.testing.assert_allclose(1.0 - sim, g["distance"], atol=1e-7)
    # the feature-less detection (column 6) carries the motion term alone, renormalised: similarity == IoU
    want = [SO.iou_xywh(g["b1"][i], g["b2"][6]) for i in range(len(g["b1"]))]
    np.testing.assert_allclose(g["similarity"][:, 6], want, atol=1e-7)


def test_similarity_scalar_helpers_and_errors():
    """Known answers in the style of the reference's tests: identical boxes -> IoU 1, disjoint -> 0, half overlap -> 1/3;
    identical unit features -> cosine 1; weights must sum to one (similarity.py:31-34)."""
    s = SimilarityCalculator(0.7, 0.3)
    assert s.iou((0, 0, 10, 10), (0, 0, 10, 10)) == 1.0
    assert s.iou((0, 0, 10, 10), (20, 20, 5, 5)) == 0.0
    assert abs(s.iou((0, 0, 10, 10), (5, 0, 10, 10)) - 1.0 / 3.0) < 1e-12
    assert s.iou((0, 0, 0, 0), (0, 0, 0, 0)) == 0.0
    f = np.zeros(256, np.float32); f[3] = 1.0
    assert s.cosine_similarity(f, f) == 1.0 and s.cosine_distance(f, f) == 0.0
    with pytest.raises(ValueError):
        s.cosine_similarity(f, f[:128])
    with pytest.raises(ValueError):
        SimilarityCalculator(0.7, 0.4)
    d1 = Detection(bbox=(0, 0, 10, 10), confidence=0.9, class_id=1, class_name="person", camera_coords=(5, 10), features=f)
    d2 = Detection(bbox=(0, 0, 10, 10), confidence=0.9, class_id=1, class_name="person", camera_coords=(5, 10))
    assert s.compute_similarity(d1, d1) == 1.0
    assert s.compute_similarity(d1, d2) == 1.0      # no features on one side: IoU alone, renormalised
    assert s.compute_distance(d1, d1) == 0.0
    assert s.compute_similarity_matrix([], [d1]).shape == (0, 1)


def test_coco_export_matches_fixture_accepted_by_reference_evaluator(golden_dir, tmp_path):
    g = json.load(open(os.path.join(golden_dir, "coco_export.json"), encoding="utf-8"))
    pred = g["prediction"]
    dets = [[], []]
    for a in pred["annotations"]:
        x, y, w, h = a["bbox"]
        dets[a["image_id"]].append(Detection(bbox=(x, y, w, h), confidence=a["score"], class_id=1, class_name="person",
                                             camera_coords=(x + w / 2, y + h)))
    sizes = [(im["height"], im["width"]) for im in pred["images"]]
    got = detections_to_coco(dets, sizes, [im["file_name"] for im in pred["images"]])
    assert got == pred
    # the reference's DetectionBenchmark.evaluate scored this JSON at fixture time: 5 TP, 1 FP, 1 FN
    assert g["evaluator"] == "DetectionBenchmark"
    assert (g["metrics"]["true_positives"], g["metrics"]["false_positives"], g["metrics"]["false_negatives"]) == (5, 1, 1)
    assert set(got) == {"images", "categories", "annotations"} and got["categories"] == [{"id": 0, "name": "person"}]
    assert set(got["annotations"][0]) == {"id", "image_id", "category_id", "bbox", "area", "score", "iscrowd"}
    path = tmp_path / "out" / "detections.json"
    write_coco(str(path), got)
    assert json.load(open(path, encoding="utf-8")) == pred
    with pytest.raises(ValueError):
        detections_to_coco(dets, sizes[:1])


# ---- BASELINE configs[4]: tiled high-resolution detection (host logic) -------------------------------------------------------
def test_tile_grid_and_merge():
    from office_person_detection_vit_amd.tiling import TiledDetector, merge_tile_detections, split_tiles, tile_grid
    assert tile_grid(2160, 3840) == [(0, 0, 1080, 1920), (0, 1920, 1080, 1920), (1080, 0, 1080, 1920), (1080, 1920, 1080, 1920)]
    assert tile_grid(7, 5, 2, 2) == [(0, 0, 3, 2), (0, 2, 3, 3), (3, 0, 4, 2), (3, 2, 4, 3)]      # remainders go last
    frame = np.arange(7 * 5 * 3, dtype=np.uint8).reshape(7, 5, 3)
    tiles, origins = split_tiles(frame, 2, 2)
    assert origins == [(0, 0), (0, 2), (3, 0), (3, 2)] and all(t.flags["C_CONTIGUOUS"] for t in tiles)
    rebuilt = np.zeros_like(frame)
    for t, (y, x) in zip(tiles, origins):
        rebuilt[y:y + t.shape[0], x:x + t.shape[1]] = t
    np.testing.assert_array_equal(rebuilt, frame)
    mk = lambda b, c, q: Detection(bbox=b, confidence=c, class_id=1, class_name="person", camera_coords=(0.0, 0.0), query_index=q)
    # a person on the vertical seam seen by both left tiles' neighbours: tile 0 box touches x = 1920, tile 1 sees the same body
    t0 = [mk((1800.0, 500.0, 120.0, 300.0), 0.9, 3)]
    t1 = [mk((0.0, 505.0, 30.0, 290.0), 0.6, 7), mk((400.0, 100.0, 80.0, 200.0), 0.8, 9)]
    out = merge_tile_detections([t0, t1, [], []], [(0, 0), (0, 1920), (1080, 0), (1080, 1920)], nms_threshold=0.4)
    assert [d.query_index for d in out] == [3, 9, 7]                      # disjoint in frame coordinates: all kept, by score
    assert out[1].bbox == (2320.0, 100.0, 80.0, 200.0) and out[1].camera_coords == (2360.0, 300.0)
    dup = merge_tile_detections([[mk((1900.0, 500.0, 100.0, 300.0), 0.9, 1)], [mk((-18.0, 502.0, 98.0, 296.0), 0.7, 2)]],
                                [(0, 0), (0, 1920)], nms_threshold=0.4)
    assert [d.query_index for d in dup] == [1]                              # same body across the seam: the weaker one goes

    # --- overlapping tiles + cut-aware merge (a body cut by a seam gives two partial boxes with IoU ~ 0) ---
    grid = tile_grid(2160, 3840, 2, 2, overlap=0.125)
    assert grid == [(0, 0, 1215, 2160), (0, 1680, 1215, 2160), (945, 0, 1215, 2160), (945, 1680, 1215, 2160)]
    origins = [(y, x) for y, x, _, _ in grid]
    sizes = [(h, w) for _, _, h, w in grid]
    # (1) a person 150 px wide standing on the base seam x = 1920 (frame x 1850..2000): narrower than the 240-px band, so BOTH
    # tiles see it whole (tile 0 up to x = 2160, tile 1 from x = 1680) -> ordinary duplicates, IoU-NMS keeps the stronger one
    whole0 = mk((1850.0, 400.0, 150.0, 400.0), 0.8, 1)
    whole1 = mk((1850.0 - 1680.0, 402.0, 150.0, 398.0), 0.9, 2)
    out = merge_tile_detections([[whole0], [whole1], [], []], origins, 0.4, sizes, (2160, 3840))
    assert [d.query_index for d in out] == [2] and out[0].bbox == (1850.0, 402.0, 150.0, 398.0)
    # (2) a person at frame x 2100..2250: tile 1 sees it whole, tile 0 only its first 60 px (cut by tile 0's right edge x = 2160);
    # IoU = 0.4 exactly at best, here 60/150 -> below the NMS threshold, so round 1's merge counted it twice
    part0 = mk((2100.0, 400.0, 60.0, 400.0), 0.95, 3)             # the partial box even scores higher
    full1 = mk((2100.0 - 1680.0, 400.0, 150.0, 400.0), 0.7, 4)
    plain = merge_tile_detections([[part0], [full1], [], []], origins, 0.4)
    assert len(plain) == 2                                           # no tile geometry -> plain IoU-NMS -> double count
    out = merge_tile_detections([[part0], [full1], [], []], origins, 0.4, sizes, (2160, 3840))
    assert len(out) == 1 and out[0].bbox == (2100.0, 400.0, 150.0, 400.0) and out[0].confidence == 0.95
    assert out[0].camera_coords == (2175.0, 800.0)
    # (3) a body wider than the band (a desk row, x 1500..2400): both tiles cut it; the parts overlap in the band only
    left = mk((1500.0, 900.0, 660.0, 200.0), 0.8, 5)              # tile 0: up to its right edge 2160
    right = mk((0.0, 900.0, 720.0, 200.0), 0.6, 6)                # tile 1: from its left edge 1680 to 2400
    out = merge_tile_detections([[left], [right], [], []], origins, 0.4, sizes, (2160, 3840))
    assert len(out) == 1 and out[0].bbox == (1500.0, 900.0, 900.0, 200.0)
    # (4) two different people in the band, both whole in both tiles: nothing is cut -> no merging, only IoU duplicates go
    a0, b0 = mk((1700.0, 300.0, 100.0, 300.0), 0.9, 7), mk((1760.0, 320.0, 100.0, 300.0), 0.8, 8)
    a1, b1 = mk((20.0, 300.0, 100.0, 300.0), 0.85, 9), mk((80.0, 320.0, 100.0, 300.0), 0.75, 10)
    out = merge_tile_detections([[a0, b0], [a1, b1], [], []], origins, 0.5, sizes, (2160, 3840))
    assert sorted(d.query_index for d in out) == [7, 8]

    class Fake:                                                             # detect_batch contract: one list per tile, in order
        def detect_batch(self, tiles):
            return [[mk((1.0, 2.0, 3.0, 4.0), 0.5 + 0.1 * i, i)] for i, _ in enumerate(tiles)]
    res = TiledDetector(Fake(), overlap=0.0).detect_batch([np.zeros((8, 8, 3), np.uint8), np.zeros((8, 8, 3), np.uint8)])
    assert [len(r) for r in res] == [4, 4] and res[1][0].bbox == (5.0, 6.0, 3.0, 4.0)
    with pytest.raises(ValueError):
        tile_grid(1, 5, 2, 2)


def test_evaluator_matches_reference_class(golden_dir):
    """tests/evaluator_checker.py::DetectionEvaluator (the checker of the GPU export test) against the reference's DetectionBenchmark.evaluate (tests/golden/evaluation.json,
    tools/gen_golden.py evaluation): counts exactly, every float to 1e-12, for both prediction layouts and four threshold pairs."""
    import json
    from evaluator_checker import DetectionEvaluator, evaluate_detections
    g = json.load(open(os.path.join(golden_dir, "evaluation.json"), encoding="utf-8"))
    assert len(g["runs"]) == 8
    for run in g["runs"]:
        pred = g["pred_coco"] if run["layout"] == "coco" else g["pred_frames"]
        got = DetectionEvaluator(run["iou_threshold"], run["confidence_threshold"]).evaluate(g["ground_truth"], pred).to_dict()
        want = run["metrics"]
        assert set(got) == set(want)
        for k, v in want.items():
            if isinstance(v, int):
                assert got[k] == v, (run, k)
            else:
                assert abs(got[k] - v) <= 1e-12, (run, k, got[k], v)
    m = evaluate_detections(g["ground_truth"], g["pred_coco"])
    assert m.true_positives + m.false_negatives == m.gt_count and "AP@50" in m.summary()
    # the exporter's own fixture was scored by the reference class when it was generated: same numbers from this evaluator
    e = json.load(open(os.path.join(golden_dir, "coco_export.json"), encoding="utf-8"))
    got = evaluate_detections(e["ground_truth"], e["prediction"]).to_dict()
    for k, v in e["metrics"].items():
        assert got[k] == pytest.approx(v, abs=1e-12), k


def test_evaluator_edge_cases():
    from evaluator_checker import DetectionEvaluator, average_precision_11pt, box_iou_xywh, match_image
    ev = DetectionEvaluator()
    empty = ev.evaluate({"annotations": []}, {"annotations": []})
    assert (empty.precision, empty.recall, empty.ap, empty.num_images) == (0.0, 0.0, 0.0, 0)
    gt = {"annotations": [{"image_id": 0, "category_id": 0, "bbox": [0, 0, 10, 10]}]}
    miss = ev.evaluate(gt, {"annotations": []})
    assert (miss.false_negatives, miss.recall, miss.ap_50) == (1, 0.0, 0.0)
    hit = ev.evaluate(gt, {"annotations": [{"image_id": 0, "category_id": 0, "bbox": [0, 0, 10, 10], "score": 0.9},
                                           {"image_id": 0, "category_id": 0, "bbox": [0, 0, 10, 10], "score": 0.8}]})
    assert (hit.true_positives, hit.false_positives, hit.precision, hit.recall) == (1, 1, 0.5, 1.0)   # the duplicate is a false positive
    assert abs(hit.ap_50 - 1.0) < 1e-12 and abs(hit.ap - 1.0) < 1e-12
    assert box_iou_xywh([0, 0, 10, 10], [5, 0, 10, 10]) == pytest.approx(1 / 3) and box_iou_xywh([0, 0, 0, 0], [0, 0, 0, 0]) == 0.0
    assert box_iou_xywh([0, 0, 1], [0, 0, 1, 1]) == 0.0
    tps, fps, missed = match_image([{"bbox": [0, 0, 10, 10]}, {"bbox": [20, 0, 10, 10]}], [{"bbox": [1, 0, 10, 10], "score": 0.5}], 0.5)
    assert len(tps) == 1 and fps == [] and missed == 1
    assert average_precision_11pt([]) == 0.0 and average_precision_11pt([(0.9, False)]) == 0.0
