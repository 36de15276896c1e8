#!/usr/bin/env python3
"""Generate the committed golden fixtures under ``tests/golden/`` (runs ONLY in the build container).

The reference's DETR arithmetic lives in Hugging Face ``transformers`` (reference pin 4.57.3; 5.15.0 is what this
container has — key renames only, same math).  This script builds ``DetrForObjectDetection`` from a
LOCAL config (no hub access), loads the repo's seeded synthetic weights into it, runs it on seeded structured frames
and stores inputs' seeds + outputs as small ``.npz`` files.  It also captures outputs of the reference's own
``src/tracking/feature_extractor.py`` (loaded by file path; its package ``__init__`` needs cv2) and of HF's
``post_process_object_detection`` / image processor.  Nothing here travels to the GPU box except the ``.npz`` data.

Usage:  python tools/gen_golden.py            (writes tests/golden/*.npz)
"""

from __future__ import annotations

import importlib.util
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from office_person_detection_vit_amd.frames import structured_frames  # noqa: E402
from office_person_detection_vit_amd.weights import DetrArch, synth_weights  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"


def build_hf(arch: DetrArch):
    from transformers import DetrConfig, DetrForObjectDetection, ResNetConfig

    cfg = DetrConfig(
        backbone_config=ResNetConfig(depths=list(arch.depths), out_features=["stage4"]),
        num_labels=arch.num_labels, num_queries=arch.num_queries,
        encoder_layers=arch.encoder_layers, decoder_layers=arch.decoder_layers,
        use_pretrained_backbone=False, use_timm_backbone=False,
    )
    cfg._attn_implementation = "eager"
    return DetrForObjectDetection(cfg).eval()


def hf_preprocess(frames_bgr):
    """HF DetrImageProcessor with resizing disabled (frames already at model resolution)."""
    from transformers import DetrImageProcessor

    proc = DetrImageProcessor(do_resize=False)
    rgb = [np.ascontiguousarray(f[:, :, ::-1]) for f in frames_bgr]
    enc = proc(images=rgb, return_tensors="pt")
    return proc, enc["pixel_values"], enc["pixel_mask"]


def model_case(tag, arch, seed, ga, sizes, frame_seed, full_outputs=True):
    w = synth_weights(arch, seed, ga)
    m = build_hf(arch)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    frames = [structured_frames(1, h, wd, seed=frame_seed + i)[0] for i, (h, wd) in enumerate(sizes)]
    proc, pv, pm = hf_preprocess(frames)
    with torch.no_grad():
        out = m(pixel_values=pv, pixel_mask=pm)
    target_sizes = [(f.shape[0], f.shape[1]) for f in frames]
    post = proc.post_process_object_detection(out, threshold=0.5, target_sizes=target_sizes)
    d = {
        "arch_depths": np.array(arch.depths), "seed": np.array(seed), "attention_gain": np.array(ga),
        "sizes": np.array(sizes), "frame_seed": np.array(frame_seed),
        "logits": out.logits.numpy(), "pred_boxes": out.pred_boxes.numpy(),
        "pixel_mask_sum": pm.sum(dim=(1, 2)).numpy(),
    }
    mem = out.encoder_last_hidden_state.numpy()
    if full_outputs:
        d["encoder_last_hidden_state"] = mem
        d["pixel_values_sample"] = pv[:, :, ::37, ::41].numpy()
    else:
        d["encoder_sum"] = mem.astype(np.float64).sum(axis=(1, 2))
        d["encoder_abs_sum"] = np.abs(mem.astype(np.float64)).sum(axis=(1, 2))
        d["encoder_sample"] = mem[:, ::97, ::13]
    for i, r in enumerate(post):
        d[f"post{i}_scores"] = r["scores"].numpy()
        d[f"post{i}_labels"] = r["labels"].numpy()
        d[f"post{i}_boxes"] = r["boxes"].numpy()
    np.savez_compressed(os.path.join(GOLD, f"{tag}.npz"), **d)
    sc = torch.softmax(out.logits, -1)[..., :-1].max(-1)[0]
    print(f"{tag}: logits {tuple(out.logits.shape)} box-std {float(out.pred_boxes.std(dim=1).mean()):.4f} "
          f"scores>0.5 {int((sc > 0.5).sum())}/{sc.numel()}")


def feature_extractor_case():
    spec = importlib.util.spec_from_file_location("ref_feature_extractor",
                                                  os.path.join(REFERENCE, "src/tracking/feature_extractor.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fe = mod.FeatureExtractor()
    rng = np.random.default_rng(7)
    enc = rng.standard_normal((25, 42, 256)).astype(np.float32)
    bboxes = [(100.0, 200.0, 50.0, 100.0), (0.0, 0.0, 1333.0, 800.0), (1300.0, 790.0, 80.0, 40.0),
              (-20.0, -10.0, 30.0, 30.0), (640.5, 399.2, 1.0, 1.0), (10.0, 700.0, 300.0, 99.0)]
    roi = fe.extract_roi_features(enc, bboxes, (800, 1333))
    raw = rng.standard_normal((5, 256)).astype(np.float32)
    raw[3] = 0.0
    norm = fe.normalize_features(raw)
    empty = fe.extract_roi_features(enc, [], (800, 1333))
    # enc/raw are regenerated in the test from default_rng(7) in the same draw order (keeps the fixture small)
    np.savez_compressed(os.path.join(GOLD, "feature_extractor.npz"), rng_seed=np.array(7), bboxes=np.array(bboxes),
                        image_shape=np.array([800, 1333]), roi=roi, norm=norm, empty_shape=np.array(empty.shape),
                        enc_sample=enc[::5, ::7, ::31])
    print("feature_extractor: roi", roi.shape, "norms", np.linalg.norm(roi, axis=1)[:3])


def _load_ref(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REFERENCE, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod   # dataclasses look their module up while the class body executes
    spec.loader.exec_module(mod)
    return mod


def similarity_and_export_case():
    """Next rows: the reference's own SimilarityCalculator (tracker cost matrix) and its detection
    evaluator run on seeded data; inputs are regenerated in the tests from the stored seed."""
    import types
    sim = _load_ref("ref_similarity", "src/tracking/similarity.py").SimilarityCalculator(0.7, 0.3)
    rng = np.random.default_rng(11)
    n1, n2 = 7, 9
    f1 = rng.standard_normal((n1, 256)).astype(np.float32)
    f2 = rng.standard_normal((n2, 256)).astype(np.float32)
    f1 /= np.linalg.norm(f1, axis=1, keepdims=True)
    f2 /= np.linalg.norm(f2, axis=1, keepdims=True)
    f2[:4] = (0.8 * f1[:4] + 0.2 * f2[:4]) / np.linalg.norm(0.8 * f1[:4] + 0.2 * f2[:4], axis=1, keepdims=True)
    b1 = (rng.uniform(0, 900, (n1, 4)) * [1, 0.6, 0.2, 0.3] + [0, 0, 20, 40]).astype(np.float32)
    b2 = (rng.uniform(0, 900, (n2, 4)) * [1, 0.6, 0.2, 0.3] + [0, 0, 20, 40]).astype(np.float32)
    b2[:4] = b1[:4] + rng.uniform(-8, 8, (4, 4)).astype(np.float32)
    has2 = np.ones(n2, bool)
    has2[6] = False   # a detection without features: the motion term alone, renormalised
    mk = lambda f, b, ok: types.SimpleNamespace(features=f if ok else None, bbox=tuple(float(v) for v in b))
    d1 = [mk(f1[i], b1[i], True) for i in range(n1)]
    d2 = [mk(f2[j], b2[j], bool(has2[j])) for j in range(n2)]
    simm = sim.compute_similarity_matrix(d1, d2)
    dist = sim.compute_distance_matrix(d1, d2)
    np.savez_compressed(os.path.join(GOLD, "similarity.npz"), seed=np.array(11), f1=f1, f2=f2, b1=b1, b2=b2, has2=has2,
                        similarity=simm, distance=dist)
    print("similarity:", simm.shape, float(simm.max()), float(simm.min()))

    # exporter + the reference evaluator: predictions = ground truth boxes jittered, one false positive, one miss
    from office_person_detection_vit_amd.data_models import Detection
    from office_person_detection_vit_amd.export import detections_to_coco
    bench = _load_ref("ref_detection_benchmark", "src/evaluation/detection_benchmark.py")
    gt = {"images": [{"id": 0, "file_name": "a.jpg", "width": 1280, "height": 720}, {"id": 1, "file_name": "b.jpg", "width": 1280, "height": 720}],
          "categories": [{"id": 0, "name": "person"}],
          "annotations": [{"id": i, "image_id": i // 3, "category_id": 0, "bbox": [100.0 + 200 * (i % 3), 150.0, 80.0, 200.0], "area": 16000.0, "iscrowd": 0}
                          for i in range(6)]}
    dets = [[], []]
    for a in gt["annotations"][:5]:   # the sixth ground-truth box is missed
        x, y, w, h = a["bbox"]
        bb = (x + 3.0, y - 2.0, w, h + 4.0)
        dets[a["image_id"]].append(Detection(bbox=bb, confidence=0.9 - 0.05 * a["id"], class_id=1, class_name="person",
                                             camera_coords=(bb[0] + bb[2] / 2, bb[1] + bb[3])))
    dets[1].append(Detection(bbox=(900.0, 400.0, 60.0, 120.0), confidence=0.55, class_id=1, class_name="person", camera_coords=(930.0, 520.0)))
    pred = detections_to_coco(dets, [(720, 1280), (720, 1280)], ["a.jpg", "b.jpg"])
    ev_cls = [getattr(bench, n) for n in dir(bench) if n.endswith("Benchmark") or n.endswith("Evaluator")]
    ev = ev_cls[0]()
    m = ev.evaluate(gt, pred)
    with open(os.path.join(GOLD, "coco_export.json"), "w", encoding="utf-8") as f:
        json.dump({"ground_truth": gt, "prediction": pred, "evaluator": type(ev).__name__,
                   "metrics": {"precision": m.precision, "recall": m.recall, "f1_score": m.f1_score, "true_positives": m.true_positives,
                               "false_positives": m.false_positives, "false_negatives": m.false_negatives, "pred_count": m.pred_count}}, f, indent=1)
    print("coco export:", type(ev).__name__, m.precision, m.recall, m.true_positives, m.false_positives, m.false_negatives)


def evaluation_case():
    """Next row (accuracy harness): the reference's DetectionBenchmark.evaluate on seeded ground truth and
    predictions (jittered boxes, duplicates, score ties, low-confidence clutter, images without one side, a non-person category)
    at several (IoU, confidence) settings, in both prediction layouts it accepts."""
    bench = _load_ref("ref_detection_benchmark", "src/evaluation/detection_benchmark.py")
    rng = np.random.default_rng(23)
    gt_ann, pred_ann, frames = [], [], []
    aid = 0
    for img in range(7):
        n = int(rng.integers(0, 6)) if img != 5 else 0          # image 5 has predictions only
        boxes = []
        for _ in range(n):
            x, y = float(rng.uniform(0, 1000)), float(rng.uniform(0, 500))
            w, h = float(rng.uniform(40, 160)), float(rng.uniform(80, 220))
            boxes.append([x, y, w, h])
            gt_ann.append({"id": aid, "image_id": img, "category_id": 0, "bbox": [x, y, w, h], "area": w * h, "iscrowd": 0}); aid += 1
        if img == 2:   # a ground-truth box of another category: ignored by the evaluator
            gt_ann.append({"id": aid, "image_id": img, "category_id": 3, "bbox": [10.0, 10.0, 50.0, 50.0], "area": 2500.0, "iscrowd": 0}); aid += 1
        dets = []
        if img != 6:   # image 6 has ground truth only
            for b in boxes:
                if rng.random() < 0.85:
                    j = rng.normal(0, [6, 6, 10, 14])
                    dets.append(([b[0] + j[0], b[1] + j[1], max(8.0, b[2] + j[2]), max(8.0, b[3] + j[3])], round(float(rng.uniform(0.3, 0.99)), 2)))
                if rng.random() < 0.3:   # a duplicate of the same person, shifted
                    dets.append(([b[0] + 15.0, b[1] - 10.0, b[2], b[3]], round(float(rng.uniform(0.2, 0.8)), 2)))
            for _ in range(int(rng.integers(0, 3))):   # clutter
                dets.append(([float(rng.uniform(0, 1100)), float(rng.uniform(0, 600)), float(rng.uniform(30, 120)), float(rng.uniform(60, 200))],
                             round(float(rng.uniform(0.05, 0.6)), 2)))
        for k, (bb, sc) in enumerate(dets):
            bb = [float(v) for v in bb]
            pred_ann.append({"id": len(pred_ann), "image_id": img, "category_id": 0, "bbox": bb, "score": sc})
        frames.append({"frame_idx": img, "det": [{"bb": [float(v) for v in bb], "conf": sc} for bb, sc in dets]})
    pred_ann.append({"id": len(pred_ann), "image_id": 1, "category_id": 2, "bbox": [5.0, 5.0, 40.0, 90.0], "score": 0.99})   # other category
    gt = {"images": [{"id": i, "file_name": f"{i}.jpg", "width": 1280, "height": 720} for i in range(7)],
          "categories": [{"id": 0, "name": "person"}], "annotations": gt_ann}
    pred_coco = {"images": gt["images"], "categories": gt["categories"], "annotations": pred_ann}
    pred_frames = {"frames": frames}
    runs = []
    for layout, pred in (("coco", pred_coco), ("frames", pred_frames)):
        for iou, conf in ((0.5, 0.0), (0.5, 0.5), (0.75, 0.3), (0.3, 0.0)):
            m = bench.DetectionBenchmark(iou_threshold=iou, confidence_threshold=conf, output_diagnostics=False).evaluate(gt, pred)
            runs.append({"layout": layout, "iou_threshold": iou, "confidence_threshold": conf, "metrics": m.to_dict()})
            print("evaluation", layout, iou, conf, m.summary())
    with open(os.path.join(GOLD, "evaluation.json"), "w", encoding="utf-8") as f:
        json.dump({"ground_truth": gt, "pred_coco": pred_coco, "pred_frames": pred_frames, "runs": runs}, f)


def resize_case():
    """HF image processor with its default resize (shortest 800 / longest 1333, PIL bilinear) on camera-sized frames."""
    from transformers import DetrImageProcessor

    proc = DetrImageProcessor()
    d = {}
    for tag, (h, w) in {"720x1280": (720, 1280), "480x640": (480, 640), "1080x1920": (1080, 1920), "900x700": (900, 700)}.items():
        frame = structured_frames(1, h, w, seed=555)[0]
        enc = proc(images=[np.ascontiguousarray(frame[:, :, ::-1])], return_tensors="pt")
        pv = enc["pixel_values"]
        d[f"{tag}_shape"] = np.array(pv.shape)
        d[f"{tag}_sample"] = pv[:, :, ::53, ::59].numpy()
        print("resize", tag, "->", tuple(pv.shape))
    np.savez_compressed(os.path.join(GOLD, "hf_resize.npz"), **d)


def main():
    os.makedirs(GOLD, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "evaluation":   # only the (fast) host-side fixture
        evaluation_case()
        return
    models_only = len(sys.argv) > 1 and sys.argv[1] == "models"   # after a change of the weight recipe (weights.RECIPE_VERSION)
    if not models_only:
        resize_case()
    r50 = DetrArch.resnet50()
    # equal-size batches (pixel_mask all ones): the configuration the HIP path serves
    model_case("r50_mild_256x320", r50, 0, 1.0, [(256, 320), (256, 320)], 1234)
    model_case("r50_sharp_256x320", r50, 0, 2.0, [(256, 320), (256, 320)], 1234)
    # ragged batch: zero padding + pixel_mask -> masked attention + cumsum position embedding (oracle only)
    model_case("r50_mild_ragged", r50, 0, 1.0, [(256, 320), (224, 288)], 2234)
    # odd sizes all the way down (H3), single frame
    model_case("r50_mild_odd_203x333", r50, 0, 1.0, [(203, 333)], 3234)
    # full benchmark resolution, one frame: logits/boxes in full, encoder output as checksums + samples
    model_case("r50_mild_800x1333", r50, 0, 1.0, [(800, 1333)], 1234, full_outputs=False)
    # r101 (config 4 architecture) at small size
    model_case("r101_mild_256x320", DetrArch.resnet101(), 0, 1.0, [(256, 320)], 1234)
    if models_only:
        return
    feature_extractor_case()
    similarity_and_export_case()
    evaluation_case()


if __name__ == "__main__":
    main()
