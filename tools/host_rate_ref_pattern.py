#!/usr/bin/env python3
"""Frames/s at the REFERENCE's own calling patterns, through the host boundary (numpy frames in, Detection objects out; PCIe and Python
included; never the bench `value`):
  batch 1: `DetectionPhase.execute` calls `detector.detect_with_features(frame)` one frame at a time (src/pipeline/phases/detection.py:91-94);
  batch 4: the DETR era's `batch_size: 4` through `detect_batch` (config.yaml.disabled:44).
Both for frames at model resolution (800x1333) and for 720x1280 camera frames (device resize to 750x1333).  usage: host_rate_ref_pattern.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import HipDetrDetector
from office_person_detection_vit_amd.weights import DetrArch, ensure_weight_file
from office_person_detection_vit_amd.frames import structured_frame

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
path = ensure_weight_file(os.environ.get("OPD_WEIGHT_CACHE", "/tmp/opd_weights"), DetrArch.resnet50(), 0, 1.0, "r50")
for (H, W) in ((800, 1333), (720, 1280)):
    base = [structured_frame(H, W, 1234 + i) for i in range(8)]
    frames = [base[i % 8] for i in range(n)]
    for mb, what in ((1, "detect_with_features(frame)"), (4, "detect_batch(4 frames)")):
        det = HipDetrDetector(model_path=path, max_batch=mb, max_size=(800, 1333), resize=True)
        det.load_model()
        run = (lambda: [det.detect_with_features(f) for f in frames]) if mb == 1 else (lambda: [det.detect_batch(frames[i:i + 4]) for i in range(0, n, 4)])
        run()   # warm-up: graph capture
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            out = run()
            best = min(best, time.perf_counter() - t0)
        print(f"{H}x{W} host frames, max_batch={mb}, {what}: {n / best:8.1f} frames/s  ({1e3 * best / n:.2f} ms per frame)", flush=True)
        det.close()
