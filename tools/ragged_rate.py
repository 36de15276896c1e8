#!/usr/bin/env python3
"""Blocking detect time of a ragged batch (mixed frame sizes: padding-mask path, launched eagerly) against a uniform batch
of the same canvas (hipGraph replay), batch 8."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from office_person_detection_vit_amd import HipDetrDetector
from office_person_detection_vit_amd.weights import DetrArch, ensure_weight_file
from office_person_detection_vit_amd.frames import structured_frame
path = ensure_weight_file(os.environ.get("OPD_WEIGHT_CACHE", "/tmp/opd_weights"), DetrArch.resnet50(), 0, 1.0, "r50")
det = HipDetrDetector(model_path=path, max_batch=8, max_size=(800, 1333), resize=False)
det.load_model()
uniform = [structured_frame(800, 1333, 10 + i) for i in range(8)]
ragged = [structured_frame(800, 1333, 10 + i) if i % 2 == 0 else structured_frame(750, 1333, 10 + i) for i in range(8)]
for name, frames in (("uniform 8 x 800x1333", uniform), ("ragged 4 x 800x1333 + 4 x 750x1333", ragged)):
    for _ in range(3):
        det.detect_batch(frames)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); out = det.detect_batch(frames); ts.append(time.perf_counter() - t0)
    print(f"{name}: {1e3 * min(ts):.2f} ms per batch (host frames, blocking), {sum(len(d) for d in out)} detections", flush=True)
