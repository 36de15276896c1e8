#!/usr/bin/env python3
"""Frames/s through the reference-shaped HOST boundary: HipDetrDetector.detect_batch on numpy frames (PCIe-inclusive; never the
bench `value`).  usage: host_rate.py [n_frames] [height width]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from office_person_detection_vit_amd import HipDetrDetector
from office_person_detection_vit_amd.weights import DetrArch, ensure_weight_file
from office_person_detection_vit_amd.frames import structured_frame

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (800, 1333)
path = ensure_weight_file(os.environ.get("OPD_WEIGHT_CACHE", "/tmp/opd_weights"), DetrArch.resnet50(), 0, 1.0, "r50")
base = [structured_frame(H, W, 1234 + i) for i in range(8)]
frames = [base[i % 8] for i in range(n)]
# (lists: the chunk goes down as a list of frame pointers, opd_detr_detect_frames; otherwise it is stacked first, into page-locked or ordinary memory)
for streams, pinned, lists in ((1, False, True), (1, True, False), (1, False, False), (3, False, True), (3, True, False), (3, False, False)):
    det = HipDetrDetector(model_path=path, max_batch=8, max_size=(800, 1333), resize=True, streams=streams, pinned_staging=pinned, frame_lists=lists)
    det.load_model()
    det.detect_batch(frames[:8 * streams])   # warm-up: graph capture per handle
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        out = det.detect_batch(frames)
        best = min(best, time.perf_counter() - t0)
    print(f"{H}x{W} host frames, streams={streams}, {'frame list' if lists else 'stacked, pinned' if pinned else 'stacked, pageable'}: {n / best:8.1f} frames/s  ({1e3 * best / (n / 8):.2f} ms per batch of 8, "
          f"{sum(len(d) for d in out)} detections)", flush=True)
    det.close()
