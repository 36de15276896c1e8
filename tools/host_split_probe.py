#!/usr/bin/env python3
"""One blocking detect_batch call on 8 host frames (a single-stream caller): one handle at max_batch 8 against the same 8 frames as sub-batches on
several handles at once (max_batch 4 x 2 streams, max_batch 3 x 3 streams).  usage: host_split_probe.py [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import HipDetrDetector
from office_person_detection_vit_amd.weights import DetrArch, ensure_weight_file
from office_person_detection_vit_amd.frames import structured_frame

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
path = ensure_weight_file(os.environ.get("OPD_WEIGHT_CACHE", "/tmp/opd_weights"), DetrArch.resnet50(), 0, 1.0, "r50")
frames = [structured_frame(800, 1333, 1234 + i) for i in range(8)]
ref = None
for mb, streams in ((8, 1), (4, 2), (3, 3), (2, 4), (8, 1), (4, 2)):
    det = HipDetrDetector(model_path=path, max_batch=mb, max_size=(800, 1333), resize=True, streams=streams)
    det.load_model()
    out = det.detect_batch(frames); det.detect_batch(frames)
    sig = [[(d.query_index, round(d.confidence, 4)) for d in dets] for dets in out]
    ref = ref or sig
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n): det.detect_batch(frames)
        best = min(best, (time.perf_counter() - t0) / n)
    print(f"max_batch={mb} x streams={streams}: {1e3 * best:.3f} ms per call of 8 frames = {8 / best:7.1f} frames/s   (same detections as the first configuration: {sig == ref})", flush=True)
    det.close()
