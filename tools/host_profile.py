#!/usr/bin/env python3
"""cProfile of HipDetrDetector.detect_batch on host frames (where do the host-side milliseconds go?)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from office_person_detection_vit_amd import HipDetrDetector, _capi
from office_person_detection_vit_amd.weights import DetrArch, ensure_weight_file
from office_person_detection_vit_amd.frames import structured_frame
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (800, 1333)
path = ensure_weight_file(os.environ.get("OPD_WEIGHT_CACHE", "/tmp/opd_weights"), DetrArch.resnet50(), 0, 1.0, "r50")
frames = [structured_frame(H, W, 1234 + i) for i in range(8)] * 4
det = HipDetrDetector(model_path=path, max_batch=8, max_size=(800, 1333), resize=True, streams=1)
det.load_model()
det.detect_batch(frames[:8])
pr = cProfile.Profile()
pr.enable()
det.detect_batch(frames)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
# raw C-ABI timings on the same batch
lib = det._lib
batch, orig, valid, target = det._preprocess_batch(frames[:8])
B, Hh, Ww, _ = batch.shape
recs, counts = (_capi.OpdDet * (B * 100))(), (C.c_int32 * B)()
hw = np.asarray(orig, dtype=np.int32)
for name, arr in (("pinned", batch), ("pageable", np.array(batch))):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        if target is not None:
            rc = lib.opd_detr_detect_resized(C.c_void_p(det.model), arr.ctypes.data_as(C.c_void_p), _capi.OPD_MEM_HOST, B, Hh, Ww, target[0], target[1], 0.5, recs, counts)
        else:
            rc = lib.opd_detr_detect(C.c_void_p(det.model), arr.ctypes.data_as(C.c_void_p), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_HOST, B, Hh, Ww, 0.5,
                                     hw.ctypes.data_as(C.c_void_p), recs, counts)
        ts.append(time.perf_counter() - t0)
    print(f"C-ABI detect on {name} host batch: {1e3 * min(ts):.2f} ms (rc {rc})")
t0 = time.perf_counter(); det._preprocess_batch(frames[:8]); print(f"_preprocess_batch: {1e3 * (time.perf_counter() - t0):.2f} ms")
t0 = time.perf_counter(); det._postprocess_batch(recs, counts, 100); print(f"_postprocess_batch: {1e3 * (time.perf_counter() - t0):.2f} ms")
