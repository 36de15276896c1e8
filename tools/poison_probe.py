#!/usr/bin/env python3
"""GPU diagnosis: do the forward's results depend on memory it has not written?

Handles are created in the library's poison mode (opd_test_set_alloc_poison): every device buffer is pre-filled with one byte
and sits between two 256-KiB red zones of the same byte.  The same batches go through an unpoisoned handle and through handles
poisoned with 0x00 and 0xFF (fp16 / fp32 NaN patterns); any output difference means that some kernel consumes uninitialised
workspace or reads next to its buffers."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from office_person_detection_vit_amd import HipDetrDetector, _capi  # noqa: E402
from office_person_detection_vit_amd.frames import structured_frames  # noqa: E402
from office_person_detection_vit_amd.weights import DetrArch, ensure_weight_file  # noqa: E402

lib = _capi.load_library(test_hooks=True)
big = len(sys.argv) > 1 and sys.argv[1] == "big"
mild = ensure_weight_file(os.environ.get("OPD_WEIGHT_CACHE", "/tmp/opd_weights"), DetrArch.resnet50(), 0, 1.0, "r50")


def run_all(poison):
    lib.opd_test_set_alloc_poison(poison)
    out = {}
    try:
        det = HipDetrDetector(model_path=mild, max_batch=8 if big else 2, max_size=(800, 1333), resize=True)
        det.load_model()
    finally:
        lib.opd_test_set_alloc_poison(-1)
    if big:
        fr = structured_frames(8, 800, 1333, seed=99)
        for rep in range(3):
            out[f"b8_800x1333_{rep}"] = det.forward_raw(fr)
        recs = det.detect_batch(fr)
        out["b8_det"] = (np.asarray([[d.bbox + (d.confidence,) for d in f][:5] + [(0,) * 5] * (5 - min(5, len(f))) for f in recs], np.float64),)
    else:
        det.resize = False
        uni = structured_frames(2, 256, 320, seed=4321)
        for rep in range(3):   # eager, capture + launch, replay
            out[f"uniform_{rep}"] = det.forward_raw(uni)
        rag = [structured_frames(1, 256, 320, seed=5)[0], structured_frames(1, 224, 288, seed=6)[0]]
        out["ragged"] = det.forward_raw(rag)
        odd = structured_frames(1, 203, 333, seed=7)
        out["odd"] = det.forward_raw(odd)
        det.resize = True
        cam = structured_frames(1, 720, 1280, seed=8)
        for rep in range(2):
            out[f"resized_{rep}"] = det.forward_raw(cam)
        d, f = det.detect_with_features(cam[0])
        out["features"] = (f if len(d) else np.zeros((1, 256), np.float32),)
    if poison >= 0:   # nothing may have been written next to a buffer either
        bad = lib.opd_test_check_redzones(C.c_void_p(det.model))
        print(f"poison {poison:>4}: red zones damaged: {bad}" + (f"  first: {_capi.last_error()}" if bad else ""), flush=True)
        out["_redzones_damaged"] = (np.asarray([bad]),)
    det.close()
    return out


ref = run_all(-1)
rc = 0
for poison in (0x00, 0xFF, -1):
    got = run_all(poison)
    rc |= int(got.pop("_redzones_damaged", (np.zeros(1),))[0].sum() > 0)
    for k in ref:
        if k not in got:
            continue
        d = [float(np.nanmax(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) if np.asarray(a).size else 0.0 for a, b in zip(got[k], ref[k])]
        nan = [int(np.isnan(np.asarray(a, np.float64)).sum()) for a in got[k]]
        flag = "" if max(d) == 0.0 and sum(nan) == 0 else "   <-- DIFFERS"
        rc |= bool(flag)
        print(f"poison {poison:>4}: {k:18s} max|diff| vs unpoisoned {d}  NaNs {nan}{flag}", flush=True)
sys.exit(rc)
