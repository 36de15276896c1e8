#!/usr/bin/env python3
"""Where one `detect_with_features(frame)` call at max_batch = 1 spends its host time (the reference's calling pattern,
src/pipeline/phases/detection.py:91-94): the staging copy, the C-ABI detect call from page-locked / from pageable memory, NMS + Python
objects, the ROI-feature call.  usage: host_b1_probe.py [n]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from office_person_detection_vit_amd import HipDetrDetector, _capi
from office_person_detection_vit_amd.weights import DetrArch, ensure_weight_file
from office_person_detection_vit_amd.frames import structured_frame

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
path = ensure_weight_file(os.environ.get("OPD_WEIGHT_CACHE", "/tmp/opd_weights"), DetrArch.resnet50(), 0, 1.0, "r50")
H, W = 800, 1333
frames = [structured_frame(H, W, 1234 + i) for i in range(8)]
det = HipDetrDetector(model_path=path, max_batch=1, max_size=(800, 1333), resize=True)
det.load_model()
for f in frames: det.detect_with_features(f)
lib = det._lib

def best(fn, reps=3):
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        for i in range(n): fn(frames[i % 8])
        b = min(b, (time.perf_counter() - t0) / n)
    return 1e3 * b

det2 = HipDetrDetector(model_path=path, max_batch=1, max_size=(800, 1333), resize=True, frame_lists=False)
det2.load_model()
for f in frames: det2.detect_with_features(f)
for rep in range(2):   # (interleaved: the two forms on the same box, twice)
    print(f"detect_with_features(frame)        {best(det.detect_with_features):.3f} ms   one call (opd_detr_detect_frames_features), frame uploaded from where it lies")
    print(f"detect_with_features(frame)        {best(det2.detect_with_features):.3f} ms   stacked copy + opd_detr_detect, then opd_detr_roi_features")
print(f"detect(frame)                      {best(det.detect):.3f} ms")
print(f"detect(frame), stacked             {best(det2.detect):.3f} ms")
print(f"_preprocess_batch([frame])         {best(lambda f: det._preprocess_batch([f])):.3f} ms   (validation + copy into the page-locked staging array)")
Q = det._info.num_queries
recs, counts = (_capi.OpdDet * Q)(), (C.c_int32 * 1)()
hw = np.asarray([[H, W]], dtype=np.int32)
staged = det._preprocess_batch([frames[0]])[0]
def cabi(ptr):
    rc = lib.opd_detr_detect_ragged(C.c_void_p(det.model), ptr, _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_HOST, 1, H, W, None, 0.5,
                                    hw.ctypes.data_as(C.c_void_p), recs, counts)
    assert rc == 0
sp = staged.ctypes.data_as(C.c_void_p)
print(f"C-ABI detect, page-locked frame    {best(lambda f: cabi(sp)):.3f} ms")
print(f"C-ABI detect, pageable frame       {best(lambda f: cabi(f.ctypes.data_as(C.c_void_p))):.3f} ms")
feats = np.empty((1, Q, 256), np.float32)
def cabi_list(f, with_feats):
    ptrs = (C.c_void_p * 1)(f.ctypes.data)
    if with_feats:
        rc = lib.opd_detr_detect_frames_features(C.c_void_p(det.model), ptrs, 1, H, W, H, W, 0.5, 1, recs, counts, feats.ctypes.data_as(C.POINTER(C.c_float)))
    else:
        rc = lib.opd_detr_detect_frames(C.c_void_p(det.model), ptrs, _capi.OPD_MEM_HOST, 1, H, W, H, W, 0.5, recs, counts)
    assert rc == 0
for rep in range(2):
    print(f"C-ABI opd_detr_detect_frames           {best(lambda f: cabi_list(f, False)):.3f} ms")
    print(f"C-ABI opd_detr_detect_frames_features  {best(lambda f: cabi_list(f, True)):.3f} ms")
try:   # (device memory plumbing through torch, as in bench.py)
    import torch
    d = torch.from_numpy(frames[0]).cuda()
    d_out = torch.zeros((Q * 8 + 1,), dtype=torch.int32, device="cuda")   # (OPD_MEM_DEVICE: the outputs are device pointers too)
    torch.cuda.synchronize()
    def cabi_dev(_):
        rc = lib.opd_detr_detect_ragged(C.c_void_p(det.model), C.c_void_p(d.data_ptr()), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_DEVICE, 1, H, W, None, 0.5,
                                        hw.ctypes.data_as(C.c_void_p), C.cast(C.c_void_p(d_out.data_ptr()), C.POINTER(_capi.OpdDet)),
                                        C.cast(C.c_void_p(d_out[Q * 8:].data_ptr()), C.POINTER(C.c_int32)))
        assert rc == 0, lib.opd_last_error()
    print(f"C-ABI detect, frame and records in HBM {best(cabi_dev):.3f} ms")
except RuntimeError as e:
    print(f"C-ABI detect, frame and records in HBM: skipped ({e})")
cabi(sp)
print(f"_postprocess_batch                 {best(lambda f: det._postprocess_batch(recs, counts, Q)):.3f} ms   ({counts[0]} records)")
dets = det.detect(frames[0])
print(f"extract_features                   {best(lambda f: det.extract_features(f, dets)):.3f} ms   ({len(dets)} detections)")
