#!/usr/bin/env python3
"""Host-side anatomy of one bench step: detect call alone, + record/count fetch to host; 50 iterations each."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import _capi
from office_person_detection_vit_amd.frames import structured_frame
from office_person_detection_vit_amd.weights import DetrArch, ensure_weight_file
B, H, W = 8, 800, 1333
path = ensure_weight_file(os.environ.get("OPD_WEIGHT_CACHE", "/tmp/opd_weights"), DetrArch.resnet50(), 0, 1.0, "r50")
lib = _capi.load_library(test_hooks=True)
cfg = _capi.OpdConfig(struct_size=C.sizeof(_capi.OpdConfig), max_batch=B, max_height=H, max_width=W, flags=0)
h = C.c_void_p()
_capi.check(lib.opd_detr_create(C.byref(cfg), path.encode(), 0, C.byref(h)), "create")
frames = torch.from_numpy(np.stack([structured_frame(H, W, 1234 + i) for i in range(B)])).cuda()
rec = torch.zeros((B, 100, 8), dtype=torch.int32, device="cuda")
cnt = torch.zeros((B,), dtype=torch.int32, device="cuda")
hw = np.asarray([[H, W]] * B, dtype=np.int32)
hrec = (_capi.OpdDet * (B * 100))()
hcnt = (C.c_int32 * B)()
def dev():
    _capi.check(lib.opd_detr_detect(h, C.c_void_p(frames.data_ptr()), 0, _capi.OPD_MEM_DEVICE, B, H, W, 0.5, hw.ctypes.data_as(C.c_void_p),
                                    C.cast(C.c_void_p(rec.data_ptr()), C.POINTER(_capi.OpdDet)), C.cast(C.c_void_p(cnt.data_ptr()), C.POINTER(C.c_int32))), "detect")
def dev_cpu():
    dev(); return cnt.cpu(), rec.cpu()
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(f"detect (device in, device out): {t(dev):.3f} ms")
print(f"detect + .cpu() of counts and records: {t(dev_cpu):.3f} ms")
lib.opd_detr_destroy(h)
