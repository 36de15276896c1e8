#!/usr/bin/env python3
"""Wave-quantisation check of the stage-2 fused tail: 1044 workgroups (133 600 px) against exactly 1024 (131 072 px) and 1536."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import _capi
lib = _capi.load_library(test_hooks=True)
for (C1, C3, shapes) in [(128, 128, [(8, 100, 167), (8, 99, 167), (8, 98, 167), (8, 97, 167), (8, 128, 128), (8, 96, 128), (8, 128, 192), (8, 160, 128)]),
                         (64, 64, [(8, 200, 334), (8, 256, 256), (8, 256, 224), (8, 256, 288)])]:
    for (B, H, W) in shapes:
        us = (C.c_float * 4)()
        _capi.check(lib.opd_test_bench_btail(B, H, W, C1, C3, 1, 32, 20, us), "bench")
        M = B * H * W
        print(f"C1 {C1:3d} {(B, H, W)}: M {M:7d}  wgs {(M + 127) // 128:5d}  {us[0]:7.1f} us  {us[0] / M * 1e3:.4f} ns/px", flush=True)
