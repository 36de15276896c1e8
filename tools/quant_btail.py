#!/usr/bin/env python3
"""Wave-quantisation check of the stage-2 fused tail: 1044 workgroups (133 600 px) against exactly 1024 (131 072 px) and 1536."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import _capi
lib = _capi.load_library()
for (B, H, W) in [(8, 100, 167), (8, 128, 128), (8, 96, 128), (8, 128, 192), (8, 160, 128), (12, 128, 128)]:
    us = (C.c_float * 4)()
    _capi.check(lib.opd_test_bench_btail(B, H, W, 128, 128, 1, 32, 20, us), "bench")
    M = B * H * W
    print(f"{(B, H, W)}: M {M:7d}  wgs {(M + 127) // 128:5d}  {us[0]:7.1f} us  {us[0] / M * 1e3:.4f} ns/px", flush=True)
