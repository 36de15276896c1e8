#!/usr/bin/env python3
"""Timing ablations of the stage-3 fused tail (kernels_btail3.hip, BtailParams.dbg): where a launch's time goes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import _capi
lib = _capi.load_library(test_hooks=True)
for s in [(8, 50, 84, 256, 256, 1), (8, 67, 120, 256, 256, 1), (7, 50, 84, 256, 256, 1)]:
    B, H, W, C1, C3, st = s
    t = []
    for dbg in (0, 8, 2, 4, 6, 16, 22):
        us = (C.c_float * 4)()
        _capi.check(lib.opd_test_bench_btail(B, H, W, C1, C3, st, dbg, 20, us), "bench_btail")
        t.append(us[0])
    print(s, "full | 3x3 + a1 exchange only | no stores | no residual | neither | no chunk DMA | no chunk DMA, stores, residual:", " ".join(f"{v:8.1f}" for v in t), flush=True)
