#!/usr/bin/env python3
"""How much of a conv_gemm launch is the A (activation) or B (weight) tile DMA?  full | no A DMA | no B DMA | neither | no MFMA"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import _capi
SHAPES = [("s2.c1 3x3 256", 8, 50, 84, 256, 256, 3, 1, 0), ("s3.c1 3x3 512", 8, 25, 42, 512, 512, 3, 1, 0),
          ("s2.c0 1024->256", 8, 50, 84, 1024, 256, 1, 1, 0), ("s3.c0 2048->512", 8, 25, 42, 2048, 512, 1, 1, 0),
          ("s2.c2 256->1024 +res", 8, 50, 84, 256, 1024, 1, 1, 1)]
lib = _capi.load_library(test_hooks=True)
us = C.c_float()
print(f"{'layer':24s} {'full':>8s} {'no A':>8s} {'no B':>8s} {'no A,B':>8s} {'no MFMA':>8s}")
for name, B, H, W, Cin, N, k, st, res in SHAPES:
    t = []
    for dbg in (0, 8, 16, 24, 1):
        _capi.check(lib.opd_test_bench_conv(B, H, W, Cin, N, k, st, res, 1, dbg, 20, C.byref(us)), "bench_conv")
        t.append(us.value)
    print(f"{name:24s} " + " ".join(f"{x:8.1f}" for x in t), flush=True)
