#!/usr/bin/env python3
"""Tile-height sweep (128 / 160 / 192 rows, auto) of the conv_gemm LDS-DMA kernel over the layer shapes that are still
separate launches in the r50 @ 800x1333 batch-8 forward.  Feeds the `pick_mt` heuristic in kernels_gemm.hip."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import _capi  # noqa: E402

SHAPES = [  # name, B, H, W, Cin, N, k, stride, residual
    ("s0b0.sc 64->256", 8, 200, 334, 64, 256, 1, 1, 0),
    ("s0b0.c0 64->64", 8, 200, 334, 64, 64, 1, 1, 0),
    ("s1b0.sc 256->512 s2", 8, 200, 334, 256, 512, 1, 2, 0),
    ("s2b0.sc 512->1024 s2", 8, 100, 167, 512, 1024, 1, 2, 0),
    ("s2b0.c0 512->256", 8, 100, 167, 512, 256, 1, 1, 0),
    ("s2b0.c1 3x3 s2", 8, 100, 167, 256, 256, 3, 2, 0),
    ("s2.c0 1024->256", 8, 50, 84, 1024, 256, 1, 1, 0),
    ("s2.c1 3x3 256", 8, 50, 84, 256, 256, 3, 1, 0),
    ("s2.c2 256->1024 +res", 8, 50, 84, 256, 1024, 1, 1, 1),
    ("s3b0.sc 1024->2048 s2", 8, 50, 84, 1024, 2048, 1, 2, 0),
    ("s3b0.c0 1024->512", 8, 50, 84, 1024, 512, 1, 1, 0),
    ("s3b0.c1 3x3 s2", 8, 50, 84, 512, 512, 3, 2, 0),
    ("s3.c0 2048->512", 8, 25, 42, 2048, 512, 1, 1, 0),
    ("s3.c1 3x3 512", 8, 25, 42, 512, 512, 3, 1, 0),
    ("s3.c2 512->2048 +res", 8, 25, 42, 512, 2048, 1, 1, 1),
    ("enc.qkv 256->768", 8400, 1, 1, 256, 768, 1, 1, 0),
    ("enc.fc1 256->2048", 8400, 1, 1, 256, 2048, 1, 1, 0),
    ("memkv 256->3072", 8400, 1, 1, 256, 3072, 1, 1, 0),
]
lib = _capi.load_library(test_hooks=True)
us = C.c_float()
print(f"{'layer':26s} {'auto':>8s} {'128':>8s} {'160':>8s} {'192':>8s}   TFLOP/s(best)")
for name, B, H, W, Cin, N, k, st, res in SHAPES:
    pad = k // 2
    OH, OW = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    fl = 2.0 * B * OH * OW * N * k * k * Cin
    t = []
    for v in (0, 0x400, 0x500, 0x600):   # flag word of opd_test_set_conv_flags: bits 8-10 force the tile height
        _capi.check(lib.opd_test_bench_conv(B, H, W, Cin, N, k, st, res, v, 0, 20, C.byref(us)), "bench_conv")
        t.append(us.value)
    print(f"{name:26s} {t[0]:8.1f} {t[1]:8.1f} {t[2]:8.1f} {t[3]:8.1f}   {fl / min(t) / 1e6:7.1f}", flush=True)
