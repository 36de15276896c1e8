#!/usr/bin/env python3
"""conv_gemm kernel generations on the stage-3/4 and transformer layer shapes: v2 (default), v2 without buffer-descriptor
staging, 4-stage x 32-deep ring, 3-stage x 64-deep ring.  usage: sweep_variants.py [variant ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import _capi
SHAPES = [("s2.c0 1024->256", 8, 50, 84, 1024, 256, 1, 1, 0), ("s2.c1 3x3 256", 8, 50, 84, 256, 256, 3, 1, 0),
          ("s2.c2 256->1024 +res", 8, 50, 84, 256, 1024, 1, 1, 1), ("s2b0.c0 512->256", 8, 100, 167, 512, 256, 1, 1, 0),
          ("s3.c0 2048->512", 8, 25, 42, 2048, 512, 1, 1, 0), ("s3.c1 3x3 512", 8, 25, 42, 512, 512, 3, 1, 0),
          ("s3.c2 512->2048 +res", 8, 25, 42, 512, 2048, 1, 1, 1), ("enc.fc1 256->2048", 8400, 1, 1, 256, 2048, 1, 1, 0)]
variants = [int(v, 0) for v in sys.argv[1:]] or [1, 33, 3, 2, 4]
lib = _capi.load_library()
us = C.c_float()
print(f"{'layer':24s} " + " ".join(f"{v:>8d}" for v in variants))
for name, B, H, W, Cin, N, k, st, res in SHAPES:
    t = []
    for v in variants:
        _capi.check(lib.opd_test_bench_conv(B, H, W, Cin, N, k, st, res, v, 0, 20, C.byref(us)), "bench_conv")
        t.append(us.value)
    print(f"{name:24s} " + " ".join(f"{x:8.1f}" for x in t), flush=True)
