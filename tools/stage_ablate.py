#!/usr/bin/env python3
"""Row-staged epilogue (dbg 0) against the 32-byte paired stores (dbg 32) of conv_gemm_dma_kernel on the store-heavy layers."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import _capi
SHAPES = [("s0b0.sc 64->256", 8, 200, 334, 64, 256, 1, 1, 0), ("s0b0.c0 64->64", 8, 200, 334, 64, 64, 1, 1, 0),
          ("s2.c2 256->1024 +res", 8, 50, 84, 256, 1024, 1, 1, 1), ("s2b0.sc 512->1024 s2", 8, 100, 167, 512, 1024, 1, 2, 0),
          ("s3.c2 512->2048 +res", 8, 25, 42, 512, 2048, 1, 1, 1), ("s3b0.sc 1024->2048 s2", 8, 50, 84, 1024, 2048, 1, 2, 0),
          ("s2.c0 1024->256", 8, 50, 84, 1024, 256, 1, 1, 0), ("s2.c1 3x3 256", 8, 50, 84, 256, 256, 3, 1, 0),
          ("s3.c0 2048->512", 8, 25, 42, 2048, 512, 1, 1, 0), ("s3.c1 3x3 512", 8, 25, 42, 512, 512, 3, 1, 0),
          ("s1b0.sc 256->512 s2", 8, 200, 334, 256, 512, 1, 2, 0), ("s1b0.c0 256->128", 8, 200, 334, 256, 128, 1, 1, 0),
          ("enc.qkv 256->768", 8400, 1, 1, 256, 768, 1, 1, 0), ("enc.fc1 256->2048", 8400, 1, 1, 256, 2048, 1, 1, 0),
          ("memkv 256->3072", 8400, 1, 1, 256, 3072, 1, 1, 0)]
lib = _capi.load_library(test_hooks=True)
us = C.c_float()
print(f"{'layer':26s} {'staged':>8s} {'paired':>8s}  (auto tile height)")
for name, B, H, W, Cin, N, k, st, res in SHAPES:
    t = []
    for dbg in (0, 32, 0, 32):
        _capi.check(lib.opd_test_bench_conv(B, H, W, Cin, N, k, st, res, 1, dbg, 20, C.byref(us)), "bench_conv")
        t.append(us.value)
    print(f"{name:26s} {min(t[0], t[2]):8.1f} {min(t[1], t[3]):8.1f}", flush=True)
