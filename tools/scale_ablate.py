#!/usr/bin/env python3
"""s2.c1 (3x3 256->256 at 50x84) at growing batch: full kernel and the no-DMA loop (LDS reads + MFMA + barriers only).
Separates per-workgroup throughput from fill / quantisation effects."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import _capi
lib = _capi.load_library(test_hooks=True)
us = C.c_float()
print(f"{'batch':>5s} {'wgs':>6s} {'full us':>8s} {'TF':>7s} {'no-DMA us':>9s} {'TF':>7s} {'no-MFMA us':>10s}")
for B in (4, 8, 10, 16, 32, 64):
    t = []
    for dbg in (0, 24, 1):
        _capi.check(lib.opd_test_bench_conv(B, 50, 84, 256, 256, 3, 1, 0, 0x501, dbg, 10, C.byref(us)), "bench_conv")
        t.append(us.value)
    fl = 2.0 * B * 50 * 84 * 256 * 2304
    wgs = ((B * 4200 + 159) // 160) * 2
    print(f"{B:5d} {wgs:6d} {t[0]:8.1f} {fl / t[0] / 1e6:7.1f} {t[1]:9.1f} {fl / t[1] / 1e6:7.1f} {t[2]:10.1f}", flush=True)
