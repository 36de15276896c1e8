#!/usr/bin/env python3
"""GPU: time Linear(K -> 256) + residual + LayerNorm as one launch: the k-loop / one-shot kernels and the deep-K row-owner ring."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from office_person_detection_vit_amd import _capi  # noqa: E402

lib = _capi.load_library(test_hooks=True)
for M, K in ((8400, 2048), (16320, 2048), (8400, 256), (800, 2048), (800, 256)):
    line = f"M {M:6d} K {K:5d}:"
    for deep in (0, 1):
        us = C.c_float()
        _capi.check(lib.opd_test_bench_gemm_ln(M, K, deep, 30, C.byref(us)), "bench")
        line += f"   {'ring' if deep else 'k-loop / one-shot'} {us.value:7.2f} us"
    print(line, flush=True)
