#!/usr/bin/env python3
"""Isolated launch times of the fused decoder's five kernels (csrc/kernels_dec.hip + the key-split attention) at batch B x 100 queries,
Lk memory keys: us per launch, back to back on the null stream.  Usage: bench_dec.py [B] [Lk] [splits]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import _capi  # noqa: E402

lib = _capi.load_library(test_hooks=True)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Lk = int(sys.argv[2]) if len(sys.argv) > 2 else 1050
for splits in ([int(sys.argv[3])] if len(sys.argv) > 3 else [1, 2, 3, 4, 5]):
    us = (C.c_float * 5)()
    _capi.check(lib.opd_test_bench_dec(B, 100, Lk, 2048, splits, 200, us), "opd_test_bench_dec")
    print(f"B={B} Lk={Lk} splits={splits}: qkv {us[0]:.2f}  self {us[1]:.2f}  cross {us[2]:.2f}  cross_out {us[3]:.2f}  ffn {us[4]:.2f}  sum {sum(us):.2f} us", flush=True)
