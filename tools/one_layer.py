#!/usr/bin/env python3
"""Run ONE conv_gemm layer shape a few times (for rocprofv3 --pmc runs).  usage: one_layer.py B H W Cin N k stride res [variant [dbg]]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import _capi
a = [int(v) for v in sys.argv[1:9]]
variant = int(sys.argv[9], 0) if len(sys.argv) > 9 else 0   # flag word of opd_test_set_conv_flags
dbg = int(sys.argv[10]) if len(sys.argv) > 10 else 0
lib = _capi.load_library(test_hooks=True)
us = C.c_float()
_capi.check(lib.opd_test_bench_conv(*a, variant, dbg, 6, C.byref(us)), "bench_conv")
print("avg us", us.value)
