#!/usr/bin/env python3
"""Run ONE fused bottleneck-tail shape a few times (for rocprofv3 --pmc runs).  usage: one_btail.py B H W C1 C3 stride [dbg]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import _capi
a = [int(v) for v in sys.argv[1:7]]
dbg = int(sys.argv[7]) if len(sys.argv) > 7 else 0
lib = _capi.load_library(test_hooks=True)
us = (C.c_float * 4)()
_capi.check(lib.opd_test_bench_btail(*a, dbg | 32, 6, us), "bench_btail")   # bit 32: no-op, skips the unfused launches
print("avg us", us[0])
