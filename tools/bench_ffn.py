"""Times kernels_ffn.hip (fused feed-forward block / stage-3 expand tail) with its timing ablations.  Run on the GPU box."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import _capi  # noqa: E402

lib = _capi.load_library()
CASES = [("FFN   M=8400  F=2048", 8400, 2048, 0), ("ETAIL M=33600 F=1024", 33600, 1024, 1), ("FFN   M=16320 F=2048 (r101 1080p)", 16320, 2048, 0)]
DBG = [(0, "default (all pieces during GEMM a)"), (8, "pieces spread over both GEMMs"), (4, "all pieces in front of the MFMAs"),
       (1, "no ds_read / MFMA (staging only)"), (2, "no staging after chunk 1 (compute only)"), (3, "neither"),
       (16, "two chunks only"), (16 + 32, "two chunks, no epilogue"), (3 + 32, "neither, no epilogue"), (32, "no epilogue")]
for name, M, F, et in CASES:
    print(name)
    for dbg, what in DBG:
        us = C.c_float()
        _capi.check(lib.opd_test_bench_ffn(M, F, et, dbg, 30, C.byref(us)), "bench_ffn")
        fl = 4.0 * M * 256 * F
        print(f"    dbg {dbg:2d} {what:44s} {us.value:8.1f} us   {fl / us.value / 1e6:7.1f} TFLOP/s", flush=True)
