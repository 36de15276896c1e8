"""Times the fused encoder FFN (enc_ffn_kernel) against the two launches it replaces (fc1 GEMM + deep-K ring) at the batch-8 encoder shape.
Run on the GPU box:  python tools/bench_enc_ffn.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import _capi  # noqa: E402

lib = _capi.load_library(test_hooks=True)
for M in (8400, 8 * 34 * 60, 1050, 4 * 34 * 60):
    us = C.c_float()
    _capi.check(lib.opd_test_bench_enc_ffn(M, 2048, 30, 0, 0, 0, C.byref(us)), "bench_enc_ffn")
    fused = us.value
    _capi.check(lib.opd_test_bench_gemm_ln(M, 2048, 1, 30, C.byref(us)), "bench_gemm_ln")
    ring = us.value
    _capi.check(lib.opd_test_bench_conv(M // 2 if M % 2 == 0 else M, 1, 2 if M % 2 == 0 else 1, 256, 2048, 1, 1, 0, 0, 0, 30, C.byref(us)), "bench_conv")
    fc1 = us.value
    _capi.check(lib.opd_test_bench_enc_ffn(M, 2048, 30, 0, 3, 0, C.byref(us)), "bench_enc_ffn")
    t3 = us.value
    _capi.check(lib.opd_test_bench_enc_ffn(M, 2048, 30, 0, 12, 0, C.byref(us)), "bench_enc_ffn")
    t12 = us.value
    _capi.check(lib.opd_test_bench_enc_ffn(M, 2048, 30, 0, 0, 1, C.byref(us)), "bench_enc_ffn")
    tf = us.value
    _capi.check(lib.opd_test_bench_gemm_ln(M, 256, 0, 30, C.byref(us)), "bench_gemm_ln")
    tos = us.value
    print(f"M = {M:6d}: fused FFN {fused:7.1f} us (with the output projection + LN in front {tf:6.1f}, against {tos:5.1f} for that launch alone; + q/k/v tail {t3:6.1f}, + 12-pass k/v tail {t12:6.1f}) | fc1 GEMM {fc1:6.1f} + deep-K ring {ring:6.1f} = {fc1 + ring:6.1f} us", flush=True)

if "--ablate" in sys.argv:
    print("\nablations at M = 8400 / 1050 (us): full | no MFMAs | no re-requests | no hidden-chunk traffic | no barriers | no MFMAs, no re-requests | only the wait / read / fence skeleton")
    for M in (8400, 1050):
        t = []
        for dbg in (0, 1, 2, 4, 8, 3, 15):
            us = C.c_float()
            _capi.check(lib.opd_test_bench_enc_ffn(M, 2048, 30, dbg, 0, 0, C.byref(us)), "bench_enc_ffn")
            t.append(us.value)
        print(f"M = {M:5d}: " + " | ".join(f"{v:6.1f}" for v in t), flush=True)
