#!/usr/bin/env python3
"""Time individual conv_gemm layer shapes of the r50 @ 800x1333 batch-8 forward on the GPU (device-resident data), with
the kernel's timing ablations: full / loads only (no MFMA) / compute only (operands stay in LDS)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import _capi  # noqa: E402

SHAPES = [  # name, B, H, W, Cin, N, k, stride, residual
    ("s0.c1 3x3 64->64", 8, 200, 334, 64, 64, 3, 1, 0),
    ("s0.c2 1x1 64->256 +res", 8, 200, 334, 64, 256, 1, 1, 1),
    ("s0.c0 1x1 256->64", 8, 200, 334, 256, 64, 1, 1, 0),
    ("s1.c1 3x3 128->128", 8, 100, 167, 128, 128, 3, 1, 0),
    ("s1.c2 1x1 128->512 +res", 8, 100, 167, 128, 512, 1, 1, 1),
    ("s2.c0 1x1 1024->256", 8, 50, 84, 1024, 256, 1, 1, 0),
    ("s2.c1 3x3 256->256", 8, 50, 84, 256, 256, 3, 1, 0),
    ("s2.c2 1x1 256->1024 +res", 8, 50, 84, 256, 1024, 1, 1, 1),
    ("s3.c1 3x3 512->512", 8, 25, 42, 512, 512, 3, 1, 0),
    ("s3.c2 1x1 512->2048 +res", 8, 25, 42, 512, 2048, 1, 1, 1),
    ("enc.fc1 256->2048", 8400, 1, 1, 256, 2048, 1, 1, 0),
    ("enc.qkv 256->768", 8400, 1, 1, 256, 768, 1, 1, 0),
    ("enc.fc2 2048->256", 8400, 1, 1, 2048, 256, 1, 1, 0),
    ("dec.kv 256->3072", 8400, 1, 1, 256, 3072, 1, 1, 0),
]


def main():
    lib = _capi.load_library(test_hooks=True)
    variants = [int(v, 0) for v in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]   # flag words (opd_test_set_conv_flags)
    us = C.c_float()
    print(f"{'layer':28s} {'var':>3s} {'full us':>9s} {'TFLOP/s':>8s} {'loads-only':>10s} {'compute-only':>12s} {'no-stage full':>12s} {'loads B only':>12s} {'no-store':>9s} {'ns+comp':>9s}")
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, B, H, W, Cin, N, k, st, res in SHAPES:
        if only and only not in name:
            continue
        pad = k // 2
        OH, OW = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
        fl = 2.0 * B * OH * OW * N * k * k * Cin
        for v in variants:
            t = []
            for dbg in (0, 1, 2, 32, 1 + 8, 64, 64 + 2):
                _capi.check(lib.opd_test_bench_conv(B, H, W, Cin, N, k, st, res, v, dbg, 20, C.byref(us)), "bench_conv")
                t.append(us.value)
            print(f"{name:28s} {v:3d} {t[0]:9.1f} {fl / t[0] / 1e6:8.1f} {t[1]:10.1f} {t[2]:12.1f} {t[3]:11.1f} {t[4]:12.1f} {t[5]:9.1f} {t[6]:9.1f}")


if __name__ == "__main__":
    main()
