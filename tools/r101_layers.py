import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import _capi
lib = _capi.load_library(test_hooks=True)
SHAPES = [("r101 s2.c0 1024->256", 8, 67, 120, 1024, 256, 1, 1, 0), ("r101 s2.c1 3x3 256", 8, 67, 120, 256, 256, 3, 1, 0), ("r101 s2.c2 256->1024 +res", 8, 67, 120, 256, 1024, 1, 1, 1)]
us = C.c_float()
print(f"{'layer':28s} {'auto':>8s} {'128':>8s} {'160':>8s} {'192':>8s}  TFLOP/s(best)")
for name, B, H, W, Cin, N, k, st, res in SHAPES:
    pad = k // 2
    OH, OW = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    fl = 2.0 * B * OH * OW * N * k * k * Cin
    t = []
    for v in (1, 0x401, 0x501, 0x601):
        _capi.check(lib.opd_test_bench_conv(B, H, W, Cin, N, k, st, res, v, 0, 20, C.byref(us)), "bench_conv")
        t.append(us.value)
    print(f"{name:28s} {t[0]:8.1f} {t[1]:8.1f} {t[2]:8.1f} {t[3]:8.1f}   {fl / min(t) / 1e6:7.1f}", flush=True)
