"""Times the fused bottleneck tail against the three launches it replaces, at the batch-8 800x1333 trunk shapes.
Run on the GPU box:  python tools/bench_btail.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import _capi  # noqa: E402

lib = _capi.load_library(test_hooks=True)
SHAPES = [  # B, H, W, C1, C3, stride
    (8, 200, 334, 64, 64, 1),     # s0b0 / s0b1 tails (-> next block's reduce)
    (8, 200, 334, 64, 128, 1),    # s0b2 tail (-> s1b0.c0)
    (8, 200, 334, 64, 0, 1),
    (8, 200, 334, 128, 128, 2),   # s1b0 tail (stride-2 3x3)
    (8, 100, 167, 128, 128, 1),   # s1b1 / s1b2 tails
    (8, 100, 167, 128, 0, 1),     # s1b3 tail
    (8, 50, 84, 256, 256, 1),     # stage-3 tails (kernels_btail3.hip)
    (8, 50, 84, 256, 0, 1),       # last stage-3 tail
    (8, 67, 120, 256, 256, 1),    # r101 at 1066x1920 (configs[3]): 22 of these
]
if "--s3" in sys.argv:
    SHAPES = SHAPES[-3:]
print(f"{'shape':>28s} {'fused us':>9s} {'c1':>7s} {'c2':>7s} {'c0n':>7s} {'unfused':>8s} {'TB/s fused':>10s}")
for s in SHAPES:
    B, H, W, C1, C3, st = s
    us = (C.c_float * 4)()
    _capi.check(lib.opd_test_bench_btail(B, H, W, C1, C3, st, 0, 20, us), "bench_btail")
    OH, OW = (H - 1) // st + 1, (W - 1) // st + 1
    M = B * OH * OW
    byt = (B * H * W * C1 + M * 4 * C1 * 2 + M * C3) * 2
    print(f"{str(s):>28s} {us[0]:9.1f} {us[1]:7.1f} {us[2]:7.1f} {us[3]:7.1f} {us[1] + us[2] + us[3]:8.1f} {byt / us[0] / 1e6:10.2f}", flush=True)

if "--ablate" in sys.argv:
    print("\nablations (us): full | no 3x3 loop | no stores | no residual | no stores+residual | 3x3 only | no 3x3, no stores, no residual | residual through VGPRs")
    for s in SHAPES:
        B, H, W, C1, C3, st = s
        t = []
        for dbg in (0, 1, 2, 4, 6, 8, 7, 16):
            us = (C.c_float * 4)()
            _capi.check(lib.opd_test_bench_btail(B, H, W, C1, C3, st, dbg, 20, us), "bench_btail")
            t.append(us[0])
        print(f"{str(s):>28s} " + " ".join(f"{v:8.1f}" for v in t), flush=True)

