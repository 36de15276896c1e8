import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
from office_person_detection_vit_amd import _capi
lib = _capi.load_library()
for s in [(8, 50, 84, 256, 256, 1), (8, 67, 120, 256, 256, 1), (7, 50, 84, 256, 256, 1)]:
    B, H, W, C1, C3, st = s
    t = []
    for dbg in (0, 8, 8 + 32, 8 + 32 + 64, 8 + 32 + 128, 8 + 32 + 64 + 128):
        us = (C.c_float * 4)()
        _capi.check(lib.opd_test_bench_btail(B, H, W, C1, C3, st, dbg, 20, us), "bench_btail")
        t.append(us[0])
    print(s, "full | 3x3 only | 3x3 no DMA | +no LDS reads | no DMA, no barrier | no DMA, no reads, no barrier:", " ".join(f"{v:8.1f}" for v in t), flush=True)
