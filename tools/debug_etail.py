import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import _capi
lib = _capi.load_library()
def h(a):
    x = np.ascontiguousarray(a, np.float32).astype(np.float16)
    return x.astype(np.float32), np.ascontiguousarray(x.view(np.uint16))
p = lambda a: a.ctypes.data_as(C.c_void_p)
for M, F, inpl in ((640, 256, 0), (640, 256, 1), (33600, 1024, 1)):
    rng = np.random.default_rng(1)
    a1, a1b = h(np.abs(rng.standard_normal((M, 256))))
    res, resb = h(np.abs(rng.standard_normal((M, F))))
    wa, wab = h(rng.standard_normal((F, 256)) / 16)
    wb, wbb = h(rng.standard_normal((256, F)) / np.sqrt(F))
    ba = (0.1 * rng.standard_normal(F)).astype(np.float32); bb = (0.1 * rng.standard_normal(256)).astype(np.float32)
    hid = np.empty((M, F), np.uint16); z = np.empty((M, 256), np.uint16)
    _capi.check(lib.opd_test_etail(p(a1b), p(resb), p(wab), p(ba), p(wbb), p(bb), p(hid), p(z), M, F, inpl), "etail")
    want = np.maximum(a1.astype(np.float64) @ wa.T.astype(np.float64) + ba + res, 0)
    got = hid.view(np.float16).astype(np.float64)
    bad = ~(np.abs(got - want) <= 2e-3 + 1.1e-3 * np.abs(want))
    rows, cols = np.nonzero(bad)
    print(M, F, inpl, "bad", bad.sum(), "nan", np.isnan(got).sum(), "rows", np.unique(rows)[:20], "cols", np.unique(cols)[:40], "rows%64", np.unique(rows % 64)[:40])
    gz = z.view(np.float16).astype(np.float64)
    wz = np.maximum(np.nan_to_num(got) @ wb.T.astype(np.float64) + bb, 0)
    badz = ~(np.abs(gz - wz) <= 2e-3 + 1.1e-3 * np.abs(wz))
    r2, c2 = np.nonzero(badz)
    print("   z bad", badz.sum(), "rows", np.unique(r2)[:20], "cols", np.unique(c2)[:40])
