#!/usr/bin/env python3
"""Where does a dec_self_kernel workgroup spend its life?  Shader-clock stamps of wave 0 at the phase boundaries (warm caches):
0 entry | 1 loads + first weight pieces requested | 2 scores + softmax done | 3 P.V, exchange, barrier | 4 o-proj done | 5 LayerNorm done |
6 exchange + barrier | 7 cross-q projection done.  Usage: trace_dec.py [B]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import _capi  # noqa: E402

lib = _capi.load_library(test_hooks=True)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wgs = 7 * B
tr = np.zeros((wgs, 8), np.uint64)
_capi.check(lib.opd_test_trace_dec_self(B, 100, tr.ctypes.data_as(C.c_void_p)), "opd_test_trace_dec_self")
d = np.diff(tr.astype(np.int64), axis=1)
names = ["issue loads", "scores+softmax", "PV+exchange", "o-proj", "LayerNorm", "exchange", "cross-q"]
print(f"dec_self_kernel, {wgs} workgroups: median shader clocks per phase (min .. max); ~2.1-2.4 clocks per ns")
for i, n in enumerate(names):
    print(f"  {n:16s} {int(np.median(d[:, i])):7d}  ({int(d[:, i].min())} .. {int(d[:, i].max())})")
life = tr[:, 7].astype(np.int64) - tr[:, 0].astype(np.int64)
print(f"  life             {int(np.median(life)):7d}  ({int(life.min())} .. {int(life.max())});  first entry .. last exit: {int(tr[:, 7].max() - tr[:, 0].min())}")
