#!/usr/bin/env python3
"""GPU: where does an attention workgroup spend a key tile?  (attention_lazy_kernel<..., TRACE>, wave 0's shader clock)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from office_person_detection_vit_amd import _capi  # noqa: E402

lib = _capi.load_library(test_hooks=True)
rng = np.random.default_rng(0)
_p = lambda a: a.ctypes.data_as(C.c_void_p)
B, L = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 1050
qkv = (rng.standard_normal((B, L, 768)) * 1.2).astype(np.float16).view(np.uint16)
q = np.ascontiguousarray(qkv)
k = np.ascontiguousarray(np.roll(qkv, -256, axis=2))
v = np.ascontiguousarray(np.roll(qkv, -512, axis=2))
cap = 8192
tr = np.zeros((cap, 12), np.uint64)
n = C.c_int()
_capi.check(lib.opd_test_trace_attention(_p(q), _p(k), _p(v), B, 8, L, L, 768, 768, 32 ** -0.5, _p(tr), cap, C.byref(n)), "trace")
t = tr[:n.value].astype(np.float64)
t = t[t[:, 7] > 0]
per = t[:, :5] / t[:, 7:8]
names = ["issue next tile's LDS-DMA", "S = K.Q^T, max, branch", "exp2, cvt, P.V, row sum", "wait for the DMA", "barrier"]
print(f"{len(t)} workgroups, {int(t[0, 7])} key tiles each; median cycles of wave 0 per key tile:")
for i, nm in enumerate(names):
    print(f"  {nm:28s} {np.median(per[:, i]):8.0f}   (p10 {np.percentile(per[:, i], 10):6.0f}, p90 {np.percentile(per[:, i], 90):6.0f})")
print(f"  {'tile total':28s} {np.median(per.sum(1)):8.0f}     workgroup life {np.median(t[:, 6]):9.0f} cycles")

r0, r1 = t[:, 8], t[:, 9]
span = (r1.max() - r0.min()) / 100.0
print(f"launch span (first workgroup start -> last end, 100 MHz clock): {span:.2f} us; starts spread over {(r0.max() - r0.min()) / 100.0:.2f} us; "
      f"workgroup wall life median {np.median(r1 - r0) / 100.0:.2f} us (p10 {np.percentile(r1 - r0, 10) / 100.0:.2f}, p90 {np.percentile(r1 - r0, 90) / 100.0:.2f})")
clk = t[:, 6] / np.maximum(r1 - r0, 1) * 100.0
print(f"shader clock during the launch: median {np.median(clk):.0f} MHz")
hw = t[:, 11].astype(np.int64)
cu = ((hw >> 8) & 15) | (((hw >> 13) & 7) << 4) | ((t[:, 10].astype(np.int64) & 15) << 8)   # (cu_id, se_id, xcc) -> one key per CU
u, cnt = np.unique(cu, return_counts=True)
print(f"distinct CUs used: {len(u)}; workgroups per CU: min {cnt.min()}, max {cnt.max()}, histogram {np.bincount(cnt).tolist()}")
late = r0 - r0.min()
print(f"workgroups that started more than 2 us after the first: {(late > 200).sum()}")
