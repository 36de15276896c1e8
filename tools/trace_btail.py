#!/usr/bin/env python3
"""Where does a fused-tail workgroup spend its life?  One traced launch of btail_kernel<..., TRACE> per shape: shader-clock stamps at
the phase boundaries (prologue, 3x3 loop, every 64-channel chunk step, store retire), medians over the grid, first round vs later
rounds, launch span against the 100-MHz wall clock."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import _capi  # noqa: E402

SHAPES = [("stage-1 tail 64 -> 256 (+ reduce 64), residual through LDS-DMA (80 KiB of LDS: 2 workgroups per CU)", 8, 200, 334, 64, 64, 0),
          ("stage-1 tail 64 -> 256 (+ reduce 64), residual through VGPRs (48 KiB of LDS: 3 workgroups per CU)", 8, 200, 334, 64, 64, 16),
          ("stage-2 tail 128 -> 512 (+ reduce 128)", 8, 100, 167, 128, 128, 0)]


def main():
    lib = _capi.load_library(test_hooks=True)
    MAXW = 8192
    buf = (C.c_ulonglong * (MAXW * 16))()
    n = C.c_int()
    for name, B, H, W, C1, C3, dbg in SHAPES:
        _capi.check(lib.opd_test_trace_btail(B, H, W, C1, C3, dbg, buf, MAXW, C.byref(n)), "trace_btail")
        t = np.frombuffer(buf, dtype=np.uint64).reshape(MAXW, 16)[: n.value].astype(np.int64)
        nch = 4 * C1 // 64
        life = t[:, 5 + nch] - t[:, 1]
        d = np.diff(t[:, 1 : 6 + nch], axis=1)
        wall0 = (t[:, 0] - t[:, 0].min()) * 10.0
        span = (t[:, 15].max() - t[:, 0].min()) * 10.0
        first = wall0 <= 1000.0
        names = ["prologue", "3x3 loop"] + [f"chunk {j}" for j in range(nch)] + ["z + tail", "stores retire"]
        print(f"{name}: {n.value} workgroups, launch span {span / 1e3:.1f} us; life median {np.median(life):.0f} clk "
              f"(p10 {np.percentile(life, 10):.0f}, p90 {np.percentile(life, 90):.0f})")
        for lbl, sel in (("first round", first), ("later rounds", ~first)):
            if sel.sum():
                print(f"   {lbl:12s} ({int(sel.sum()):4d}): " + "  ".join(f"{names[i]} {np.median(d[sel, i]):.0f}" for i in range(len(names))), flush=True)


def main_stage3():
    """kernels_btail3.hip (eight waves, one workgroup per CU): stamps = entry, prologue, 3x3 loop, a1 exchange, every second chunk, z stores."""
    lib = _capi.load_library(test_hooks=True)
    MAXW = 8192
    buf = (C.c_ulonglong * (MAXW * 16))()
    n = C.c_int()
    for name, B, H, W in (("stage-3 tail 256 -> 1024 (+ reduce 256), batch 7 (230 workgroups: one round)", 7, 50, 84),
                          ("stage-3 tail, batch 8 (263 workgroups)", 8, 50, 84), ("stage-3 tail, r101 1066x1920 (503 workgroups)", 8, 67, 120)):
        _capi.check(lib.opd_test_trace_btail(B, H, W, 256, 256, 0, buf, MAXW, C.byref(n)), "trace_btail")
        t = np.frombuffer(buf, dtype=np.uint64).reshape(MAXW, 16)[: n.value].astype(np.int64)
        life = t[:, 14] - t[:, 1]
        d = np.diff(t[:, 1:15], axis=1)
        wall0 = (t[:, 0] - t[:, 0].min()) * 10.0
        span = (t[:, 15].max() - t[:, 0].min()) * 10.0
        first = wall0 <= 1000.0
        names = ["prologue", "3x3 loop", "a1 exchange"] + [f"chunks {2 * j},{2 * j + 1}" for j in range(8)] + ["z stores", "stores retire"]
        print(f"{name}: {n.value} workgroups, launch span {span / 1e3:.1f} us; life median {np.median(life):.0f} clk "
              f"(p10 {np.percentile(life, 10):.0f}, p90 {np.percentile(life, 90):.0f})")
        for lbl, sel in (("first round", first), ("later rounds", ~first)):
            if sel.sum():
                print(f"   {lbl:12s} ({int(sel.sum()):4d}): " + "  ".join(f"{names[i]} {np.median(d[sel, i]):.0f}" for i in range(len(names))), flush=True)


if __name__ == "__main__":
    main_stage3() if "--s3" in sys.argv else main()
