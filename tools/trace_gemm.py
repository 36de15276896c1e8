#!/usr/bin/env python3
"""Where does a workgroup of conv_gemm_dma_kernel spend its life?  One traced launch per layer shape (conv_gemm_dma_kernel<..., TRACE>):
per-workgroup shader-clock stamps at the phase boundaries, summarised as medians over the grid, plus the dispatch picture (start
times against the 100-MHz wall clock: how many rounds, how long the launch as a whole).  Usage: trace_gemm.py [filter]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from office_person_detection_vit_amd import _capi  # noqa: E402

SHAPES = [  # name, B, H, W, Cin, N, k, stride, residual
    ("enc.qkv 256->768", 8400, 1, 1, 256, 768, 1, 1, 0),
    ("enc.fc1 256->2048", 8400, 1, 1, 256, 2048, 1, 1, 0),
    ("enc.fc2 2048->256", 8400, 1, 1, 2048, 256, 1, 1, 0),
    ("enc.fc2 2048->256 split-K 4", 8400, 1, 1, 2048, 256, 1, 1, -4),
    ("proj 2048->256 split-K 4", 8400, 1, 1, 2048, 256, 1, 1, -4),
    ("s2.c0 1x1 1024->256", 8, 50, 84, 1024, 256, 1, 1, 0),
    ("s2.c1 3x3 256->256", 8, 50, 84, 256, 256, 3, 1, 0),
    ("s2.c2 1x1 256->1024 +res", 8, 50, 84, 256, 1024, 1, 1, 1),
    ("s3.c1 3x3 512->512", 8, 25, 42, 512, 512, 3, 1, 0),
    ("s3.c2 1x1 512->2048 +res", 8, 25, 42, 512, 2048, 1, 1, 1),
]
PH = ["prologue", "first tile", "k-loop", "epilogue issue", "stores retire"]


def main():
    lib = _capi.load_library(test_hooks=True)
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    MAXW = 8192
    buf = (C.c_ulonglong * (3 * MAXW * 8))()
    n = C.c_int()
    for name, B, H, W, Cin, N, k, st, res in SHAPES:
        if only and only not in name:
            continue
        split = -res if res < 0 else 0   # (negative "residual" column: split-K factor)
        _capi.check(lib.opd_test_trace_conv(B, H, W, Cin, N, k, st, max(res, 0), split, 3, buf, MAXW, C.byref(n)), "trace")
        t3 = np.frombuffer(buf, dtype=np.uint64).reshape(3, MAXW, 8).astype(np.int64)
        t3 = [x[x[:, 1] != 0] for x in t3]
        spans = [(x[:, 7].max() - x[:, 0].min()) * 10.0 for x in t3]                       # first entry .. last exit, ns
        gaps = [(t3[i + 1][:, 0].min() - t3[i][:, 7].max()) * 10.0 for i in range(2)]       # last exit .. next kernel's first entry
        period = (t3[2][:, 0].min() - t3[0][:, 0].min()) * 10.0 / 2
        t = t3[1]
        wgs = len(t)
        d = np.diff(t[:, 1:7], axis=1)   # shader clocks per phase
        life = t[:, 6] - t[:, 1]
        wall0 = (t[:, 0] - t[:, 0].min()) * 10.0   # ns
        clk_per_ns = np.median(life) / 1.0
        # launch span: first entry .. last end, in wall ns (end = start + life / f; f from the span of the slowest workgroup)
        print(f"{name}: {wgs} workgroups; life median {np.median(life):.0f} clk (p10 {np.percentile(life, 10):.0f}, p90 {np.percentile(life, 90):.0f})")
        print("   " + "  ".join(f"{PH[i]} {np.median(d[:, i]):.0f}" for i in range(5)) + "   (median clk)")
        first = wall0 <= 1000.0
        for lbl, sel in (("first round", first), ("later rounds", ~first)):
            if sel.sum():
                print(f"   {lbl:12s} ({int(sel.sum())}): " + "  ".join(f"{PH[i]} {np.median(d[sel, i]):.0f}" for i in range(5))
                      + f"   prologue p10 {np.percentile(d[sel, 0], 10):.0f} p90 {np.percentile(d[sel, 0], 90):.0f}")
        order = np.sort(wall0)
        late = order[order > 1000.0]
        print(f"   starts: {np.sum(order <= 1000.0)} within 1 us of the first, then {len(late)} later"
              + (f" (median +{np.median(late) / 1e3:.1f} us, last +{order[-1] / 1e3:.1f} us)" if len(late) else ""))
        print(f"   back-to-back launches: period {period / 1e3:.1f} us = first entry -> last exit {np.mean(spans) / 1e3:.1f} us"
              f" + last exit -> next kernel's first entry {np.mean(gaps) / 1e3:.1f} us", flush=True)


if __name__ == "__main__":
    main()
