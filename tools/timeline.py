#!/usr/bin/env python3
"""Timeline of the LAST forward in a rocprofv3 kernel trace (CSV) of bench.py: every kernel in launch order with its
start offset, duration and the idle gap before it; per-kernel-family and per-phase totals.

Usage: timeline.py <kernel_trace.csv> [--all]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a forward begins with the pre-processing kernel, or -- uint8 frames, pre-processing inside the stem -- with the stem itself
    starts = [i for i, r in enumerate(rows) if "preprocess" in r["Kernel_Name"]] or [i for i, r in enumerate(rows) if "stem_pool" in r["Kernel_Name"]]
    if not starts:
        sys.exit("no forward found in trace (neither a preprocess nor a stem kernel)")
    seg = rows[starts[-1]:]
    # cut at the last postprocess kernel
    ends = [i for i, r in enumerate(seg) if "postprocess" in r["Kernel_Name"]]
    if ends:
        seg = seg[:ends[-1] + 1]
    t0 = int(seg[0]["Start_Timestamp"])
    fam = defaultdict(lambda: [0, 0.0])
    prev_end = t0
    busy = gaps = 0.0
    show = "--all" in sys.argv
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        d, g = (e - s) / 1e3, (s - prev_end) / 1e3
        busy += d
        gaps += max(g, 0.0)
        k = short(r["Kernel_Name"])
        fam[k][0] += 1
        fam[k][1] += d
        if show:
            print(f"{(s - t0) / 1e3:9.1f} {d:8.1f} {g:7.1f}  {k}  wgs={int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)}")
        prev_end = max(prev_end, e)
    total = (prev_end - t0) / 1e3
    print(f"forward: {total:.1f} us wall, {busy:.1f} us in kernels, {gaps:.1f} us idle between kernels, {len(seg)} launches")
    for k, (n, d) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"{d:9.1f} us {n:4d} x {d / n:8.1f}  {k}")


if __name__ == "__main__":
    main()
