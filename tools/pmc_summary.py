#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel name over the dispatches in counter_collection.csv files.
usage: pmc_summary.py <kernel-substring> <csv> [<csv> ...]"""
import csv, sys
from collections import defaultdict
pat = sys.argv[1]
acc = defaultdict(lambda: [0.0, 0])
dur = [0.0, 0]
for f in sys.argv[2:]:
    seen = set()
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        a = acc[r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
        if (f, r["Dispatch_Id"]) not in seen:
            seen.add((f, r["Dispatch_Id"]))
            dur[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; dur[1] += 1
print(f"kernel ~ '{pat}': {dur[1]} dispatches, avg {dur[0] / max(dur[1], 1):.1f} us (under the profiler)")
for k in sorted(acc):
    print(f"  {k:40s} {acc[k][0] / acc[k][1]:16.1f}")
