#!/usr/bin/env python3
"""Where a step goes when several batches are in flight: concurrency analysis of a rocprofv3 kernel trace of the multi-stream bench.

The headline line is measured with three detector handles submitting to three streams; the per-kernel averages of such a trace
overlap, so their sum (3.7 ms per forward in round 4) says little about the 2.7-ms step.  This tool sweeps the start / end
timestamps of the steady-state window (the middle half of the trace) and prints
  * the share of wall time with 0, 1, 2, 3+ kernels in flight,
  * per kernel family: launches, mean duration, the time it ran ALONE and its attributed time (every instant split evenly among the
    kernels in flight: attributed times sum to the busy wall time), per forward.

Usage: overlap.py <kernel_trace.csv> [frames_per_forward_marker=postprocess_kernel]"""
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
    marker = sys.argv[2] if len(sys.argv) > 2 else "postprocess_kernel"
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows]
    ev.sort()
    t_lo, t_hi = ev[0][0], max(e for _, e, _ in ev)
    w0, w1 = t_lo + (t_hi - t_lo) // 4, t_hi - (t_hi - t_lo) // 4   # steady state: the middle half
    pts = []
    for i, (s, e, k) in enumerate(ev):
        s2, e2 = max(s, w0), min(e, w1)
        if e2 > s2:
            pts.append((s2, 1, i))
            pts.append((e2, -1, i))
    pts.sort()
    live = set()
    conc = defaultdict(float)
    alone = defaultdict(float)
    attr = defaultdict(float)
    prev = w0
    for t, d, i in pts:
        if t > prev:
            n = len(live)
            conc[min(n, 4)] += t - prev
            for j in live:
                attr[ev[j][2]] += (t - prev) / n
            if n == 1:
                alone[ev[next(iter(live))][2]] += t - prev
        prev = t
        if d > 0:
            live.add(i)
        else:
            live.discard(i)
    if w1 > prev:
        conc[0] += w1 - prev
    wall = w1 - w0
    forwards = sum(1 for s, e, k in ev if marker in k and w0 <= s < w1)
    print(f"window {wall / 1e6:.2f} ms, {forwards} forwards -> {wall / 1e3 / max(forwards, 1):.1f} us per forward")
    print("kernels in flight: " + "  ".join(f"{n}{'+' if n == 4 else ''}: {100 * conc[n] / wall:.1f} %" for n in range(5)))
    fam_n = defaultdict(int)
    fam_d = defaultdict(float)
    for s, e, k in ev:
        if w0 <= s < w1:
            fam_n[k] += 1
            fam_d[k] += e - s
    print(f"{'kernel':60s} {'n/fwd':>6s} {'mean us':>8s} {'dur/fwd':>8s} {'alone/fwd':>9s} {'attrib/fwd':>10s}")
    tot_attr = 0.0
    for k in sorted(attr, key=lambda k: -attr[k]):
        f = max(forwards, 1)
        tot_attr += attr[k] / f / 1e3
        print(f"{k[:60]:60s} {fam_n[k] / f:6.1f} {fam_d[k] / max(fam_n[k], 1) / 1e3:8.1f} {fam_d[k] / f / 1e3:8.1f} {alone[k] / f / 1e3:9.1f} {attr[k] / f / 1e3:10.1f}")
    print(f"{'sum of attributed time per forward (us)':60s} {'':6s} {'':8s} {sum(fam_d.values()) / max(forwards, 1) / 1e3:8.1f} {sum(alone.values()) / max(forwards, 1) / 1e3:9.1f} {tot_attr:10.1f}")


if __name__ == "__main__":
    main()
