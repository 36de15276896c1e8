#!/usr/bin/env python3
"""Per-kernel SQ picture of ONE forward from a rocprofv3 --pmc pass of bench.py (SQ block only, 8 counters):
MFMA pipe utilisation, what the waves do with their cycles, LDS bank conflicts.

  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY \
            SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline

Units (MI355X_MICROARCH.md): SQ_BUSY_CYCLES is summed over the 32 shader engines, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
(cycles), the wave counters are quad-cycles summed over waves.  MFMA utilisation = MFMA_BUSY / (1024 x kernel cycles) with
kernel cycles = SQ_BUSY_CYCLES / 32.  LDS: SQ_LDS_BANK_CONFLICT = extra LDS-array cycles spent on conflicts, SQ_LDS_IDX_ACTIVE = all LDS-array cycles
(both in cycles, summed over the 256 CUs), so their quotient is the share of the LDS array's work that conflicts cause, and IDX_ACTIVE /
(256 x kernel cycles) how busy the array is at all.  (Round 3 divided by SQ_ACTIVE_INST_LDS, which counts QUAD-cycles of waves with an LDS
instruction in flight: a different thing, and the "44-87 %" / "103 %" figures of that file overstate the conflicts; a csv with that
counter instead of SQ_LDS_IDX_ACTIVE is still accepted and labelled.)
usage: pmc_mfma.py <counter_collection.csv> [summary.json]"""
import collections, csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
ids = sorted({int(r["Dispatch_Id"]) for r in rows})
by = collections.defaultdict(dict)
name, dur = {}, {}
for r in rows:
    d = int(r["Dispatch_Id"])
    by[d][r["Counter_Name"]] = float(r["Counter_Value"])
    n = r["Kernel_Name"]
    name[d] = (n[n.find("::") + 2:] if "::" in n else n).split("(")[0]
    dur[d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
first = "preprocess_u8" if any("preprocess_u8" in name[d] for d in ids) else "stem_pool"   # (pre-processing inside the stem)
start = max(d for d in ids if first in name[d])   # last forward of the trace
agg = collections.OrderedDict()
for d in ids:
    if d < start:
        continue
    a = agg.setdefault(name[d], collections.defaultdict(float))
    a["n"] += 1; a["us"] += dur[d]
    for k, v in by[d].items():
        a[k] += v
idx = any("SQ_LDS_IDX_ACTIVE" in by[d] for d in ids)
ldshdr = "LDS conflict / LDS array cycles | LDS array busy" if idx else "LDS conflict cycles / ACTIVE_INST_LDS quad-cycles"
print(f"{'kernel':40s} {'n':>3s} {'us':>8s} {'MFMA util':>9s} {'active':>7s} {'wait':>6s} {'issue-stall':>11s}   {ldshdr}")
tot_mfma = tot_cyc = 0.0
for k, a in agg.items():
    cyc = a["SQ_BUSY_CYCLES"] / 32.0
    util = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc) if cyc else 0.0
    wc = a["SQ_WAVE_CYCLES"] or 1.0
    den = a["SQ_LDS_IDX_ACTIVE"] if idx else a["SQ_ACTIVE_INST_LDS"]
    lds = a["SQ_LDS_BANK_CONFLICT"] / den if den else 0.0
    busy = f" | {100 * a['SQ_LDS_IDX_ACTIVE'] / (256.0 * cyc):5.1f}%" if idx and cyc else ""
    tot_mfma += a["SQ_VALU_MFMA_BUSY_CYCLES"]; tot_cyc += cyc
    print(f"{k:40s} {int(a['n']):3d} {a['us']:8.1f} {100 * util:8.1f}% {100 * a['SQ_ACTIVE_INST_ANY'] / wc:6.1f}% {100 * a['SQ_WAIT_ANY'] / wc:5.1f}% "
          f"{100 * a['SQ_WAIT_INST_ANY'] / wc:10.1f}% {100 * lds:25.1f}%{busy}")
print(f"whole forward: MFMA pipes busy {100 * tot_mfma / (1024.0 * tot_cyc):.1f} % of the kernel cycles")
if len(sys.argv) > 2:   # machine-readable form for bench.py's roofline object
    import json
    out = {"note": "rocprofv3 --pmc SQ_* pass over one serial forward (tools/pmc_mfma.sh); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles)",
           "whole_forward_mfma_busy_pct": round(100 * tot_mfma / (1024.0 * tot_cyc), 2), "kernels": {}}
    for k, a in agg.items():
        cyc = a["SQ_BUSY_CYCLES"] / 32.0
        out["kernels"][k] = {"launches": int(a["n"]), "us": round(a["us"], 1), "mfma_busy_pct": round(100 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 2) if cyc else 0.0}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
