#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes of bench.py (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE; the counters cannot
share a pass, MI355X_MICROARCH.md §rocprofv3 PMC slots) into per-kernel HBM traffic of ONE forward.

Corrections per MI355X_MICROARCH.md §HBM: values are KiB; on gfx950 FETCH_SIZE reports exactly half of a wide coalesced
streaming read (confirmed here: preprocess_u8_kernel reads 25.6 MB of frames and reports 12.5 MB) -> doubled; WRITE_SIZE
is exact for 16-byte streaming stores (confirmed: the stem writes 273.2 MB and reports 273.2 MB).

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys


# the kernels bench.py times as the "implicit-GEMM family" (classes conv + linear of opd_detr_kernel_times)
GEMM_FAMILY = ("conv_gemm", "btail_kernel", "btail256_kernel", "gemm_ln256", "enc_ffn_kernel", "gemm_k256_kernel", "stem_pool")


def per_kernel(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    first = "preprocess_u8" if any("preprocess_u8" in r["Kernel_Name"] for r in rows) else "stem_pool"   # (pre-processing inside the stem)
    start = max(i for i, r in enumerate(rows) if first in r["Kernel_Name"])  # last forward in the trace
    agg = collections.OrderedDict()
    for r in rows[start:]:
        n = r["Kernel_Name"]
        n = (n[n.find("::") + 2:] if "::" in n else n).split("(")[0]
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"note": "HBM bytes of one forward (batch 8, 800x1333); FETCH_SIZE x2 (gfx950 correction), KiB -> bytes", "kernels": {}}
    gemm_launches, gemm_bytes = 0, 0.0
    for k, (n, kib) in fetch.items():
        rd = 2.0 * kib * 1024.0
        wr = write.get(k, [0, 0.0])[1] * 1024.0
        out["kernels"][k] = {"launches": n, "read_bytes": rd, "write_bytes": wr}
        if k.startswith(GEMM_FAMILY):
            gemm_launches += n
            gemm_bytes += rd + wr
    out["conv_gemm_family"] = {"launches": gemm_launches, "bytes_per_forward": gemm_bytes,
                               "bytes_per_launch": gemm_bytes / max(gemm_launches, 1)}
    out["total_bytes_per_forward"] = sum(v["read_bytes"] + v["write_bytes"] for v in out["kernels"].values())
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out["conv_gemm_family"]), out["total_bytes_per_forward"])


if __name__ == "__main__":
    main()
