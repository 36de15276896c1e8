#!/usr/bin/env python3
"""Join a rocprofv3 kernel trace (CSV) of bench.py with the layer list of the detect path: per-launch duration,
algorithmic TFLOP/s and GB/s (unfused minimum bytes) for the LAST step in the trace.

Usage: trace_layers.py <kernel_trace.csv> [B H W]"""
import csv
import sys


def down2(n):
    return (n - 1) // 2 + 1


def layers(B=8, H=800, W=1333, depths=(3, 4, 6, 3)):
    """(name, M, N, K, read bytes, write bytes) of every conv_gemm launch of one forward, in launch order."""
    L = []
    H1, W1 = down2(H), down2(W)
    H2, W2 = down2(H1), down2(W1)
    L.append(("stem7x7", B * H1 * W1, 64, 147, B * H * W * 4 * 2, B * H1 * W1 * 64 * 2))
    cin, h, w = 64, H2, W2
    for s, (d, cout) in enumerate(zip(depths, (256, 512, 1024, 2048))):
        mid = cout // 4
        for l in range(d):
            st = 2 if (l == 0 and s > 0) else 1
            oh, ow = (down2(h), down2(w)) if st == 2 else (h, w)
            if l == 0:
                L.append((f"s{s}b{l}.sc", B * oh * ow, cout, cin, B * h * w * cin * 2 // (st * st), B * oh * ow * cout * 2))
            L.append((f"s{s}b{l}.c0", B * h * w, mid, cin, B * h * w * cin * 2, B * h * w * mid * 2))
            L.append((f"s{s}b{l}.c1", B * oh * ow, mid, 9 * mid, B * h * w * mid * 2, B * oh * ow * mid * 2))
            L.append((f"s{s}b{l}.c2", B * oh * ow, cout, mid, B * oh * ow * mid * 2 + B * oh * ow * cout * 2, B * oh * ow * cout * 2))
            cin, h, w = cout, oh, ow
    M = B * h * w
    L.append(("proj", M, 256, 2048, M * 2048 * 2, M * 256 * 6))
    for i in range(6):
        L += [(f"enc{i}.qkv", M, 768, 256, M * 256 * 2, M * 768 * 2), (f"enc{i}.o", M, 256, 256, M * 256 * 2 + M * 256 * 4, M * 256 * 4),
              (f"enc{i}.fc1", M, 2048, 256, M * 256 * 2, M * 2048 * 2), (f"enc{i}.fc2", M, 256, 2048, M * 2048 * 2 + M * 256 * 4, M * 256 * 4)]
    L.append(("memkv", M, 3072, 256, M * 256 * 2, M * 3072 * 2))
    Md = B * 100
    for i in range(6):
        L += [(f"dec{i}.qkv", Md, 768, 256, 0, 0), (f"dec{i}.so", Md, 256, 256, 0, 0), (f"dec{i}.cq", Md, 256, 256, 0, 0),
              (f"dec{i}.co", Md, 256, 256, 0, 0), (f"dec{i}.fc1", Md, 2048, 256, 0, 0), (f"dec{i}.fc2", Md, 256, 2048, 0, 0)]
    return L


def main():
    f = sys.argv[1]
    B, H, W = (int(v) for v in sys.argv[2:5]) if len(sys.argv) >= 5 else (8, 800, 1333)
    rows = [r for r in csv.DictReader(open(f)) if "conv_gemm" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    L = layers(B, H, W)
    last = rows[-len(L):]
    tot = 0.0
    print(f"{'layer':12s} {'M':>8s} {'N':>5s} {'K':>5s} {'us':>8s} {'TFLOP/s':>8s} {'GB/s':>7s} {'wgs':>6s} kernel")
    for (name, M, N, K, rb, wb), r in zip(L, last):
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += us
        fl = 2.0 * M * N * K
        by = rb + wb + N * K * 2
        kn = r["Kernel_Name"]
        tag = kn[kn.find("<"):kn.find(">") + 1] if "<" in kn else kn[:24]
        print(f"{name:12s} {M:8d} {N:5d} {K:5d} {us:8.1f} {fl / us / 1e6:8.1f} {by / us / 1e3 if rb else 0:7.0f} "
              f"{int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):6d} {tag}")
    print("total conv_gemm us per step:", round(tot, 1))


if __name__ == "__main__":
    main()
