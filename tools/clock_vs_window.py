#!/usr/bin/env python3
"""GPU: why does a 200-step window report fewer frames/s than a 20-step one?  Runs bench.py at several --steps values and samples the
shader clock (sysfs pp_dpm_sclk / rocm-smi) and the board power every 20 ms meanwhile.  usage: clock_vs_window.py [K ...]"""
import glob
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ks = [int(v) for v in sys.argv[1:]] or [20, 200, 1000, 20, 3000]


def read_sclk():
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for line in open(f):
                if "*" in line:
                    return float(re.search(r"(\d+)Mhz", line).group(1))
        except OSError:
            pass
    return None


def read_power():
    for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
        try:
            return float(open(f).read()) / 1e6
        except (OSError, ValueError):
            pass
    return None


print("sclk readable:", read_sclk(), "MHz; power readable:", read_power(), "W", flush=True)
for k in ks:
    samples = []
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append((time.perf_counter(), read_sclk(), read_power()))
            time.sleep(0.02)
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(k), "--warmup", "5", "--no-cpu-baseline", "--serial-steps", "0"],
                         capture_output=True, text=True)
    t1 = time.perf_counter()
    stop.set()
    th.join()
    d = json.loads(out.stdout.strip().splitlines()[-1])
    window = d["ms_per_step"] * k / 1e3
    # the timed window is the last `window` seconds before the process starts shutting down: take the samples of the busiest stretch
    clk = [s[1] for s in samples if s[1]]
    pw = [s[2] for s in samples if s[2]]
    tail = max(3, int(window / 0.02) + 2)
    fmt = lambda v: "n/a" if not v else f"min {min(v):.0f} median {sorted(v)[len(v) // 2]:.0f} max {max(v):.0f}"
    print(f"K = {k:5d}: {d['value']:8.1f} frames/s, {d['ms_per_step']:.3f} ms/step, window {window * 1e3:8.1f} ms | sclk over the run: {fmt(clk)}; "
          f"last {tail} samples: {fmt(clk[-tail - 25:-25] if len(clk) > tail + 25 else clk)} | power {fmt(pw)}", flush=True)
