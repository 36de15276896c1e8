#!/usr/bin/env python3
"""bench.py at several batch sizes: per-frame time of each trunk stage (is a stage cheaper per frame when its
inter-kernel tensors fit the 256 MB Infinity Cache?).  usage: stage_by_batch.py 8 4 2"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for b in sys.argv[1:]:
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--batch", b],
                         capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print(f"batch {b}: {d['value']:8.1f} fps  {d['ms_per_step']:.3f} ms/step  per-frame stage us:",
          [round(1000 * x / int(b), 1) for x in d["stage_ms"]], flush=True)
