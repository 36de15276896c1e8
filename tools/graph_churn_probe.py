#!/usr/bin/env python3
"""GPU diagnosis of the graph-replay-after-handle-churn corruption.

One scenario per process (argv[1]); the epoch guard of run_forward is switched OFF so that a graph captured before the
churn is replayed after it.  With taps on, every launch of the forward is followed by a checksum launch of its output
(captured into the graph), and the first tap that differs between the capture run and the replay names the first
diverging kernel.
  scenarios: handles | handles_taps | handles_pageable | handles_nodec0 | free_only | create_only | torch_churn | guard_on
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from office_person_detection_vit_amd import HipDetrDetector, _capi  # noqa: E402
from office_person_detection_vit_amd.frames import structured_frames  # noqa: E402
from office_person_detection_vit_amd.weights import DetrArch, ensure_weight_file  # noqa: E402

scen = sys.argv[1]
lib = _capi.load_library(test_hooks=True)
lib.opd_test_set_graph_guard(1 if scen == "guard_on" else 0)
mild = ensure_weight_file("/tmp/opd_weights", DetrArch.resnet50(), 0, 1.0, "r50")
sharp = ensure_weight_file("/tmp/opd_weights", DetrArch.resnet50(), 0, 2.0, "r50")


def mk(p, **kw):
    d = HipDetrDetector(model_path=p, max_batch=2, max_size=(800, 1333), resize=False, **kw)
    d.load_model()
    return d


def diff(a, b):
    return [float(np.abs(x - y).max()) for x, y in zip(a, b)]


def taps(det):
    sums = (C.c_ulonglong * 512)()
    names = C.create_string_buffer(1 << 16)
    n = lib.opd_test_read_taps(C.c_void_p(det.model), sums, 512, names, len(names))
    return [(nm, int(sums[i])) for i, nm in enumerate(names.value.decode().split("\n")[:n])]


golden = [structured_frames(1, 256, 320, seed=1234 + i)[0] for i in range(2)]
probe = structured_frames(2, 256, 320, seed=4321)
A = mk(mild, pinned_staging=(scen != "handles_pageable"))
if scen == "handles_nodec0":
    lib.opd_test_set_fuse_gemm_ln(C.c_void_p(A.model), 0)   # long way round: memset nodes, unfused LN launches in the graph
use_taps = scen.endswith("_taps")
if use_taps:
    lib.opd_test_set_taps(C.c_void_p(A.model), 1)
eager = A.forward_raw(probe)          # eager
A.forward_raw(golden)                 # capture + first launch (other frames)
ref = A.forward_raw(probe)            # replay
t_ref = taps(A) if use_taps else None
print(scen, "replay vs eager before churn:", diff(ref, eager), flush=True)
if scen in ("handles", "handles_taps", "handles_pageable", "handles_nodec0", "guard_on"):
    f = mk(mild); f.forward_raw(probe); f.close()
    B = mk(sharp); B.forward_raw(golden)
elif scen == "free_only":
    f = mk(mild); f.forward_raw(probe); f.close()
elif scen == "create_only":
    B = mk(sharp); B.forward_raw(golden)
elif scen == "torch_churn":
    import torch
    x = torch.full((1 << 30,), float("nan"), device="cuda"); torch.cuda.synchronize(); del x
    torch.cuda.empty_cache()
    y = torch.full((3 << 28,), float("nan"), device="cuda"); torch.cuda.synchronize()
for rep in range(3):
    got = A.forward_raw(probe)
    print(scen, f"replay {rep} after churn vs before:", diff(got, ref), flush=True)
    if use_taps:
        t = taps(A)
        bad = [(i, a[0]) for i, (a, b) in enumerate(zip(t, t_ref)) if a[1] != b[1]]
        print(scen, f"  taps: {len(t)}; differing: {len(bad)}; first: {bad[:6]}", flush=True)
        if bad:
            i0 = bad[0][0]
            print(scen, "  neighbourhood:", [(i, t[i][0], t[i][1] == t_ref[i][1]) for i in range(max(0, i0 - 3), min(len(t), i0 + 4))], flush=True)
