#!/usr/bin/env python3
"""GPU debug: are two handles on the same weights bit-identical?  With the fused FFN / stage-3 tail kernels on and off."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from office_person_detection_vit_amd import HipDetrDetector, _capi  # noqa: E402
from office_person_detection_vit_amd.frames import structured_frames  # noqa: E402
from office_person_detection_vit_amd.weights import DetrArch, ensure_weight_file  # noqa: E402

path = ensure_weight_file("/tmp/opd_weights", DetrArch.resnet50(), 0, 1.0, "r50")
frames = structured_frames(2, 256, 320, seed=1212)
lib = _capi.load_library()
for flags in (3, 2, 1, 0):
    outs = []
    for rep in range(3):
        det = HipDetrDetector(model_path=path, max_batch=2, max_size=(800, 1333), resize=False, use_graph=False)
        det.load_model()
        lib.opd_test_set_fuse_ffn(C.c_void_p(det.model), flags)
        outs.append(det.forward_raw(frames))
        outs.append(det.forward_raw(frames))
        det.close()
    ref = outs[0]
    diffs = [max(float(np.abs(o[i] - ref[i]).max()) for i in range(3)) for o in outs[1:]]
    print(f"fuse flags {flags} (bit0 ffn, bit1 etail): max |diff| vs first run over 5 more runs: {diffs}", flush=True)
