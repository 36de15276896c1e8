#!/usr/bin/env python3
"""GPU experiment: box drift against the golden vectors under each of the product's fusion switches (which fusion moved the
drift, and by how much?).  Every switch changes only fp32 summation order / the place of an fp16 rounding, so the spread of
the numbers below is the noise floor of the fp16-storage path at that resolution.  Usage: drift_toggles.py [out.txt]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from office_person_detection_vit_amd import _capi  # noqa: E402
from office_person_detection_vit_amd.detector import HipDetrDetector  # noqa: E402
from office_person_detection_vit_amd.frames import structured_frames  # noqa: E402
from office_person_detection_vit_amd.weights import DetrArch, ensure_weight_file  # noqa: E402

CASES = (("r101_mild_256x320", (3, 4, 23, 3)), ("r50_mild_256x320", (3, 4, 6, 3)), ("r50_mild_odd_203x333", (3, 4, 6, 3)),
         ("r50_mild_800x1333", (3, 4, 6, 3)))


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    lib = _capi.load_library(test_hooks=True)
    cache = os.environ.get("OPD_WEIGHT_CACHE", "/tmp/opd_weights")
    for tag, depths in CASES:
        g = np.load(os.path.join(ROOT, "tests", "golden", tag + ".npz"))
        frames = [structured_frames(1, int(h), int(w), seed=int(g["frame_seed"]) + i)[0] for i, (h, w) in enumerate(g["sizes"])]
        name = "r50" if depths == (3, 4, 6, 3) else "r" + "_".join(map(str, depths))
        path = ensure_weight_file(cache, DetrArch(depths=depths), 0, 1.0, name)
        rows = []
        for label, env, calls in (("default", {}, ()),
                                  ("stage-2 first block through the fused tail", {"OPD_DUAL_OVER_TAIL": "0"}, ()),
                                  ("shortcut fusions off", {}, (("opd_test_set_fuse_btail", 1),)),
                                  ("tails + shortcut fusions off", {}, (("opd_test_set_fuse_btail", 0),)),
                                  ("projection+LN, deep-K FFN-2, dec0 constant, heads LN off", {}, (("opd_test_set_fuse_gemm_ln", 0),))):
            for k, v in env.items():
                os.environ[k] = v
            det = HipDetrDetector(model_path=path, confidence_threshold=0.5, max_batch=len(frames), max_size=(800, 1333), resize=False)
            det.load_model()
            for k in env:
                del os.environ[k]
            for fn, val in calls:
                _capi.check(getattr(lib, fn)(C.c_void_p(det.model), val), fn)
            logits, boxes, enc = det.forward_raw(frames)
            det.close()
            rows.append((label, float(np.abs(boxes - g["pred_boxes"]).max()), float(np.abs(boxes - g["pred_boxes"]).mean())))
        print(tag, file=out)
        for label, mx, mean in rows:
            print(f"  {label:48s} |dbox| max {mx:.3e}  mean {mean:.3e}", file=out, flush=True)


if __name__ == "__main__":
    main()
