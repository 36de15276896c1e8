import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from office_person_detection_vit_amd import HipDetrDetector
from office_person_detection_vit_amd.frames import structured_frames
from office_person_detection_vit_amd.weights import DetrArch, ensure_weight_file
mild = ensure_weight_file("/tmp/opd_weights", DetrArch.resnet50(), 0, 1.0, "r50")
sharp = ensure_weight_file("/tmp/opd_weights", DetrArch.resnet50(), 0, 2.0, "r50")
def mk(p, ms=(800, 1333), **kw):
    d = HipDetrDetector(model_path=p, max_batch=2, max_size=ms, resize=False, **kw); d.load_model(); return d
def diff(a, b): return max(float(np.abs(x - y).max()) for x, y in zip(a, b))
golden = [structured_frames(1, 256, 320, seed=1234 + i)[0] for i in range(2)]
probe = structured_frames(2, 256, 320, seed=4321)
scen = sys.argv[1]
A = mk(mild, use_graph=(scen != "eagerA"))
A.forward_raw(golden)
ref = A.forward_raw(probe)
if scen == "fresh_then_sharp" or scen == "eagerA":
    f = mk(mild); f.forward_raw(probe); f.close()
    B = mk(sharp); B.forward_raw(golden)
elif scen == "sharp_only":
    B = mk(sharp); B.forward_raw(golden)
elif scen == "sharp_created_only":
    B = mk(sharp)
elif scen == "fresh_kept_then_sharp":
    f = mk(mild); f.forward_raw(probe)
    B = mk(sharp); B.forward_raw(golden)
print(scen, "A after:", diff(A.forward_raw(probe), ref), diff(A.forward_raw(probe), ref))
