import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from office_person_detection_vit_amd import HipDetrDetector, _capi
from office_person_detection_vit_amd.frames import structured_frames
from office_person_detection_vit_amd.weights import DetrArch, ensure_weight_file, load_safetensors, save_safetensors
src = open(os.path.join(ROOT, "tests", "test_detector_gpu.py")).read()
ns = {}
exec(src[src.index("def _to_4x_name"):src.index("def test_checkpoint_with_4x")], ns)
path = ensure_weight_file("/tmp/opd_weights", DetrArch.resnet50(), 0, 1.0, "r50")
w = load_safetensors(path)
frames = structured_frames(2, 256, 320, seed=1212)
def run(p):
    d = HipDetrDetector(model_path=p, max_batch=2, max_size=(800, 1333), resize=False)
    d.load_model(); o = d.forward_raw(frames); d.close(); return o
ref = run(path)
def variant(name, fn):
    p = f"/tmp/opd_weights/dbg_{name}.safetensors"
    save_safetensors({fn(k): v for k, v in w.items()}, p)
    o = run(p)
    print(name, [float(np.abs(a - b).max()) for a, b in zip(o, ref)], flush=True)
variant("same", lambda k: k)
variant("all4x", ns["_to_4x_name"])
variant("only_backbone", lambda k: ns["_to_4x_name"](k) if k.startswith("model.backbone") else k)
variant("only_transformer", lambda k: ns["_to_4x_name"](k) if not k.startswith("model.backbone") else k)
variant("only_outproj", lambda k: k.replace(".o_proj.", ".out_proj."))
variant("only_fc", lambda k: k.replace(".mlp.fc1.", ".fc1.").replace(".mlp.fc2.", ".fc2.") if k.startswith(("model.encoder.layers.", "model.decoder.layers.")) else k)
variant("reversed_order", lambda k: k)   # placeholder (dict order identical)
