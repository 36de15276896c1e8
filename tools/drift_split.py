#!/usr/bin/env python3
"""Where does the fp16 path's box drift come from?  CPU experiment with the oracle's storage-emulation switches
(oracle/detr_oracle.py::forward(emulate=..., emulate_transformer=...)): the fp32 oracle against itself with (a) fp16 storage of
every backbone activation + folded fp16 conv kernels, (b) fp16 transformer operands (weights, GEMM inputs, P) with the fp32
residual stream the product keeps, (c) both = the product's numerics.  Usage: drift_split.py [r50|r101] H W [gain]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from office_person_detection_vit_amd.frames import structured_frames  # noqa: E402
from office_person_detection_vit_amd.weights import DetrArch, ensure_weight_file, load_safetensors  # noqa: E402
from oracle import detr_oracle as O  # noqa: E402


def main():
    arch_name, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    ga = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
    arch = DetrArch.resnet101() if arch_name == "r101" else DetrArch.resnet50()
    path = ensure_weight_file(os.environ.get("OPD_WEIGHT_CACHE", "/tmp/opd_weights"), arch, 0, ga, arch_name if ga == 1.0 else f"{arch_name}_g{ga}")
    w = O.to_torch(load_safetensors(path))
    frames = structured_frames(2, H, W, seed=1234)
    pv, pm = O.preprocess(frames)
    taps0 = {}
    lg0, bx0, mem0 = O.forward(w, pv, pm, taps=taps0)
    sm = lambda t: torch.softmax(t, -1)
    print(f"{arch_name} {H}x{W} gain {ga}: box spread over queries {float(bx0.std(dim=1).mean()):.3e}")
    for name, kw in (("backbone f16 storage", dict(emulate="f16")), ("transformer f16 operands", dict(emulate_transformer="f16")),
                     ("both (= product numerics)", dict(emulate="f16", emulate_transformer="f16"))):
        taps = {}
        lg, bx, mem = O.forward(w, pv, pm, taps=taps, **kw)
        rel = {k: float((taps[k] - taps0[k]).abs().max() / taps0[k].abs().max()) for k in ("stage3", "proj", "enc5", "hs") if k in taps}
        print(f"  {name:28s} |dbox| {float((bx - bx0).abs().max()):.2e}  |dprob| {float((sm(lg) - sm(lg0)).abs().max()):.2e}  "
              f"|denc| {float((mem - mem0).abs().max()):.2e}   rel.err " + " ".join(f"{k} {v:.1e}" for k, v in rel.items()), flush=True)


if __name__ == "__main__":
    main()
