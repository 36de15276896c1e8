#!/usr/bin/env python3
"""CPU probe: does a smarter fp16 rounding of the folded backbone kernels (error diffusion along the reduction, so that the rounding
errors of one output channel sum to ~0 over neighbouring taps / channels) shrink the box drift that round-to-nearest causes on RAW
weights?  fp32 oracle vs the same oracle with every folded conv kernel replaced by its fp16 image.  Usage: wround_probe.py H W"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from office_person_detection_vit_amd.frames import structured_frames  # noqa: E402
from office_person_detection_vit_amd.weights import DetrArch, ensure_weight_file, load_safetensors  # noqa: E402
from oracle import detr_oracle as O  # noqa: E402


def diffuse(wf: np.ndarray, order: str) -> np.ndarray:
    """wf [Cout][Cin][KH][KW] fp32 -> fp16-representable fp32, error diffusion per output channel along the reduction."""
    co, ci, kh, kw = wf.shape
    if order == "chan_outer":      # the taps of one input channel are neighbours
        flat = wf.reshape(co, ci * kh * kw).copy()
    else:                          # device K order: (kh, kw, cin)
        flat = wf.transpose(0, 2, 3, 1).reshape(co, kh * kw * ci).copy()
    out = np.empty_like(flat)
    carry = np.zeros(co, np.float64)
    for k in range(flat.shape[1]):
        t = flat[:, k].astype(np.float64) + carry
        q = t.astype(np.float16).astype(np.float64)
        out[:, k] = q
        carry = t - q
    if order == "chan_outer":
        return out.reshape(co, ci, kh, kw).astype(np.float32)
    return out.reshape(co, kh, kw, ci).transpose(0, 3, 1, 2).astype(np.float32)


def fold_round(w, mode):
    w = dict(w)
    for k in list(w.keys()):
        if k.endswith(".convolution.weight") and k.startswith("model.backbone"):
            pre = k[: -len(".convolution.weight")] + ".normalization"
            scale = w[pre + ".weight"] * (w[pre + ".running_var"] + O.BN_EPS).rsqrt()
            wf = (w[k] * scale.view(-1, 1, 1, 1)).numpy()
            if mode == "rtn":
                q = wf.astype(np.float16).astype(np.float32)
            else:
                q = diffuse(wf, mode)
            w[k] = torch.from_numpy(q)
            w[pre + ".bias"] = w[pre + ".bias"] - w[pre + ".running_mean"] * scale
            w[pre + ".weight"] = torch.ones_like(scale)
            w[pre + ".running_mean"] = torch.zeros_like(scale)
            w[pre + ".running_var"] = torch.ones_like(scale) - O.BN_EPS
    return w


def main():
    H, W = int(sys.argv[1]), int(sys.argv[2])
    path = ensure_weight_file(os.environ.get("OPD_WEIGHT_CACHE", "/tmp/opd_weights"), DetrArch.resnet50(), 0, 1.0, "r50", device_exact=False)
    w = O.to_torch(load_safetensors(path))
    for seed in (5150, 1234):
        frames = structured_frames(2, H, W, seed=seed)
        pv, pm = O.preprocess(frames)
        lg0, bx0, mem0 = O.forward(w, pv, pm)
        for mode in ("rtn", "chan_outer", "device_k"):
            lg, bx, mem = O.forward(fold_round(w, mode), pv, pm)
            print(f"seed {seed} {mode:11s} |dbox| {float((bx - bx0).abs().max()):.2e} mean {float((bx - bx0).abs().mean()):.2e} |denc| {float((mem - mem0).abs().max()):.2e}", flush=True)


if __name__ == "__main__":
    main()
