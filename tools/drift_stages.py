#!/usr/bin/env python3
"""CPU: per-stage split of the gain-2 ('sharp') drift: fp16 storage emulated only in selected backbone stages, and with an fp32
residual stream (block outputs unrounded, conv operands rounded) in stages >= S.
usage: drift_stages.py GAIN H W"""
import os, sys, numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from office_person_detection_vit_amd.frames import structured_frames
from office_person_detection_vit_amd.weights import DetrArch, ensure_weight_file, load_safetensors
from oracle import detr_oracle as O
torch.set_num_threads(8)
ga = float(sys.argv[1]); H, W = int(sys.argv[2]), int(sys.argv[3])
path = ensure_weight_file(os.environ.get("OPD_WEIGHT_CACHE", "/tmp/opd_weights"), DetrArch(), 0, ga, "r50" if ga == 1.0 else f"r50_g{ga}")
w0 = O.to_torch(load_safetensors(path))
frames = structured_frames(2, H, W, seed=1234)
pv, pm = O.preprocess(frames)
lg0, bx0, mem0 = O.forward(w0, pv, pm)
print(f"gain {ga} {H}x{W}: box spread {float(bx0.std(dim=1).mean()):.3e}")
q16 = lambda t: t.to(torch.float16).to(torch.float32)

def fwd(stages_q, fp32_res_from=99):
    w = dict(w0)
    # fold BN + round folded kernels (device stores them fp16; exact for the device-exact recipe)
    for k in list(w.keys()):
        if k.endswith(".convolution.weight"):
            pre = k[: -len(".convolution.weight")] + ".normalization"
            scale = w[pre + ".weight"] * (w[pre + ".running_var"] + O.BN_EPS).rsqrt()
            w[k] = q16(w[k] * scale.view(-1, 1, 1, 1))
            w[pre + ".bias"] = w[pre + ".bias"] - w[pre + ".running_mean"] * scale
            w[pre + ".weight"] = torch.ones_like(scale); w[pre + ".running_mean"] = torch.zeros_like(scale); w[pre + ".running_var"] = torch.ones_like(scale) - O.BN_EPS
    arch = O.infer_arch(w)
    x = pv if "stem" not in stages_q else q16(pv)
    p = "model.backbone.model.embedder.embedder"
    x = F.relu(O._fbn(F.conv2d(x, w[p + ".convolution.weight"], None, stride=2, padding=3), w, p + ".normalization"))
    x = F.max_pool2d(x, 3, 2, 1)
    if "stem" in stages_q: x = q16(x)
    for si, depth in enumerate(arch["depths"]):
        q = q16 if si in stages_q else None
        keep32 = si >= fp32_res_from
        x32 = x
        for li in range(depth):
            p = f"model.backbone.model.encoder.stages.{si}.layers.{li}"
            stride = 2 if (li == 0 and si > 0) else 1
            xin = q(x32) if (q is not None) else x32          # what the convolutions read
            res = x32 if keep32 else xin
            if (p + ".shortcut.convolution.weight") in w:
                res = O._fbn(F.conv2d(xin, w[p + ".shortcut.convolution.weight"], None, stride=stride), w, p + ".shortcut.normalization")
                if q is not None and not keep32: res = q(res)
            h = O._conv_layer(xin, w, p + ".layer.0", 1, True, q)
            h = O._conv_layer(h, w, p + ".layer.1", stride, True, q)
            h = O._conv_layer(h, w, p + ".layer.2", 1, False, None)
            x32 = F.relu(h + res)
            if q is not None and not keep32: x32 = q(x32)
        x = x32
    feat = x
    h_, wd = feat.shape[-2:]
    mask = F.interpolate(pm[None].float(), size=(h_, wd)).to(torch.bool)[0]
    proj = F.conv2d(feat if 3 not in stages_q else q16(feat), w["model.input_projection.weight"], w["model.input_projection.bias"])
    xx = proj.flatten(2).transpose(1, 2)
    pos = O.sine_position_embedding(mask, arch["d_model"])
    mem = O.encoder(w, xx, pos, arch["encoder_layers"], arch["heads"], None, None, None)
    hs = O.decoder(w, mem, pos, arch["decoder_layers"], arch["heads"], None, None, None)
    b = F.relu(O._lin(hs, w, "bbox_predictor.layers.0")); b = F.relu(O._lin(b, w, "bbox_predictor.layers.1"))
    return O._lin(b, w, "bbox_predictor.layers.2").sigmoid()

for name, sq, f32 in (("stem only", {"stem"}, 99), ("stage 1 only", {0}, 99), ("stage 2 only", {1}, 99), ("stage 3 only", {2}, 99), ("stage 4 only", {3}, 99),
                      ("all stages (product)", {"stem", 0, 1, 2, 3}, 99), ("all, fp32 residual stream in stages 3-4", {"stem", 0, 1, 2, 3}, 2),
                      ("all, fp32 residual stream in stages 2-4", {"stem", 0, 1, 2, 3}, 1), ("all, fp32 residual stream everywhere", {"stem", 0, 1, 2, 3}, 0)):
    bx = fwd(sq, f32)
    print(f"  {name:44s} |dbox| max {float((bx - bx0).abs().max()):.2e} mean {float((bx - bx0).abs().mean()):.2e}", flush=True)
