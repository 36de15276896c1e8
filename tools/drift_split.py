#!/usr/bin/env python3
"""Where does the fp16 path's box drift come from?  CPU experiment with the oracle's storage-emulation switches
(oracle/detr_oracle.py::forward(emulate=..., emulate_transformer=..., transformer_sites=...)): the fp32 oracle against itself with
fp16 storage emulated at chosen rounding sites — backbone kernels / activations, and per transformer block the GEMM weights, the GEMM
inputs (fp16 shadows of the fp32 residual stream), q / k / v, the softmax weights P, the attention output, the FFN hidden tensor.
`--raw`: weights WITHOUT make_device_exact (ordinary fp32 checkpoint values: the weight sites then matter).
Usage: drift_split.py [r50|r101] H W [gain] [--raw] [--frames N] [--quick]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from office_person_detection_vit_amd.frames import structured_frames  # noqa: E402
from office_person_detection_vit_amd.weights import DetrArch, ensure_weight_file, load_safetensors  # noqa: E402
from oracle import detr_oracle as O  # noqa: E402

ACT = ["enc.attn.in", "enc.attn.q", "enc.attn.kv", "enc.attn.p", "enc.attn.o", "enc.ffn.in", "enc.ffn.h",
       "dec.cross.kvin", "dec.cross.kv", "dec.self.in", "dec.self.q", "dec.self.kv", "dec.self.p", "dec.self.o",
       "dec.cross.in", "dec.cross.q", "dec.cross.p", "dec.cross.o", "dec.ffn.in", "dec.ffn.h"]
WGT = ["w.proj", "w.enc.attn", "w.enc.ffn", "w.dec.cross.kv", "w.dec.self", "w.dec.cross.q", "w.dec.cross.o", "w.dec.ffn"]
# the decoder-side sites a fused decoder with split (hi + lo) operands would make exact; the memory K/V GEMM separately
DEC_ACT = [s for s in ACT if s.startswith(("dec.self", "dec.cross.in", "dec.cross.q", "dec.cross.p", "dec.cross.o", "dec.ffn"))]
DEC_W = ["w.dec.self", "w.dec.cross.q", "w.dec.cross.o", "w.dec.ffn"]
KV = ["dec.cross.kvin", "dec.cross.kv", "w.dec.cross.kv"]
ENC = [s for s in ACT if s.startswith("enc.")] + ["w.proj", "w.enc.attn", "w.enc.ffn"]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    arch_name, H, W = args[0], int(args[1]), int(args[2])
    ga = float(args[3]) if len(args) > 3 else 1.0
    raw = "--raw" in sys.argv
    nfr = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 2
    arch = DetrArch.resnet101() if arch_name == "r101" else DetrArch.resnet50()
    path = ensure_weight_file(os.environ.get("OPD_WEIGHT_CACHE", "/tmp/opd_weights"), arch, 0, ga, arch_name if ga == 1.0 else f"{arch_name}_g{ga}",
                              device_exact=not raw)
    w = O.to_torch(load_safetensors(path))
    frames = structured_frames(nfr, H, W, seed=5150 if raw else 1234)
    pv, pm = O.preprocess(frames)
    t0 = time.time()
    taps0 = {}
    lg0, bx0, mem0 = O.forward(w, pv, pm, taps=taps0)
    feat32 = taps0["stage3"]
    sm = lambda t: torch.softmax(t, -1)
    print(f"{arch_name} {H}x{W} gain {ga} {'RAW' if raw else 'device-exact'} weights, {nfr} frames: box spread over queries "
          f"{float(bx0.std(dim=1).mean()):.3e}   (fp32 forward {time.time() - t0:.1f} s)", flush=True)

    def report(name, lg, bx, mem):
        print(f"  {name:58s} |dbox| {float((bx - bx0).abs().max()):.2e}  mean {float((bx - bx0).abs().mean()):.2e}  "
              f"|dprob| {float((sm(lg) - sm(lg0)).abs().max()):.2e}  |denc| {float((mem - mem0).abs().max()):.2e}", flush=True)

    # ---- backbone: kernels / activations / both -------------------------------------------------------------------------
    feats = {}
    for name, sites in (("backbone: folded kernels fp16", ["bb.w"]), ("backbone: activations fp16", ["bb.act"]), ("backbone: both", None)):
        taps = {}
        lg, bx, mem = O.forward(w, pv, pm, taps=taps, emulate="f16", backbone_sites=sites)
        feats[name] = taps["stage3"]
        report(name, lg, bx, mem)
    feat16 = feats["backbone: both"]
    # ---- transformer on the fp32 backbone features: one site group at a time ----------------------------------------------
    def tr(name, sites, feat=feat32):
        lg, bx, mem = O.forward(w, pv, pm, emulate_transformer="f16", transformer_sites=sites, backbone_features=feat)
        report(name, lg, bx, mem)

    tr("transformer: every site (weights + activations)", None)
    tr("transformer: weights only", WGT)
    tr("transformer: activations only", ACT)
    tr("  encoder side (weights + activations, incl. proj)", ENC)
    tr("  memory K/V GEMM (weights + in + out)", KV)
    tr("  decoder (weights + activations, no K/V, no heads)", DEC_ACT + DEC_W)
    if "--quick" not in sys.argv:
        for s in WGT:
            tr(f"    only {s}", [s])
        for s in ACT:
            tr(f"    only {s}", [s])
    # ---- what a fused decoder with exact (split) operands would leave -------------------------------------------------------
    rest = [s for s in ACT + WGT if s not in DEC_ACT + DEC_W]
    tr("all but the decoder sites (decoder exact)", rest)
    tr("all but decoder + K/V sites (decoder and K/V exact)", [s for s in rest if s not in KV])
    tr("all but decoder + K/V + encoder weights", [s for s in rest if s not in KV and not s.startswith("w.")])
    # ---- the product's numerics and the projected ones ----------------------------------------------------------------
    tr("PRODUCT emulation: backbone both + every transformer site", None, feat16)
    tr("PROJECTED: backbone both + all but the decoder sites", rest, feat16)
    tr("PROJECTED: backbone both + all but decoder + K/V sites", [s for s in rest if s not in KV], feat16)


if __name__ == "__main__":
    main()
