"""What can the end-to-end parity bounds detect?  (SURVEY.md Appendix A acceptance check; VERDICT r1 missing #4.)

The fp32 oracle against itself with deliberately wrong attention arithmetic, on the weight sets and frames the GPU parity
tests use, compared with the bound those tests assert (tests/test_detector_gpu.py::TOL):

* structural errors (the 1/sqrt(d_head) scale missing in one layer, two heads exchanged in V or in the cross-attention K, a
  LayerNorm gain ignored) move a box by >= 20x the asserted bound on the "mild" set and >= 3x on the "sharp" set;
* Appendix A's probe — ONE encoder q_proj scaled by 1.05 — moves a box by 4x (256x320) / 1.4x (800x1333) the bound of the
  mild set and 1.3x the bound of the sharp set: detectable, but NOT the 5x Appendix A asks for.  No operating point of the
  recipe does better (attention gain 1..4, shallow or deep backbone: tools/drift_split.py): both the probe
  and the fp16 storage noise pass through the same attention non-linearity, so their ratio does not move.  Errors of that size
  are caught by the per-kernel tests instead (tests/test_kernels_gpu.py: attention / GEMM / LayerNorm kernels against torch on
  identical inputs to one output rounding), which is why those exist for every kernel.
"""

import numpy as np
import pytest
import torch

from office_person_detection_vit_amd.frames import structured_frames
from office_person_detection_vit_amd.weights import DetrArch, ensure_weight_file, load_safetensors
from oracle import detr_oracle as O

BOX_BOUND = {1.0: 2e-3, 2.0: 3e-2}   # tests/test_detector_gpu.py::TOL at 256x320; 1e-3 at 800x1333 (mild)


def _swap_heads(t):
    t = t.clone()
    a = t[0:32].clone()
    t[0:32] = t[32:64]
    t[32:64] = a
    return t


def _wrong_models(w):
    out = {}
    k = "model.encoder.layers.2.self_attn."
    w2 = dict(w)
    w2[k + "q_proj.weight"], w2[k + "q_proj.bias"] = w[k + "q_proj.weight"] * 32 ** 0.5, w[k + "q_proj.bias"] * 32 ** 0.5
    out["encoder layer 2: 1/sqrt(d_head) missing"] = w2
    w2 = dict(w)
    w2[k + "v_proj.weight"], w2[k + "v_proj.bias"] = _swap_heads(w[k + "v_proj.weight"]), _swap_heads(w[k + "v_proj.bias"])
    out["encoder layer 2: V of heads 0 and 1 exchanged"] = w2
    k = "model.decoder.layers.3.encoder_attn."
    w2 = dict(w)
    w2[k + "k_proj.weight"], w2[k + "k_proj.bias"] = _swap_heads(w[k + "k_proj.weight"]), _swap_heads(w[k + "k_proj.bias"])
    out["decoder layer 3 cross-attention: K of heads 0 and 1 exchanged"] = w2
    w2 = dict(w)
    w2["model.encoder.layers.4.final_layer_norm.weight"] = torch.ones_like(w["model.encoder.layers.4.final_layer_norm.weight"])
    out["encoder layer 4: final LayerNorm gain ignored"] = w2
    return out


def _probe(w):
    key = "model.encoder.layers.0.self_attn.q_proj.weight"
    w2 = dict(w)
    w2[key] = w[key] * 1.05
    return w2


def _dbox(w, w2, pv, pm, bx0):
    _, bx, _ = O.forward(w2, pv, pm)
    return float((bx - bx0).abs().max())


@pytest.mark.parametrize("gain,structural_factor,probe_factor", [(1.0, 20.0, 3.0), (2.0, 3.0, 1.0)])
def test_what_the_end_to_end_bound_detects(weight_cache, gain, structural_factor, probe_factor):
    path = ensure_weight_file(weight_cache, DetrArch.resnet50(), 0, gain, "r50")
    w = O.to_torch(load_safetensors(path))
    pv, pm = O.preprocess(structured_frames(1, 256, 320, seed=1234))
    _, bx0, _ = O.forward(w, pv, pm)
    bound = BOX_BOUND[gain]
    for name, w2 in _wrong_models(w).items():
        d = _dbox(w, w2, pv, pm, bx0)
        assert d >= structural_factor * bound, f"{name}: moves a box by {d:.2e} only ({d / bound:.1f}x the bound)"
    d = _dbox(w, _probe(w), pv, pm, bx0)
    assert d >= probe_factor * bound, f"Appendix-A probe moves a box by {d:.2e} ({d / bound:.1f}x the bound)"
    assert d < 5.0 * bound, "the probe now passes Appendix A's 5x criterion: update the docstring and DESIGN.md section 3"


def test_benchmark_resolution_detects_structural_errors(weight_cache):
    """800x1333 (BASELINE configs[1]; asserted bound 1e-3): structural errors >= 40x the bound, the Appendix-A probe 1.4x."""
    path = ensure_weight_file(weight_cache, DetrArch.resnet50(), 0, 1.0, "r50")
    w = O.to_torch(load_safetensors(path))
    pv, pm = O.preprocess(structured_frames(1, 800, 1333, seed=1234))
    _, bx0, _ = O.forward(w, pv, pm)
    for name, w2 in _wrong_models(w).items():
        d = _dbox(w, w2, pv, pm, bx0)
        assert d >= 40 * 1e-3, f"{name}: moves a box by {d:.2e} only"
    d = _dbox(w, _probe(w), pv, pm, bx0)
    assert 1e-3 <= d < 5e-3
