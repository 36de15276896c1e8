"""The bf16 operand mode (``OPD_FLAG_BF16`` / ``HipDetrDetector(dtype="bf16")`` / ``bench.py --dtype bf16``): BASELINE.json configs[1] names
"1x MI355X bf16"; the default mode is fp16 (same MFMA rate, 8x less rounding error: SURVEY.md section 7 H2).  Every kernel file with 16-bit
operands is ONE source compiled for both element types (``csrc/opd_elem.h``), so these tests check the bf16 instantiation of the same
kernels against torch on identical bf16 inputs, and the end-to-end path against the fp32 oracle with a MEASURED, stated bound (and next to
the oracle's own bf16 storage emulation, which predicts it)."""

import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from office_person_detection_vit_amd import HipDetrDetector, _capi
from office_person_detection_vit_amd.frames import structured_frames
from office_person_detection_vit_amd.weights import DetrArch, ensure_weight_file, load_safetensors
from oracle import detr_oracle as O

pytestmark = pytest.mark.gpu


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _bf(a):
    """fp32 array -> (bf16-rounded fp32 array, uint16 bit pattern)."""
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16)
    return t.to(torch.float32).numpy(), np.ascontiguousarray(t.view(torch.int16).numpy().view(np.uint16))


def _from_bf(bits):
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).view(torch.bfloat16).to(torch.float32).numpy()


@pytest.fixture()
def bf16_hooks():
    lib = _capi.load_library()
    lib.opd_test_set_elem_bf16(1)
    yield lib
    lib.opd_test_set_elem_bf16(0)


@pytest.mark.parametrize("B,H,W,Cin,N,k,stride,relu,res", [(2, 24, 40, 64, 64, 3, 1, True, False), (1, 25, 42, 256, 128, 1, 1, True, True),
                                                          (2, 26, 34, 128, 128, 3, 2, True, False), (1, 20, 33, 512, 256, 1, 1, False, True)])
def test_conv_gemm_bf16(bf16_hooks, B, H, W, Cin, N, k, stride, relu, res):
    """conv_gemm_dma_kernel, bf16 instantiation: implicit GEMM + bias (+ residual) (+ ReLU) against torch on the same bf16 operands; one
    output rounding (2^-8 relative) plus fp32 summation order."""
    lib = bf16_hooks
    rng = np.random.default_rng(Cin + N + k)
    pad = k // 2
    x, xb = _bf(rng.standard_normal((B, H, W, Cin)))
    w, _ = _bf(rng.standard_normal((N, Cin, k, k)) / np.sqrt(Cin * k * k))
    wb = _bf(w.transpose(0, 2, 3, 1).reshape(N, k * k * Cin))[1]
    bias = rng.standard_normal(N).astype(np.float32) * 0.1
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    M = B * OH * OW
    r, rb = (_bf(rng.standard_normal((M, N))) if res else (None, None))
    out = np.empty((M, N), np.uint16)
    rc = lib.opd_test_conv_gemm(_p(xb), _p(wb), _p(bias), _p(rb), None, _p(out), B, H, W, Cin, OH, OW, N, k, k, stride, pad, int(relu), 0, 0, 0)
    _capi.check(rc, "opd_test_conv_gemm")
    y = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w), torch.from_numpy(bias), stride=stride, padding=pad).permute(0, 2, 3, 1).reshape(M, N)
    if res:
        y = y + torch.from_numpy(r)
    if relu:
        y = F.relu(y)
    got, want = _from_bf(out), y.numpy()
    np.testing.assert_allclose(got, want, rtol=2.0 ** -7, atol=2e-3)


@pytest.mark.parametrize("B,Lq,Lk", [(2, 130, 130), (1, 100, 1050), (2, 1050, 1050)])
def test_attention_bf16(bf16_hooks, B, Lq, Lk):
    """attention_kernel, bf16 instantiation (q, k, v, P and the output in bf16, softmax statistics fp32)."""
    lib = bf16_hooks
    rng = np.random.default_rng(Lq + Lk)
    heads, D = 8, 256
    q, qb = _bf(rng.standard_normal((B, Lq, D)))
    k, kb = _bf(rng.standard_normal((B, Lk, D)))
    v, vb = _bf(rng.standard_normal((B, Lk, D)))
    o = np.empty((B * Lq, D), np.uint16)
    _capi.check(lib.opd_test_attention(_p(qb), _p(kb), _p(vb), _p(o), B, heads, Lq, Lk, 32 ** -0.5), "opd_test_attention")
    t = lambda a, L: torch.from_numpy(a).double().reshape(B, L, heads, 32).transpose(1, 2)
    p = torch.softmax(t(q, Lq) @ t(k, Lk).transpose(2, 3) * 32 ** -0.5, -1)
    want = (p @ t(v, Lk)).transpose(1, 2).reshape(B * Lq, D).numpy()
    np.testing.assert_allclose(_from_bf(o), want, atol=1.5e-2)   # P rounded to bf16 (2^-8 per weight) + one output rounding


@pytest.mark.parametrize("size,bound", [((256, 320), 1.2e-2), ((800, 1333), 6e-3)])   # measured on MI355X (round 4): 7.7e-3, 4.2e-3
def test_bf16_mode_end_to_end_bound(weight_cache, parity_log, size, bound):
    """HipDetrDetector(dtype="bf16") against the fp32 oracle, next to what the oracle's own bf16 storage emulation predicts for a path that
    keeps the transformer's residual stream and the decoder's linear layers in fp32 (SURVEY.md section 7 H2 measured 1.6e-3 for a bf16 backbone
    alone at gain 1, 8.9e-3 for everything in bf16).  The bound is a STATED one for this mode, not the north-star tolerance: fp16 is the
    default because bf16 activations cannot meet 1e-3."""
    H, W = size
    path = ensure_weight_file(weight_cache, DetrArch(), 0, 1.0, "r50")
    det = HipDetrDetector(model_path=path, max_batch=2, max_size=(800, 1333), resize=False, dtype="bf16")
    det.load_model()
    try:
        frames = structured_frames(2, H, W, seed=1616)
        lg, bx, enc = det.forward_raw(frames)
        dets = det.detect_batch(frames)
    finally:
        det.close()
    ref = HipDetrDetector(model_path=path, max_batch=2, max_size=(800, 1333), resize=False)   # the default fp16 mode on the same frames
    ref.load_model()
    try:
        lg16, bx16, _ = ref.forward_raw(frames)
    finally:
        ref.close()
    w = O.to_torch(load_safetensors(path))
    pv, pm = O.preprocess(frames)
    lg0, bx0, mem0 = O.forward(w, pv, pm)
    _, bxe, _ = O.forward(w, pv, pm, emulate="bf16", emulate_transformer="bf16",
                          transformer_sites=["w.proj", "w.enc", "enc.", "dec.cross.kvin", "dec.cross.kv", "w.dec.cross.kv", "dec.cross.q", "dec.cross.p",
                                             "dec.self.q", "dec.self.kv", "dec.self.p"])
    sm = lambda t: torch.softmax(torch.as_tensor(t), -1).numpy()
    dbox = float(np.abs(bx - bx0.numpy()).max())
    dbox16 = float(np.abs(bx16 - bx0.numpy()).max())
    pred = float((bxe - bx0).abs().max())
    parity_log(f"r50 mild {H}x{W}, BF16 operand mode vs live oracle", dbox, float(np.abs(sm(lg) - sm(lg0.numpy())).max()),
               float(np.abs(enc - mem0.numpy()).max()), bound,
               f"stated bound of the bf16 mode; fp16 mode on the same frames {dbox16:.1e}, oracle's bf16 storage emulation {pred:.1e}")
    assert dbox <= bound
    assert dbox16 < dbox                      # the default mode is the tighter one
    assert np.isfinite(lg).all() and len(dets) == 2


@pytest.mark.parametrize("M,FF,front", [(8400, 2048, True), (130, 1024, False)])
def test_enc_ffn_bf16(bf16_hooks, M, FF, front):
    """enc_ffn_kernel, bf16 instantiation (x, the weight streams, the hidden chunk and the fp16-typed outputs in bf16; accumulators, residual
    stream and LayerNorm in fp32), with and without its front phase, against double precision on the same bf16 operands with the same
    intermediate roundings."""
    lib = bf16_hooks
    rng = np.random.default_rng(M + FF)
    xin, xinb = _bf(rng.standard_normal((M, 256)))
    wo, wob = _bf(rng.standard_normal((256, 256)) / 16.0)
    w1, w1b = _bf(rng.standard_normal((FF, 256)) / 16.0)
    w2, w2b = _bf(rng.standard_normal((256, FF)) / np.sqrt(FF))
    f32 = lambda n, s=0.1: (rng.standard_normal(n) * s).astype(np.float32)
    b1, b2, bo = f32(FF, 0.3), f32(256), f32(256)
    g1, be1, gamma, beta = 1.0 + f32(256), f32(256), 1.0 + f32(256), f32(256)
    res = rng.standard_normal((M, 256)).astype(np.float32)
    y = np.empty((M, 256), np.float32)
    y16 = np.empty((M, 256), np.uint16)
    yp16 = np.empty((M, 256), np.uint16)
    _capi.check(lib.opd_test_enc_ffn(_p(xinb), _p(w1b), _p(b1), _p(w2b), _p(b2), _p(res), _p(gamma), _p(beta), None, 0, _p(y), _p(y16), _p(yp16), M, FF, 1,
                                     None, None, 0, 0, None, _p(wob) if front else None, _p(bo) if front else None, _p(g1) if front else None,
                                     _p(be1) if front else None, 1), "opd_test_enc_ffn")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).double()
    rb = lambda t: t.float().to(torch.bfloat16).double()      # one rounding to bf16
    if front:
        x1 = F.layer_norm(T(xin) @ T(wo).T + T(bo) + T(res), (256,), T(g1), T(be1), 1e-5)
        x16, r32 = rb(x1), x1
    else:
        x16, r32 = T(xin), T(res)
    hid = rb(torch.relu(x16 @ T(w1).T + T(b1)))
    want = F.layer_norm(hid @ T(w2).T + T(b2) + r32, (256,), T(gamma), T(beta), 1e-5).float().numpy()
    # (bf16 roundings of x and of the hidden activations next to a boundary may go the other way than in the double-precision chain: 2^-8 of
    #  one of 2048 terms)
    np.testing.assert_allclose(y, want, atol=4e-3, rtol=1e-4)
    np.testing.assert_allclose(_from_bf(y16), y, atol=1e-2, rtol=2.0 ** -8)
