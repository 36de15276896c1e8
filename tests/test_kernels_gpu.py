"""Kernel-level parity (GPU): each hand-written HIP kernel against a torch fp32 CPU reference of the same op on the
SAME fp16-rounded inputs, called through the exported test hooks of libopd_hip.so.

Tolerances (stated per test): outputs are fp16, so one output rounding (2^-11 relative) plus fp32 accumulation-order
noise; integer-valued cases must be bit exact.
"""

import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F
F_ = F  # (some tests use F as a size name)

from office_person_detection_vit_amd import _capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    return _capi.load_library()


@pytest.fixture(params=[0, 0x400, 0x500, 0x600, 0x20], ids=["auto_tiles", "128rows", "160rows", "192rows", "flat_staging"])
def gemm_variant(request, lib):
    """Every instantiation of the implicit-GEMM kernel the dispatcher can pick: the quantisation-aware tile height, each height
    forced, and the flat-address staging path that tensors beyond 2 GiB take (opd_test_set_conv_flags)."""
    lib.opd_test_set_conv_flags(request.param)
    yield request.param
    lib.opd_test_set_conv_flags(0)


DET = np.dtype([("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4"), ("score", "<f4"), ("label", "<i4"), ("query_index", "<i4"),
                ("frame", "<i4")])   # opd_det (include/opd_detr.h)


def _h(a):
    """fp32 array -> (fp16-rounded fp32 array, uint16 bit pattern)."""
    h = np.ascontiguousarray(a, dtype=np.float32).astype(np.float16)
    return h.astype(np.float32), np.ascontiguousarray(h.view(np.uint16))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def run_conv(lib, x_nhwc, w_oihw, bias, stride, pad, relu, res=None, res32=None, out_f32=False, bias_period=0):
    """x [B,H,W,Cin], w [N,Cin,KH,KW] (fp32, already fp16-representable); returns out [B,OH,OW,N] fp32."""
    B, H, W, Cin = x_nhwc.shape
    N, _, KH, KW = w_oihw.shape
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (W + 2 * pad - KW) // stride + 1
    xb = np.ascontiguousarray(x_nhwc.astype(np.float16).view(np.uint16))
    wt = np.ascontiguousarray(w_oihw.transpose(0, 2, 3, 1).reshape(N, KH * KW * Cin).astype(np.float16).view(np.uint16))
    M = B * OH * OW
    out = np.empty((M, N), np.float32 if out_f32 else np.uint16)
    r16 = np.ascontiguousarray(res.reshape(M, N).astype(np.float16).view(np.uint16)) if res is not None else None
    r32 = np.ascontiguousarray(res32.reshape(M, N).astype(np.float32)) if res32 is not None else None
    b = np.ascontiguousarray(bias.astype(np.float32))
    rc = lib.opd_test_conv_gemm(_p(xb), _p(wt), _p(b), _p(r16), _p(r32), _p(out), B, H, W, Cin, OH, OW, N, KH, KW, stride, pad,
                                int(relu), bias_period, int(out_f32), 0)
    _capi.check(rc, "opd_test_conv_gemm")
    o = out if out_f32 else out.view(np.float16).astype(np.float32)
    return o.reshape(B, OH, OW, N)


def ref_conv(x_nhwc, w_oihw, bias, stride, pad, relu, res=None):
    y = F.conv2d(torch.from_numpy(x_nhwc).permute(0, 3, 1, 2), torch.from_numpy(w_oihw), torch.from_numpy(bias),
                 stride=stride, padding=pad).permute(0, 2, 3, 1)
    if res is not None:
        y = y + torch.from_numpy(res)
    if relu:
        y = F.relu(y)
    return y.numpy()


CONV_CASES = [
    # B, H, W, Cin, N, k, stride, relu, residual
    (2, 13, 17, 64, 64, 1, 1, True, False),     # stage-1 1x1, ragged M
    (1, 20, 23, 256, 128, 1, 1, True, False),   # stage-2 reduce
    (2, 9, 11, 64, 256, 1, 1, True, True),      # expand + residual + ReLU
    (2, 14, 15, 64, 64, 3, 1, True, False),     # 3x3 s1 pad 1
    (2, 15, 13, 128, 128, 3, 2, True, False),   # 3x3 s2, odd sizes (H3)
    (1, 17, 21, 256, 512, 1, 2, False, False),  # strided shortcut
    (1, 7, 9, 512, 512, 3, 1, True, False),     # deep K (4608)
    (1, 100, 128, 128, 512, 1, 1, False, False),  # 400 tiles -> the BN=128 instantiation
    (1, 160, 160, 256, 256, 1, 1, True, True),    # 200 256-row tiles, K=256 -> v3 <4,128>
    (1, 226, 227, 256, 64, 1, 1, True, False),    # 201 256-row tiles, N=64 -> v3 <4,64>
    (1, 30, 33, 128, 64, 3, 1, True, False),      # 3x3 N=64 deep K, few tiles -> v3 <2,64>
    (3, 11, 7, 64, 128, 3, 1, True, False),       # 3x3 on images NARROWER than the 128-pixel tile: several rows and
                                                  # image boundaries inside one strip (exercises the per-lane tap masks)
    (2, 50, 84, 256, 256, 3, 1, True, False),     # stage-3 3x3 shape at batch 2
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_gemm_matches_torch(lib, gemm_variant, case):
    B, H, W, Cin, N, k, stride, relu, use_res = case
    rng = np.random.default_rng(hash(case) % (2 ** 32))
    x, _ = _h(rng.standard_normal((B, H, W, Cin)))
    w, _ = _h(rng.standard_normal((N, Cin, k, k)) * np.sqrt(2.0 / (Cin * k * k)))
    bias = rng.standard_normal(N).astype(np.float32) * 0.1
    pad = k // 2
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = _h(rng.standard_normal((B, OH, OW, N)))[0] if use_res else None
    got = run_conv(lib, x, w, bias, stride, pad, relu, res)
    want = ref_conv(x, w, bias, stride, pad, relu, res)
    scale = float(np.abs(want).max())
    # fp16 output rounding (2^-11 rel) + accumulation order: 1.5e-3 of the tensor scale is > 3 sigma
    np.testing.assert_allclose(got, want, atol=1.5e-3 * scale, rtol=1e-3)


W8_CASES = [
    # (B, H, W, Cin, N, k, stride, relu, residual)
    (1, 9, 11, 64, 256, 3, 1, True, False),       # one ragged row tile, the shortest ring (9 k-steps)
    (2, 13, 17, 128, 512, 3, 1, True, False),     # two column tiles, ragged rows
    (2, 14, 15, 128, 256, 3, 2, True, False),     # stride 2 (first block of a stage)
    (1, 25, 42, 512, 512, 3, 1, True, False),     # stage 4's 3x3 on one frame (1050 pixels: 9 row tiles, the last ragged; 72 k-steps)
    (1, 16, 24, 192, 256, 1, 1, False, False),    # 1x1, exactly 3 k-steps (the minimum)
    (2, 10, 21, 512, 1024, 1, 1, True, True),     # 1x1 expand + residual + ReLU
    (1, 25, 42, 2048, 512, 1, 1, True, False),    # stage 4's reduce (32 k-steps)
]


@pytest.mark.parametrize("case", W8_CASES)
def test_conv_w8_equals_four_wave_kernel_bit_for_bit(lib, case):
    """kernels_w8.hip (128 x 256 tiles, eight waves, three-stage ring, staggered wave groups) against kernels_gemm.hip on the same operands:
    per accumulator element the k order and the MFMA order are the same, so the outputs must be IDENTICAL -- which kernel runs a layer is a
    speed choice only.  Plus the torch reference at the usual tolerance."""
    B, H, W, Cin, N, k, stride, relu, use_res = case
    rng = np.random.default_rng(hash(case) % (2 ** 32))
    x, _ = _h(rng.standard_normal((B, H, W, Cin)))
    w, _ = _h(rng.standard_normal((N, Cin, k, k)) * np.sqrt(2.0 / (Cin * k * k)))
    bias = rng.standard_normal(N).astype(np.float32) * 0.1
    pad = k // 2
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = _h(rng.standard_normal((B, OH, OW, N)))[0] if use_res else None
    lib.opd_test_set_conv_flags(0)
    base = run_conv(lib, x, w, bias, stride, pad, relu, res)
    lib.opd_test_set_conv_flags(1 << 12)
    try:
        got = run_conv(lib, x, w, bias, stride, pad, relu, res)
    finally:
        lib.opd_test_set_conv_flags(0)
    want = ref_conv(x, w, bias, stride, pad, relu, res)
    scale = float(np.abs(want).max())
    np.testing.assert_allclose(got, want, atol=1.5e-3 * scale, rtol=1e-3)
    assert np.array_equal(got.view(np.uint32), base.view(np.uint32)), f"max |diff| {np.abs(got - base).max()}"


def test_conv_w8_repeated_launches_are_bit_identical(lib):
    """Race screen for the ring's counted waits and raw barriers: 30 launches of stage 4's 3x3 shape must agree bit for bit."""
    rng = np.random.default_rng(11)
    x, _ = _h(rng.standard_normal((2, 25, 42, 512)))
    w, _ = _h(rng.standard_normal((512, 512, 3, 3)) * np.sqrt(2.0 / (512 * 9)))
    bias = rng.standard_normal(512).astype(np.float32) * 0.1
    lib.opd_test_set_conv_flags(1 << 12)
    try:
        first = run_conv(lib, x, w, bias, 1, 1, True)
        for _ in range(30):
            again = run_conv(lib, x, w, bias, 1, 1, True)
            assert np.array_equal(first.view(np.uint32), again.view(np.uint32))
    finally:
        lib.opd_test_set_conv_flags(0)


def test_conv_gemm_integer_exact(lib, gemm_variant):
    """Small-integer operands: every product and sum is exact in fp16/fp32 -> bit-exact; asymmetric weights catch a
    transposed fragment map (cdna_hip_programming.md §3: 'A=I-check with ASYMMETRIC B')."""
    rng = np.random.default_rng(5)
    B, H, W, Cin, N = 1, 12, 11, 64, 128
    x = rng.integers(-3, 4, (B, H, W, Cin)).astype(np.float32)
    w = np.zeros((N, Cin, 3, 3), np.float32)
    for n in range(N):
        w[n, (n * 7) % Cin, n % 3, (n // 3) % 3] = 1 + (n % 5)   # one tap per output channel, asymmetric
        w[n, (n * 3 + 1) % Cin, (n + 1) % 3, n % 3] += -2
    bias = np.arange(N, dtype=np.float32) - 60
    got = run_conv(lib, x, w, bias, 1, 1, False)
    want = ref_conv(x, w, bias, 1, 1, False)
    np.testing.assert_array_equal(got, want)


def test_conv_residual_integer_exact(lib, gemm_variant):
    """Residual + ReLU epilogue on integer data (exact): catches any mis-pairing in the permlane16_swap 16-byte
    store / residual-load path of the v2 epilogue, including rows >= M of the last tile."""
    rng = np.random.default_rng(6)
    B, H, W, Cin, N = 1, 9, 15, 64, 256   # M = 135: one full tile + 7 rows
    x = rng.integers(-2, 3, (B, H, W, Cin)).astype(np.float32)
    w = rng.integers(-1, 2, (N, Cin, 1, 1)).astype(np.float32)
    bias = (np.arange(N, dtype=np.float32) % 7) - 3
    res = rng.integers(-20, 21, (B, H, W, N)).astype(np.float32)
    got = run_conv(lib, x, w, bias, 1, 0, True, res)
    want = ref_conv(x, w, bias, 1, 0, True, res)
    np.testing.assert_array_equal(got, want)


def test_gemm_rowbias_f32_residual(lib, gemm_variant):
    """Transformer flavour: out_f32 = x.W^T + rowbias[m % period] + res32 (pos-embedding fold, residual stream)."""
    rng = np.random.default_rng(11)
    M, K, N, period = 300, 256, 768, 100
    x, _ = _h(rng.standard_normal((M, 1, 1, K)))
    w, _ = _h(rng.standard_normal((N, K, 1, 1)) / 16)
    rb = rng.standard_normal((period, N)).astype(np.float32)
    res = rng.standard_normal((M, N)).astype(np.float32)
    got = run_conv(lib, x, w, rb, 1, 0, False, res32=res, out_f32=True, bias_period=period).reshape(M, N)
    want = x.reshape(M, K) @ w.reshape(N, K).T + rb[np.arange(M) % period] + res
    np.testing.assert_allclose(got, want, atol=2e-4, rtol=1e-5)


@pytest.mark.parametrize("M,K,splits,ln", [(800, 256, 4, True), (800, 2048, 8, True), (1050, 2048, 4, True), (530, 2048, 4, False)])
def test_gemm_splitk_reduce_ln(lib, M, K, splits, ln):
    """Split-K slices + the fused deterministic reduce / residual / LayerNorm kernel (decoder and FFN-2 path)."""
    rng = np.random.default_rng(M + K + splits)
    x, xb = _h(rng.standard_normal((M, K)))
    w, wb = _h(rng.standard_normal((256, K)) / np.sqrt(K))
    bias = rng.standard_normal(256).astype(np.float32) * 0.1
    res = rng.standard_normal((M, 256)).astype(np.float32)
    g = rng.uniform(0.8, 1.2, 256).astype(np.float32)
    b = (rng.standard_normal(256) * 0.05).astype(np.float32)
    y = np.empty((M, 256), np.float32)
    y16 = np.empty((M, 256), np.uint16)
    rc = lib.opd_test_gemm_splitk_ln(_p(xb), _p(wb), _p(bias), _p(res), _p(g) if ln else None, _p(b) if ln else None, _p(y), _p(y16),
                                     M, K, splits)
    _capi.check(rc, "opd_test_gemm_splitk_ln")
    pre = torch.from_numpy(x @ w.T + bias + res)
    want = F.layer_norm(pre, (256,), torch.from_numpy(g), torch.from_numpy(b), 1e-5).numpy() if ln else pre.numpy()
    np.testing.assert_allclose(y, want, atol=3e-5, rtol=1e-5)
    np.testing.assert_array_equal(y16.view(np.float16), y.astype(np.float16))
    # deterministic: a second call gives the same bits
    y2 = np.empty_like(y)
    lib.opd_test_gemm_splitk_ln(_p(xb), _p(wb), _p(bias), _p(res), _p(g) if ln else None, _p(b) if ln else None, _p(y2), _p(y16), M, K, splits)
    np.testing.assert_array_equal(y, y2)


def _ref_attention(q, k, v, heads, scale):
    B, Lq, D = q.shape
    Lk = k.shape[1]
    dh = D // heads
    Q = torch.from_numpy(q).view(B, Lq, heads, dh).transpose(1, 2)
    K = torch.from_numpy(k).view(B, Lk, heads, dh).transpose(1, 2)
    V = torch.from_numpy(v).view(B, Lk, heads, dh).transpose(1, 2)
    p = torch.softmax(Q @ K.transpose(2, 3) * scale, dim=-1)
    return (p @ V).transpose(1, 2).reshape(B, Lq, D).numpy()


@pytest.mark.parametrize("shape", [(2, 100, 100), (1, 130, 130), (2, 100, 150), (1, 70, 1050), (3, 197, 333), (2, 64, 64), (1, 1, 1)])
def test_attention_matches_torch(lib, shape):
    """softmax(QK^T/sqrt(32))V: decoder self / cross shapes, partial query and key tiles, several frames (the LDS-DMA of a frame's
    last key tile reads into the next frame's rows), a single key."""
    B, Lq, Lk = shape
    heads, D = 8, 256
    rng = np.random.default_rng(Lq * 1000 + Lk)
    q, qb = _h(rng.standard_normal((B, Lq, D)) * 1.5)
    k, kb = _h(rng.standard_normal((B, Lk, D)) * 1.5)
    v, vb = _h(rng.standard_normal((B, Lk, D)))
    out = np.empty((B, Lq, D), np.uint16)
    scale = 32 ** -0.5
    _capi.check(lib.opd_test_attention(_p(qb), _p(kb), _p(vb), _p(out), B, heads, Lq, Lk, scale), "opd_test_attention")
    got = out.view(np.float16).astype(np.float32)
    want = _ref_attention(q, k, v, heads, scale)
    # P is rounded to fp16 before the PV product (rel 2^-11 per weight) and the output to fp16
    np.testing.assert_allclose(got, want, atol=2e-3, rtol=2e-3)


def test_attention_sharp_rows(lib):
    """A dominating key per query (forces large running-max jumps between key tiles): online-softmax rescale path."""
    B, heads, Lq, Lk, D = 1, 8, 64, 300, 256
    rng = np.random.default_rng(9)
    q = rng.standard_normal((B, Lq, D)).astype(np.float32)
    k = rng.standard_normal((B, Lk, D)).astype(np.float32)
    for i in range(Lq):  # key (5*i + 70) % Lk aligned with query i -> score ~ +40 at a tile that varies with i
        k[0, (5 * i + 70) % Lk] = q[0, i] * 4.0
    q, qb = _h(q)
    k, kb = _h(k)
    v, vb = _h(rng.standard_normal((B, Lk, D)))
    out = np.empty((B, Lq, D), np.uint16)
    scale = 32 ** -0.5
    _capi.check(lib.opd_test_attention(_p(qb), _p(kb), _p(vb), _p(out), B, heads, Lq, Lk, scale), "opd_test_attention")
    got = out.view(np.float16).astype(np.float32)
    want = _ref_attention(q, k, v, heads, scale)
    np.testing.assert_allclose(got, want, atol=3e-3, rtol=3e-3)


def test_attention_lagging_reference(lib):
    """Scores that climb steadily along the key axis (about 6 in the log2 domain per 64-key tile, below the kernel's lazy-rescale
    head-room of 8, so the exponent reference lags the running maximum for a tile and then jumps), plus one query whose scores
    fall: the lazily rescaled softmax must equal the exact one."""
    B, heads, Lq, Lk, D = 1, 8, 48, 400, 256
    rng = np.random.default_rng(11)
    q = rng.standard_normal((B, Lq, D)).astype(np.float32)
    q /= np.linalg.norm(q.reshape(B, Lq, heads, 32), axis=-1, keepdims=True).repeat(32, -1).reshape(B, Lq, D)   # unit heads
    k = rng.standard_normal((B, Lk, D)).astype(np.float32) * 0.3
    ramp = (np.arange(Lk, dtype=np.float32) / 64.0) * (6.0 / 1.4427 / 32 ** -0.5)     # + 6 (log2 domain) per tile along q[0, 5]
    k[0] += ramp[:, None] * q[0, 5][None, :]
    k[0, :, :32] -= 2.0 * ramp[:, None] * q[0, 7, :32][None, :]                         # head 0 of query 7: falling scores
    q, qb = _h(q)
    k, kb = _h(k)
    v, vb = _h(rng.standard_normal((B, Lk, D)))
    out = np.empty((B, Lq, D), np.uint16)
    scale = 32 ** -0.5
    _capi.check(lib.opd_test_attention(_p(qb), _p(kb), _p(vb), _p(out), B, heads, Lq, Lk, scale), "opd_test_attention")
    got = out.view(np.float16).astype(np.float32)
    want = _ref_attention(q, k, v, heads, scale)
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, want, atol=3e-3, rtol=3e-3)


def test_layernorm(lib):
    rng = np.random.default_rng(2)
    rows = 1051
    x = (rng.standard_normal((rows, 256)) * 3 + 0.7).astype(np.float32)
    g = rng.uniform(0.8, 1.2, 256).astype(np.float32)
    b = rng.standard_normal(256).astype(np.float32) * 0.05
    y = np.empty((rows, 256), np.float32)
    y16 = np.empty((rows, 256), np.uint16)
    _capi.check(lib.opd_test_layernorm(_p(x), _p(g), _p(b), _p(y), _p(y16), rows), "opd_test_layernorm")
    want = F.layer_norm(torch.from_numpy(x), (256,), torch.from_numpy(g), torch.from_numpy(b), 1e-5).numpy()
    np.testing.assert_allclose(y, want, atol=2e-6, rtol=1e-5)
    np.testing.assert_array_equal(y16.view(np.float16), y.astype(np.float16))


def test_maxpool_exact(lib):
    rng = np.random.default_rng(4)
    B, H, W, Cc = 2, 23, 31, 64
    x, xb = _h(rng.standard_normal((B, H, W, Cc)))
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = np.empty((B, OH, OW, Cc), np.uint16)
    _capi.check(lib.opd_test_maxpool(_p(xb), _p(out), B, H, W, Cc, OH, OW), "opd_test_maxpool")
    want = F.max_pool2d(torch.from_numpy(x).permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).numpy()
    np.testing.assert_array_equal(out.view(np.float16).astype(np.float32), want)


def test_preprocess_matches_oracle(lib):
    from office_person_detection_vit_amd.frames import structured_frames
    from oracle import detr_oracle as O

    H, W = 37, 53
    frames = structured_frames(2, H, W, seed=77)
    pv, _ = O.preprocess(frames)
    batch = np.ascontiguousarray(np.stack(frames))
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = 2 * OH + 6, 2 * OW + 6
    out = np.empty((2, Hp, Wp, 4), np.uint16)
    _capi.check(lib.opd_test_preprocess_u8(_p(batch), _p(out), 2, H, W, Hp, Wp, None), "opd_test_preprocess_u8")
    got = out.view(np.float16)
    want = pv.permute(0, 2, 3, 1).numpy().astype(np.float16)
    np.testing.assert_array_equal(got[:, 3:3 + H, 3:3 + W, :3], want)  # same op order in fp32, one rounding to fp16
    assert not got[..., 3].any()
    border = got.copy()
    border[:, 3:3 + H, 3:3 + W, :] = 0
    assert not border.any()  # the zero border is the stem's padding


def test_preprocess_ragged_batch_matches_oracle(lib):
    """Frames of different sizes on one canvas: the device writes zeros outside each frame's valid rectangle, exactly
    HF's pad-after-normalise (oracle ``preprocess`` on the ragged list)."""
    from office_person_detection_vit_amd.frames import structured_frames
    from oracle import detr_oracle as O

    sizes = [(37, 53), (29, 41)]
    frames = [structured_frames(1, h, w, seed=5 + i)[0] for i, (h, w) in enumerate(sizes)]
    pv, pm = O.preprocess(frames)                      # [2, 3, 37, 53] zero padded, mask
    H, W = 37, 53
    canvas = np.full((2, H, W, 3), 201, np.uint8)      # garbage outside the frames: must be ignored
    for i, f in enumerate(frames):
        canvas[i, :f.shape[0], :f.shape[1]] = f
    valid = np.asarray(sizes, np.int32)
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = 2 * OH + 6, 2 * OW + 6
    out = np.empty((2, Hp, Wp, 4), np.uint16)
    _capi.check(lib.opd_test_preprocess_u8(_p(canvas), _p(out), 2, H, W, Hp, Wp, _p(valid)), "opd_test_preprocess_u8")
    got = out.view(np.float16)
    want = pv.permute(0, 2, 3, 1).numpy().astype(np.float16)
    np.testing.assert_array_equal(got[:, 3:3 + H, 3:3 + W, :3], want)
    assert int(pm[1].sum()) == 29 * 41 and not got[1, 3 + 29:, :, :].any() and not got[1, :, 3 + 41:, :].any()


def test_attention_key_mask_matches_torch(lib):
    """Ragged batch: keys outside each frame's valid (rows x cols) rectangle of the key map contribute nothing —
    the reference adds finfo.min to their scores (HF:models/detr/modeling_detr.py:402-427)."""
    B, heads, D, fh, fw = 3, 8, 256, 9, 14
    Lk, Lq = fh * fw, 100
    rng = np.random.default_rng(21)
    q, qb = _h(rng.standard_normal((B, Lq, D)) * 1.5)
    k, kb = _h(rng.standard_normal((B, Lk, D)) * 1.5)
    v, vb = _h(rng.standard_normal((B, Lk, D)))
    valid = np.asarray([[9, 14], [7, 9], [1, 1]], np.int32)     # full map, a proper sub-rectangle, a single key
    out = np.empty((B, Lq, D), np.uint16)
    scale = 32 ** -0.5
    _capi.check(lib.opd_test_attention_masked(_p(qb), _p(kb), _p(vb), _p(out), B, heads, Lq, Lk, scale, _p(valid), fw),
                "opd_test_attention_masked")
    got = out.view(np.float16).astype(np.float32)
    dh = D // heads
    Q = torch.from_numpy(q).view(B, Lq, heads, dh).transpose(1, 2)
    K = torch.from_numpy(k).view(B, Lk, heads, dh).transpose(1, 2)
    V = torch.from_numpy(v).view(B, Lk, heads, dh).transpose(1, 2)
    sc = Q @ K.transpose(2, 3) * scale
    ys, xs = torch.meshgrid(torch.arange(fh), torch.arange(fw), indexing="ij")
    for b in range(B):
        ok = ((ys < int(valid[b, 0])) & (xs < int(valid[b, 1]))).flatten()
        sc[b, :, :, ~ok] = torch.finfo(torch.float32).min
    want = (torch.softmax(sc, -1) @ V).transpose(1, 2).reshape(B, Lq, D).numpy()
    np.testing.assert_allclose(got, want, atol=2e-3, rtol=2e-3)
    np.testing.assert_allclose(got[2], np.broadcast_to(v[2, :1], (Lq, D)), atol=1e-3)   # one valid key: output = its value


@pytest.mark.parametrize("H,W", [(45, 51), (64, 96), (37, 34)])
def test_stem_conv_padded_dma(lib, H, W):
    """The production stem: 7x7 s2 p3 on the zero-bordered NHWC4 image through the LDS-DMA kernel (odd and even sizes)."""
    rng = np.random.default_rng(H * 100 + W)
    B, N = 2, 64
    x, _ = _h(rng.standard_normal((B, H, W, 3)))
    w, _ = _h(rng.standard_normal((N, 3, 7, 7)) * 0.1)
    bias = rng.standard_normal(N).astype(np.float32) * 0.1
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = 2 * OH + 6, 2 * OW + 6
    x4p = np.zeros((B, Hp, Wp, 4), np.float16)
    x4p[:, 3:3 + H, 3:3 + W, :3] = x
    wt = np.zeros((N, 8, 8, 4), np.float16)
    wt[:, :7, :7, :3] = w.transpose(0, 2, 3, 1)
    out = np.empty((B * OH * OW, N), np.uint16)
    rc = lib.opd_test_stem2(_p(np.ascontiguousarray(x4p.view(np.uint16))), _p(np.ascontiguousarray(wt.view(np.uint16))), _p(bias),
                            _p(out), B, Hp, Wp, OH, OW)
    _capi.check(rc, "opd_test_stem2")
    got = out.view(np.float16).astype(np.float32).reshape(B, OH, OW, N)
    want = ref_conv(x, w, bias, 2, 3, True)
    np.testing.assert_allclose(got, want, atol=1.5e-3 * float(np.abs(want).max()), rtol=1e-3)


@pytest.mark.parametrize("B,H,W,ragged", [(2, 64, 96, False), (2, 61, 83, False), (1, 203, 333, False), (3, 97, 131, True), (2, 800, 1333, False)])
def test_stem_pool_with_preprocessing_inside_is_bit_identical(lib, B, H, W, ragged):
    """stem_pool2_kernel<U8>: the uint8 BGR frames go straight into the stem (normalisation by table look-up while the input patch is
    staged) against preprocess_u8_kernel -> stem_pool2_kernel on the same frames.  The staged patches are the same fp16 values, so the
    pooled maps must be BIT-identical -- image borders, odd sizes, tiles overhanging the map, and the zero region of a ragged batch."""
    rng = np.random.default_rng(B * 7919 + H * 31 + W)
    frames = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    frames[0, : min(H, 5), : min(W, 7)] = 255   # extremes of the table next to the border
    frames[-1, -3:, -5:] = 0
    frames[-1, -1, -1] = (201, 77, 254)          # the very last bytes of the buffer (its size need not be a multiple of 4)
    valid = None
    if ragged:
        valid = np.asarray([[H, W], [H - 13, W - 29], [H // 2 + 1, W // 3 + 2]][:B], dtype=np.int32)
    w = (rng.standard_normal((64, 8, 8, 4)) * 0.1).astype(np.float16)
    w[:, 7, :, :] = 0
    w[:, :, 7, :] = 0
    w[:, :, :, 3] = 0
    bias = (rng.standard_normal(64) * 0.1).astype(np.float32)
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    PH, PW = (OH - 1) // 2 + 1, (OW - 1) // 2 + 1
    fused = np.empty((B, PH, PW, 64), np.uint16)
    split = np.empty((B, PH, PW, 64), np.uint16)
    rc = lib.opd_test_stem_pool_u8(_p(np.ascontiguousarray(frames)), _p(valid), _p(np.ascontiguousarray(w.view(np.uint16))), _p(bias),
                                   _p(fused), _p(split), B, H, W)
    _capi.check(rc, "opd_test_stem_pool_u8")
    assert np.isfinite(split.view(np.float16).astype(np.float32)).all() and split.any()
    np.testing.assert_array_equal(fused, split)


@pytest.mark.parametrize("H,W", [(45, 51), (64, 96), (37, 34), (120, 131), (90, 410)])
def test_fused_stem_pool(lib, H, W):
    """Stem conv + FrozenBN/ReLU + 3x3 s2 max-pool in ONE kernel vs conv2d -> relu -> max_pool2d (odd / even sizes, tiles
    that overhang the map on every side; the last case makes a workgroup walk 6 tiles and the next one the remaining one)."""
    rng = np.random.default_rng(H * 1000 + W)
    B, N = 2, 64
    x, _ = _h(rng.standard_normal((B, H, W, 3)))
    w, _ = _h(rng.standard_normal((N, 3, 7, 7)) * 0.1)
    bias = rng.standard_normal(N).astype(np.float32) * 0.1
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    PH, PW = (OH - 1) // 2 + 1, (OW - 1) // 2 + 1
    Hp, Wp = 2 * OH + 6, 2 * OW + 6
    x4p = np.zeros((B, Hp, Wp, 4), np.float16)
    x4p[:, 3:3 + H, 3:3 + W, :3] = x
    wt = np.zeros((N, 8, 8, 4), np.float16)
    wt[:, :7, :7, :3] = w.transpose(0, 2, 3, 1)
    out = np.empty((B, PH, PW, N), np.uint16)
    rc = lib.opd_test_stem_pool(_p(np.ascontiguousarray(x4p.view(np.uint16))), _p(np.ascontiguousarray(wt.view(np.uint16))), _p(bias),
                                _p(out), B, Hp, Wp, OH, OW, PH, PW)
    _capi.check(rc, "opd_test_stem_pool")
    got = out.view(np.float16).astype(np.float32)
    conv = torch.from_numpy(ref_conv(x, w, bias, 2, 3, True)).permute(0, 3, 1, 2)
    conv = conv.to(torch.float16).to(torch.float32)  # the unfused path rounds the stem output to fp16 before pooling
    want = F.max_pool2d(conv, 3, 2, 1).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(got, want, atol=1.5e-3 * float(np.abs(want).max()), rtol=1e-3)


# ---- fused bottleneck tail (kernels_btail.hip): 3x3 -> 1x1 expand + residual + ReLU -> next 1x1 reduce ------------------
def run_btail(lib, x1, w1, b1, w2, b2, res, w3, b3, stride):
    """x1 [B,H,W,C1]; w1 [C1,C1,3,3]; w2 [C2,C1]; w3 [C3,C2] or None; returns (y [B,OH,OW,C2], z [B,OH,OW,C3] | None)."""
    B, H, W, C1 = x1.shape
    C2 = 4 * C1
    C3 = 0 if w3 is None else w3.shape[0]
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    M = B * OH * OW
    f16 = lambda a: np.ascontiguousarray(a.astype(np.float16).view(np.uint16))
    w1t = f16(w1.transpose(0, 2, 3, 1).reshape(C1, 9 * C1))
    y = np.empty((M, C2), np.uint16)
    z = np.empty((M, max(C3, 1)), np.uint16)
    f32 = lambda a: np.ascontiguousarray(a.astype(np.float32))
    args = [f16(x1), w1t, f32(b1), f16(w2), f32(b2), f16(res.reshape(M, C2)) if res is not None else None,
            f16(w3) if C3 else None, f32(b3) if C3 else None, y, z]
    rc = lib.opd_test_btail(*[_p(a) for a in args], B, H, W, C1, C3, stride)
    _capi.check(rc, "opd_test_btail")
    yy = y.view(np.float16).astype(np.float32).reshape(B, OH, OW, C2)
    zz = z.view(np.float16).astype(np.float32).reshape(B, OH, OW, C3) if C3 else None
    return yy, zz


def ref_btail(x1, w1, b1, w2, b2, res, w3, b3, stride):
    """fp32 reference with the fp16 storage points of the UNFUSED path (a1 and y are rounded to fp16 once)."""
    r16 = lambda a: a.astype(np.float16).astype(np.float32)
    a1 = r16(ref_conv(x1, w1, b1, stride, 1, True))
    y = r16(ref_conv(a1, w2[:, :, None, None], b2, 1, 0, True, res))
    z = r16(ref_conv(y, w3[:, :, None, None], b3, 1, 0, True)) if w3 is not None else None
    return y, z


BTAIL_CASES = [
    # B, H, W, C1, C3, stride, residual
    (2, 13, 17, 64, 64, 1, True),     # stage-1 block -> next block's reduce, ragged M (442 rows)
    (1, 24, 21, 64, 128, 1, True),    # last stage-1 block -> stage-2 first reduce
    (2, 9, 11, 64, 0, 1, True),       # no fused reduce
    (1, 16, 16, 64, 64, 1, False),    # no residual, M a multiple of 128
    (2, 15, 13, 128, 128, 2, True),   # stage-2 first block: stride-2 3x3, odd sizes
    (1, 18, 23, 128, 128, 1, True),   # stage-2 block
    (3, 11, 7, 128, 0, 1, True),      # images narrower than the tile; no fused reduce
    # stage 3 (kernels_btail3.hip: eight waves, wave pairs exchange B operands through LDS)
    (2, 13, 17, 256, 256, 1, True),   # stage-3 block -> next block's reduce, ragged M (442 rows)
    (1, 16, 16, 256, 0, 1, True),     # last stage-3 block (no fused reduce), M a multiple of 128
    (1, 9, 11, 256, 256, 1, False),   # no residual, one partial tile
    (2, 15, 13, 256, 256, 2, True),   # stride-2 3x3, odd sizes
]


@pytest.mark.parametrize("case", BTAIL_CASES)
def test_btail_matches_torch(lib, case):
    B, H, W, C1, C3, stride, use_res = case
    C2 = 4 * C1
    rng = np.random.default_rng(hash(case) % (2 ** 32))
    x1, _ = _h(np.abs(rng.standard_normal((B, H, W, C1))))
    w1, _ = _h(rng.standard_normal((C1, C1, 3, 3)) * np.sqrt(2.0 / (9 * C1)))
    w2, _ = _h(rng.standard_normal((C2, C1)) * np.sqrt(2.0 / C1))
    w3 = _h(rng.standard_normal((C3, C2)) * np.sqrt(2.0 / C2))[0] if C3 else None
    b1 = rng.standard_normal(C1).astype(np.float32) * 0.1
    b2 = rng.standard_normal(C2).astype(np.float32) * 0.1
    b3 = rng.standard_normal(C3).astype(np.float32) * 0.1 if C3 else None
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = _h(rng.standard_normal((B, OH, OW, C2)))[0] if use_res else None
    y, z = run_btail(lib, x1, w1, b1, w2, b2, res, w3, b3, stride)
    yr, zr = ref_btail(x1, w1, b1, w2, b2, res, w3, b3, stride)
    # one fp16 rounding of a1 may differ by an ulp through accumulation order and propagates: 2e-3 of the tensor scale
    np.testing.assert_allclose(y, yr, atol=2e-3 * float(np.abs(yr).max()), rtol=2e-3)
    if C3:
        np.testing.assert_allclose(z, zr, atol=2e-3 * float(np.abs(zr).max()), rtol=2e-3)


@pytest.mark.parametrize("C1,C3", [(64, 64), (64, 128), (128, 128), (256, 256)])
def test_btail_integer_exact_and_equals_unfused(lib, C1, C3):
    """Small-integer operands: all three GEMMs are exact, so the fused kernel must be BIT-identical to the torch
    reference and to the three unfused conv_gemm launches; asymmetric one-hot-ish weights catch a wrong k-permutation."""
    rng = np.random.default_rng(C1 + C3)
    B, H, W, C2 = 2, 10, 9, 4 * C1
    x1 = rng.integers(0, 3, (B, H, W, C1)).astype(np.float32)
    w1 = np.zeros((C1, C1, 3, 3), np.float32)
    for n in range(C1):
        w1[n, (n * 7 + 3) % C1, n % 3, (n // 3) % 3] = 1 + (n % 3)
        w1[n, (n * 5 + 1) % C1, (n + 1) % 3, (n // 2) % 3] = -1
    w2 = np.zeros((C2, C1), np.float32)
    for n in range(C2):
        w2[n, (n * 11 + 5) % C1] = 1 + (n % 2)
        w2[n, (n * 3 + 2) % C1] -= 1
    w3 = np.zeros((C3, C2), np.float32)
    for n in range(C3):
        w3[n, (n * 13 + 7) % C2] = 1
        w3[n, (n * 29 + 1) % C2] += 1 + (n % 2)
        w3[n, (n * 17 + 4) % C2] -= 1
    b1 = rng.integers(-1, 2, C1).astype(np.float32)
    b2 = rng.integers(-1, 2, C2).astype(np.float32)
    b3 = rng.integers(-2, 3, C3).astype(np.float32)
    res = rng.integers(-4, 5, (B, H, W, C2)).astype(np.float32)
    y, z = run_btail(lib, x1, w1, b1, w2, b2, res, w3, b3, 1)
    yr, zr = ref_btail(x1, w1, b1, w2, b2, res, w3, b3, 1)
    assert np.abs(yr).max() < 2048 and np.abs(zr).max() < 2048   # exactly representable in fp16
    np.testing.assert_array_equal(y, yr)
    np.testing.assert_array_equal(z, zr)
    a1 = run_conv(lib, x1, w1, b1, 1, 1, True)
    yu = run_conv(lib, a1, w2[:, :, None, None], b2, 1, 0, True, res)
    zu = run_conv(lib, yu, w3[:, :, None, None], b3, 1, 0, True)
    np.testing.assert_array_equal(y, yu)
    np.testing.assert_array_equal(z, zu)


@pytest.mark.parametrize("C1,C3,B,H,W", [(256, 256, 6, 50, 84), (128, 128, 3, 100, 167), (64, 64, 1, 200, 334)])
def test_btail_repeated_launches_are_bit_identical(lib, C1, C3, B, H, W):
    """Race screen: 40 launches of a fused tail on the same random operands (hundreds of workgroups, alternating tile walk direction, a
    256-MiB copy on a second stream per launch to vary the memory latencies) must give 40 identical y and z tensors.  The stage-3 kernel
    orders its LDS-DMA traffic by counted waits and raw barriers only: a wait that is one count short passes every reference check
    whenever the data happens to land in time."""
    rng = np.random.default_rng(C1 + B)
    C2 = 4 * C1
    f16 = lambda a: np.ascontiguousarray(a.astype(np.float16).view(np.uint16))
    x1 = f16(np.abs(rng.standard_normal((B, H, W, C1))))
    w1 = f16((rng.standard_normal((C1, C1, 3, 3)) * np.sqrt(2.0 / (9 * C1))).transpose(0, 2, 3, 1).reshape(C1, 9 * C1))
    w2 = f16(rng.standard_normal((C2, C1)) * np.sqrt(2.0 / C1))
    w3 = f16(rng.standard_normal((C3, C2)) * np.sqrt(2.0 / C2))
    res = f16(rng.standard_normal((B * H * W, C2)))
    b1 = (rng.standard_normal(C1) * 0.1).astype(np.float32)
    b2 = (rng.standard_normal(C2) * 0.1).astype(np.float32)
    b3 = (rng.standard_normal(C3) * 0.1).astype(np.float32)
    n_diff = C.c_int(-1)
    rc = lib.opd_test_btail_repeat(_p(x1), _p(w1), _p(b1), _p(w2), _p(b2), _p(res), _p(w3), _p(b3), B, H, W, C1, C3, 40, C.byref(n_diff))
    _capi.check(rc, "opd_test_btail_repeat")
    assert n_diff.value == 0


def run_btail_sc(lib, x1, w1, b1, w2, b2sc, xs, wsc, w3, b3):
    """fused tail with the block's shortcut convolution inside: xs [B,H,W,64], wsc [256,64]; stride 1, C1 = C3 = 64."""
    B, H, W, C1 = x1.shape
    M = B * H * W
    f16 = lambda a: np.ascontiguousarray(a.astype(np.float16).view(np.uint16))
    f32 = lambda a: np.ascontiguousarray(a.astype(np.float32))
    y = np.empty((M, 256), np.uint16)
    z = np.empty((M, 64), np.uint16)
    args = [f16(x1), f16(w1.transpose(0, 2, 3, 1).reshape(C1, 9 * C1)), f32(b1), f16(w2), f32(b2sc), f16(xs.reshape(M, 64)), f16(wsc), f16(w3), f32(b3), y, z]
    _capi.check(lib.opd_test_btail_sc(*[_p(a) for a in args], B, H, W), "opd_test_btail_sc")
    return y.view(np.float16).astype(np.float32).reshape(B, H, W, 256), z.view(np.float16).astype(np.float32).reshape(B, H, W, 64)


@pytest.mark.parametrize("B,H,W", [(2, 13, 17), (1, 16, 16), (2, 200, 334)])
def test_btail_with_fused_shortcut(lib, B, H, W):
    """First block of stage 1: y = relu(conv1x1(a1, w2) + conv1x1(xs, wsc) + b2 + bsc), z = relu(conv1x1(y, w3) + b3) with the
    shortcut as a second GEMM into the expand's accumulators.  Reference: fp32 with a1 and y rounded to fp16 once (the shortcut
    itself is not rounded: one rounding fewer than the two-launch path, like the fp32 reference).  Ragged M, full tiles only, and
    the batch-8 stage-1 shape's row count per frame."""
    rng = np.random.default_rng(B * H + W)
    x1, _ = _h(np.abs(rng.standard_normal((B, H, W, 64))))
    xs, _ = _h(np.abs(rng.standard_normal((B, H, W, 64))))
    w1, _ = _h(rng.standard_normal((64, 64, 3, 3)) / 24)
    w2, _ = _h(rng.standard_normal((256, 64)) / 8)
    wsc, _ = _h(rng.standard_normal((256, 64)) / 8)
    w3, _ = _h(rng.standard_normal((64, 256)) / 16)
    b1, b2, bsc, b3 = (0.1 * rng.standard_normal(n).astype(np.float32) for n in (64, 256, 256, 64))
    y, z = run_btail_sc(lib, x1, w1, b1, w2, b2 + bsc, xs, wsc, w3, b3)
    r16 = lambda a: a.astype(np.float16).astype(np.float32)
    a1 = r16(ref_conv(x1, w1, b1, 1, 1, True))
    pre = ref_conv(a1, w2[:, :, None, None], b2 + bsc, 1, 0, False) + ref_conv(xs, wsc[:, :, None, None], np.zeros(256, np.float32), 1, 0, False)
    yr = r16(np.maximum(pre, 0))
    np.testing.assert_allclose(y, yr, atol=4e-3, rtol=2e-3)
    zr = r16(ref_conv(y, w3[:, :, None, None], b3, 1, 0, True))     # from the kernel's own y
    np.testing.assert_allclose(z, zr, atol=4e-3, rtol=2e-3)


@pytest.mark.parametrize("B,H,W", [(1, 10, 14), (2, 13, 17), (1, 64, 64), (2, 200, 334)])
def test_btail_residual_rebuild_and_strided_output_are_bit_identical(lib, B, H, W):
    """Stage 1's three tails both ways (kernels_btail.hip, round 5).  New route: the first tail stores its a1 (64 channels) instead of its
    output (256), the second REBUILDS that output chunk by chunk as its residual (BtailParams::rc) and the third stores its own output
    only at even (oh, ow), which is all the next stage's stride-2 shortcut reads.  Everything the old route stores must come out with the
    same bits: y / z of the second tail everywhere, z of the third everywhere, y of the third where it is stored -- and nothing else of
    that buffer may be touched (it is pre-filled with 0xA5 bytes)."""
    rng = np.random.default_rng(1000 * B + 10 * H + W)
    M = B * H * W
    f16 = lambda a: np.ascontiguousarray(a.astype(np.float16).view(np.uint16))
    f32 = lambda a: np.ascontiguousarray(a.astype(np.float32))
    x1 = f16(np.abs(rng.standard_normal((B, H, W, 64))))
    xs = f16(np.abs(rng.standard_normal((M, 64))))
    wsc = f16(rng.standard_normal((256, 64)) / 8)
    w1 = [f16((rng.standard_normal((64, 64, 3, 3)) / 24).transpose(0, 2, 3, 1).reshape(64, 576)) for _ in range(3)]
    w2 = [f16(rng.standard_normal((256, 64)) / 8) for _ in range(3)]
    w3 = [f16(rng.standard_normal((c3, 256)) / 16) for c3 in (64, 64, 128)]
    b1 = [f32(0.1 * rng.standard_normal(64)) for _ in range(3)]
    b2 = [f32(0.1 * rng.standard_normal(256)) for _ in range(3)]
    b3 = [f32(0.1 * rng.standard_normal(c3)) for c3 in (64, 64, 128)]
    arr = lambda xs_: (C.c_void_p * len(xs_))(*[x.ctypes.data for x in xs_])
    yb = [np.empty((M, 256), np.uint16) for _ in range(3)]
    zb = [np.empty((M, 64), np.uint16) for _ in range(3)]
    yc = [np.empty((B, H, W, 256), np.uint16) for _ in range(3)]
    zc = [np.empty((M, 128), np.uint16) for _ in range(3)]
    rc = lib.opd_test_btail_chain(_p(x1), _p(xs), arr(w1), arr(b1), arr(w2), arr(b2), _p(wsc), arr(w3), arr(b3), arr(yb), arr(zb), arr(yc), arr(zc),
                                  B, H, W, 0xA5)
    _capi.check(rc, "opd_test_btail_chain")
    assert np.isfinite(yb[0].view(np.float16)).all() and float(np.abs(yb[0].view(np.float16).astype(np.float32)).max()) > 0.5
    assert np.array_equal(yb[0], yb[1]), "second tail: y differs between a residual read back and a residual rebuilt"
    untouched = np.ones((B, H, W), bool)
    untouched[:, ::2, ::2] = False
    for route in (1, 2):   # 2: the second tail stores no output at all and the third rebuilds both (btail_rc2_kernel)
        assert np.array_equal(zb[0], zb[route]), route
        assert np.array_equal(zc[0], zc[route]), route
        assert np.array_equal(yc[0][:, ::2, ::2], yc[route][:, ::2, ::2]), f"route {route}: third tail's y at the positions a stride-2 1x1 reads"
        assert (yc[route][untouched] == 0xA5A5).all(), f"route {route}: third tail stored y at positions nobody reads"
    assert not (yc[0] == 0xA5A5).all()


def test_btail_fused_shortcut_integer_exact(lib):
    """Integer operands: bit-exact against the exact sums, and against the unfused route (shortcut launch -> residual tail)."""
    rng = np.random.default_rng(77)
    B, H, W = 2, 10, 9
    x1 = rng.integers(0, 3, (B, H, W, 64)).astype(np.float32)
    xs = rng.integers(0, 4, (B, H, W, 64)).astype(np.float32)
    w1 = np.zeros((64, 64, 3, 3), np.float32)
    for n in range(64):
        w1[n, (n * 7 + 3) % 64, n % 3, (n // 3) % 3] = 1 + (n % 3)
    w2 = np.zeros((256, 64), np.float32)
    wsc = np.zeros((256, 64), np.float32)
    for n in range(256):
        w2[n, (n * 11 + 5) % 64] = 1 + (n % 2)
        wsc[n, (n * 5 + 9) % 64] = 1
        wsc[n, (n * 3 + 1) % 64] -= 2
    w3 = np.zeros((64, 256), np.float32)
    for n in range(64):
        w3[n, (n * 13 + 7) % 256] = 1
        w3[n, (n * 29 + 1) % 256] += 1
    b1 = rng.integers(-1, 2, 64).astype(np.float32)
    b2 = rng.integers(-1, 2, 256).astype(np.float32)
    bsc = rng.integers(-2, 3, 256).astype(np.float32)
    b3 = rng.integers(-2, 3, 64).astype(np.float32)
    y, z = run_btail_sc(lib, x1, w1, b1, w2, b2 + bsc, xs, wsc, w3, b3)
    a1 = ref_conv(x1, w1, b1, 1, 1, True)
    yr = np.maximum(ref_conv(a1, w2[:, :, None, None], b2, 1, 0, False) + ref_conv(xs, wsc[:, :, None, None], bsc, 1, 0, False), 0)
    zr = ref_conv(yr, w3[:, :, None, None], b3, 1, 0, True)
    assert np.abs(yr).max() < 2048 and np.abs(zr).max() < 2048
    np.testing.assert_array_equal(y, yr)
    np.testing.assert_array_equal(z, zr)
    sc = run_conv(lib, xs, wsc[:, :, None, None], bsc, 1, 0, False)          # the two-launch route: shortcut, then residual tail
    yu, zu = run_btail(lib, x1, w1, b1, w2, b2, sc, w3, b3, 1)
    np.testing.assert_array_equal(y, yu)
    np.testing.assert_array_equal(z, zu)


# ---- Linear(K -> 256) + bias + residual + LayerNorm in one kernel (kernels_rowln.hip) ---------------------------------------
@pytest.mark.parametrize("variant", [1, 0], ids=["oneshot_k256", "kloop"])
@pytest.mark.parametrize("M,K,use_res", [(800, 256, True), (8400, 256, True), (37, 256, True), (100, 2048, False), (33, 64, True), (48, 256, False),
                                         (97, 256, True)])
def test_gemm_ln_matches_torch(lib, M, K, use_res, variant):
    """y = LN(x W^T + b + res): fp16 operands, fp32 accumulate and statistics; both kernels (K = 256 goes to the one-shot kernel
    by default: 48-row workgroups, whole weight matrix staged at once; other K and variant 0 to the k-loop kernel).  Tolerance: fp32 summation order only
    (the reference is computed in fp32 from the same fp16-rounded operands): 2e-5 abs on O(1) outputs; the fp16 copy
    one rounding (2^-11 rel)."""
    rng = np.random.default_rng(M * 7 + K)
    x, xb = _h(rng.standard_normal((M, K)))
    w, wb = _h(rng.standard_normal((256, K)) / np.sqrt(K))
    bias = rng.standard_normal(256).astype(np.float32) * 0.1
    res = rng.standard_normal((M, 256)).astype(np.float32) if use_res else None
    gamma = (1.0 + 0.1 * rng.standard_normal(256)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(256)).astype(np.float32)
    y = np.empty((M, 256), np.float32)
    y16 = np.empty((M, 256), np.uint16)
    lib.opd_test_set_gemm_ln_kloop(1 - variant)
    try:
        rc = lib.opd_test_gemm_ln(_p(xb), _p(wb), _p(bias), _p(res), _p(gamma), _p(beta), _p(y), _p(y16), M, K)
    finally:
        lib.opd_test_set_gemm_ln_kloop(0)
    _capi.check(rc, "opd_test_gemm_ln")
    pre = torch.from_numpy(x).double() @ torch.from_numpy(w).double().T + torch.from_numpy(bias).double()
    if use_res:
        pre = pre + torch.from_numpy(res).double()
    want = F.layer_norm(pre, (256,), torch.from_numpy(gamma).double(), torch.from_numpy(beta).double(), 1e-5).float().numpy()
    np.testing.assert_allclose(y, want, atol=3e-5, rtol=1e-5)
    np.testing.assert_allclose(y16.view(np.float16).astype(np.float32), want, atol=2e-3, rtol=1e-3)


@pytest.mark.parametrize("M,K,period,in_place,ln", [(8400, 2048, 1050, True, True), (100, 2048, 0, False, True), (333, 128, 111, True, True),
                                                     (64, 64, 0, False, True), (1, 192, 0, True, True), (8400, 2048, 1050, False, False),
                                                     (77, 320, 11, True, False)])
def test_gemm_ln_deep_matches_torch(lib, M, K, period, in_place, ln):
    """Row-owner ring kernel (the encoder's FFN-2): y = LN(x W^T + b + res) with the whole reduction walked by one workgroup
    (three-stage LDS-DMA ring, counted waits), optionally in place on the residual stream and with the position shadow
    fp16(y + pos[row % period]); K = 64 .. 2048 (1, 2, 3 and many k-steps), ragged last tile."""
    rng = np.random.default_rng(M * 3 + K)
    x, xb = _h(rng.standard_normal((M, K)))
    w, wb = _h(rng.standard_normal((256, K)) / np.sqrt(K))
    bias = rng.standard_normal(256).astype(np.float32) * 0.1
    res = rng.standard_normal((M, 256)).astype(np.float32)
    gamma = (1.0 + 0.1 * rng.standard_normal(256)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(256)).astype(np.float32)
    pos = rng.standard_normal((period, 256)).astype(np.float32) if period else None
    y = np.empty((M, 256), np.float32)
    y16 = np.empty((M, 256), np.uint16)
    yp16 = np.empty((M, 256), np.uint16)
    _capi.check(lib.opd_test_gemm_ln_deep(_p(xb), _p(wb), _p(bias), _p(res), _p(gamma if ln else None), _p(beta if ln else None), _p(pos), period,
                                          _p(y), _p(y16), _p(yp16), M, K, int(in_place)), "opd_test_gemm_ln_deep")
    pre = torch.from_numpy(x).double() @ torch.from_numpy(w).double().T + torch.from_numpy(bias).double() + torch.from_numpy(res).double()
    if ln:
        want = F.layer_norm(pre, (256,), torch.from_numpy(gamma).double(), torch.from_numpy(beta).double(), 1e-5).float().numpy()
    else:   # gamma == null: the plain linear layer (+ residual), e.g. the input projection
        want = pre.float().numpy()
    np.testing.assert_allclose(y, want, atol=3e-5, rtol=1e-5)
    np.testing.assert_allclose(y16.view(np.float16).astype(np.float32), want, atol=2e-3, rtol=1e-3)
    if period:
        wantp = (y + pos[np.arange(M) % period]).astype(np.float16)     # one rounding of the fp32 sum, from the kernel's own fp32 y
        np.testing.assert_array_equal(yp16.view(np.float16), wantp)


@pytest.mark.parametrize("M,FF,period,in_place", [(8400, 2048, 1050, True), (100, 2048, 0, False), (333, 256, 111, True), (64, 128, 0, False),
                                                  (1, 384, 0, True), (130, 1024, 13, False)])
def test_enc_ffn_matches_torch(lib, M, FF, period, in_place):
    """The encoder's FFN block in one launch (kernels_rowln.hip::enc_ffn_kernel): y = LN(res + fp16(relu(x W1^T + b1)) W2^T + b2) with the hidden
    chunk in LDS, weights as per-wave fragment streams through wave-private LDS-DMA rings; in place on the residual stream / the fp16 copy as in
    the model, position shadow, 1 .. 16 hidden chunks, ragged last slab.  Reference: double precision on the same fp16 operands with the hidden
    activations rounded to fp16 once (what the two-launch form stores)."""
    rng = np.random.default_rng(M * 7 + FF)
    x, xb = _h(rng.standard_normal((M, 256)))
    w1, w1b = _h(rng.standard_normal((FF, 256)) / 16.0)
    w2, w2b = _h(rng.standard_normal((256, FF)) / np.sqrt(FF))
    b1 = (rng.standard_normal(FF) * 0.3).astype(np.float32)
    b2 = (rng.standard_normal(256) * 0.1).astype(np.float32)
    res = rng.standard_normal((M, 256)).astype(np.float32)
    gamma = (1.0 + 0.1 * rng.standard_normal(256)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(256)).astype(np.float32)
    pos = rng.standard_normal((period, 256)).astype(np.float32) if period else None
    y = np.empty((M, 256), np.float32)
    y16 = np.empty((M, 256), np.uint16)
    yp16 = np.empty((M, 256), np.uint16)
    tail, tail_pos = (3, 2) if period else ((12, 6) if M == 100 else (0, 0))     # the next layer's q / k / v; the decoder's memory k / v (no pos table: y)
    wt, wtb = _h(rng.standard_normal((max(tail, 1) * 256, 256)) / 16.0)
    tb = (rng.standard_normal(max(tail, 1) * 256) * 0.2).astype(np.float32)
    tout = np.empty((M, max(tail, 1) * 256), np.uint16)
    # the stream is packed WITH the front projection's pieces in every case (as the model's is) and run without the front phase here
    _capi.check(lib.opd_test_enc_ffn(_p(xb), _p(w1b), _p(b1), _p(w2b), _p(b2), _p(res), _p(gamma), _p(beta), _p(pos), period, _p(y), _p(y16), _p(yp16),
                                     M, FF, int(in_place), _p(wtb), _p(tb), tail, tail_pos if period else 0, _p(tout), None, None, None, None, 1),
                "opd_test_enc_ffn")
    hid = torch.relu(torch.from_numpy(x).double() @ torch.from_numpy(w1).double().T + torch.from_numpy(b1).double())
    hid = hid.float().half().double()                      # fp32 accumulate -> one fp16 rounding
    pre = hid @ torch.from_numpy(w2).double().T + torch.from_numpy(b2).double() + torch.from_numpy(res).double()
    want = F.layer_norm(pre, (256,), torch.from_numpy(gamma).double(), torch.from_numpy(beta).double(), 1e-5).float().numpy()
    # (a hidden value whose fp32 sum lands next to an fp16 rounding boundary may round the other way than the double-precision reference:
    #  2^-11 relative on one of FF terms)
    np.testing.assert_allclose(y, want, atol=2e-4, rtol=1e-5)
    np.testing.assert_allclose(y16.view(np.float16).astype(np.float32), y, atol=2e-3, rtol=1e-3)
    if period:
        wantp = (y + pos[np.arange(M) % period]).astype(np.float16)
        np.testing.assert_array_equal(yp16.view(np.float16), wantp)
    if tail:   # tail projection of the kernel's own fp16 outputs: pass t < tail_pos on fp16(y + pos), the others on fp16(y); fp32 accumulate, one rounding
        xin = y16.view(np.float16).astype(np.float64)
        xpin = yp16.view(np.float16).astype(np.float64) if period else xin
        got = tout.view(np.float16).astype(np.float32)
        for t in range(tail):
            src = xpin if (period and t < tail_pos) else xin
            want_t = src @ wt[256 * t:256 * t + 256].astype(np.float64).T + tb[256 * t:256 * t + 256]
            np.testing.assert_allclose(got[:, 256 * t:256 * t + 256], want_t, atol=2e-3, rtol=1.2e-3)


@pytest.mark.parametrize("M,FF,period", [(8400, 2048, 1050), (100, 2048, 0), (333, 256, 111), (1, 128, 0)])
def test_enc_ffn_with_front_projection_matches_torch(lib, M, FF, period):
    """enc_ffn_kernel with its FRONT phase: the attention output goes in, x = LN1(res + attn Wo^T + bo) is made inside (fp32 to the residual
    stream, fp16 into LDS as the FFN's operand) and y = LN2(x + fp16(relu(fp16(x) W1^T + b1)) W2^T + b2) comes out: what the model runs per
    encoder layer after the attention (HF:models/detr/modeling_detr.py:640-660)."""
    rng = np.random.default_rng(M * 11 + FF)
    at, atb = _h(rng.standard_normal((M, 256)))
    wo, wob = _h(rng.standard_normal((256, 256)) / 16.0)
    bo = (rng.standard_normal(256) * 0.1).astype(np.float32)
    g1 = (1.0 + 0.1 * rng.standard_normal(256)).astype(np.float32)
    be1 = (0.1 * rng.standard_normal(256)).astype(np.float32)
    w1, w1b = _h(rng.standard_normal((FF, 256)) / 16.0)
    w2, w2b = _h(rng.standard_normal((256, FF)) / np.sqrt(FF))
    b1 = (rng.standard_normal(FF) * 0.3).astype(np.float32)
    b2 = (rng.standard_normal(256) * 0.1).astype(np.float32)
    res = rng.standard_normal((M, 256)).astype(np.float32)
    gamma = (1.0 + 0.1 * rng.standard_normal(256)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(256)).astype(np.float32)
    pos = rng.standard_normal((period, 256)).astype(np.float32) if period else None
    y = np.empty((M, 256), np.float32)
    y16 = np.empty((M, 256), np.uint16)
    yp16 = np.empty((M, 256), np.uint16)
    _capi.check(lib.opd_test_enc_ffn(_p(atb), _p(w1b), _p(b1), _p(w2b), _p(b2), _p(res), _p(gamma), _p(beta), _p(pos), period, _p(y), _p(y16), _p(yp16),
                                     M, FF, 1, None, None, 0, 0, None, _p(wob), _p(bo), _p(g1), _p(be1), 1), "opd_test_enc_ffn")
    T = lambda a: torch.from_numpy(a).double()
    x1 = F.layer_norm(T(at) @ T(wo).T + T(bo) + T(res), (256,), T(g1), T(be1), 1e-5)
    x16 = x1.float().half().double()
    hid = torch.relu(x16 @ T(w1).T + T(b1)).float().half().double()
    want = F.layer_norm(hid @ T(w2).T + T(b2) + x1, (256,), T(gamma), T(beta), 1e-5).float().numpy()
    # (fp16 roundings of x and of the hidden activations that land next to a boundary may go the other way than in the double-precision chain)
    np.testing.assert_allclose(y, want, atol=6e-4, rtol=1e-5)
    np.testing.assert_allclose(y16.view(np.float16).astype(np.float32), y, atol=2e-3, rtol=1e-3)
    if period:
        np.testing.assert_array_equal(yp16.view(np.float16), (y + pos[np.arange(M) % period]).astype(np.float16))


def test_enc_ffn_is_reproducible_under_load(lib):
    """Race screen for enc_ffn_kernel (wave-private LDS-DMA rings with hand-counted waits, a barrier per chunk, front and tail phases): 12
    launches on the same operands at the encoder's full size -- 175 workgroups, every CU streaming weights -- must agree bit for bit; a
    misplaced wait shows as a slab that comes and goes with timing."""
    M, FF, period, tail = 8400, 2048, 1050, 3
    rng = np.random.default_rng(77)
    _, atb = _h(rng.standard_normal((M, 256)))
    _, wob = _h(rng.standard_normal((256, 256)) / 16.0)
    _, w1b = _h(rng.standard_normal((FF, 256)) / 16.0)
    _, w2b = _h(rng.standard_normal((256, FF)) / np.sqrt(FF))
    _, wtb = _h(rng.standard_normal((tail * 256, 256)) / 16.0)
    f32 = lambda n, s=0.1: (rng.standard_normal(n) * s).astype(np.float32)
    b1, b2, bo, tb = f32(FF, 0.3), f32(256), f32(256), f32(tail * 256, 0.2)
    g1, be1, gamma, beta = 1.0 + f32(256), f32(256), 1.0 + f32(256), f32(256)
    res = rng.standard_normal((M, 256)).astype(np.float32)
    pos = rng.standard_normal((period, 256)).astype(np.float32)
    first = None
    for rep in range(12):
        y = np.empty((M, 256), np.float32)
        y16 = np.empty((M, 256), np.uint16)
        yp16 = np.empty((M, 256), np.uint16)
        tout = np.empty((M, tail * 256), np.uint16)
        _capi.check(lib.opd_test_enc_ffn(_p(atb), _p(w1b), _p(b1), _p(w2b), _p(b2), _p(res), _p(gamma), _p(beta), _p(pos), period, _p(y), _p(y16), _p(yp16),
                                         M, FF, 1, _p(wtb), _p(tb), tail, 2, _p(tout), _p(wob), _p(bo), _p(g1), _p(be1), 1), "opd_test_enc_ffn")
        assert np.isfinite(y).all()
        cur = (y.tobytes(), y16.tobytes(), yp16.tobytes(), tout.tobytes())
        if first is None:
            first = cur
        else:
            assert cur == first, f"launch {rep} differs from launch 0"


# ---- one-shot small-M linear layer (kernels_rowln.hip::gemm_k256_kernel) -------------------------------------------------------
@pytest.mark.parametrize("M,N,K,period,relu", [(800, 768, 256, 100, False), (800, 256, 256, 100, False), (800, 2048, 256, 0, True),
                                               (800, 256, 2048, 0, False), (37, 64, 256, 0, True), (130, 256, 512, 0, False)])
def test_gemm_k256_matches_torch(lib, M, N, K, period, relu):
    rng = np.random.default_rng(M + N + K)
    x, xb = _h(rng.standard_normal((M, K)))
    w, wb = _h(rng.standard_normal((N, K)) / np.sqrt(K))
    bias = (rng.standard_normal((max(period, 1), N)) * 0.1).astype(np.float32)
    want = torch.from_numpy(x).double() @ torch.from_numpy(w).double().T
    want = want + (torch.from_numpy(bias).double()[torch.arange(M) % period] if period else torch.from_numpy(bias[0]).double())
    if relu:
        want = F.relu(want)
    want = want.float().numpy()
    out16 = np.empty((M, N), np.uint16)
    out32 = np.empty((M, N), np.float32)
    rc = lib.opd_test_gemm_k256(_p(xb), _p(wb), _p(np.ascontiguousarray(bias)), _p(out16), _p(out32), M, N, K, period, int(relu))
    _capi.check(rc, "opd_test_gemm_k256")
    if K > 256:   # fp32 slabs summed in slice order
        np.testing.assert_allclose(out32, want, atol=3e-5 * np.sqrt(K / 256), rtol=1e-5)
    else:         # one fp16 output rounding
        np.testing.assert_allclose(out16.view(np.float16).astype(np.float32), want, atol=1.5e-3 * float(np.abs(want).max()), rtol=1e-3)


def test_gemm_k256_integer_exact(lib):
    rng = np.random.default_rng(3)
    M, N, K = 70, 128, 256
    x = rng.integers(-2, 3, (M, K)).astype(np.float32)
    w = np.zeros((N, K), np.float32)
    for n in range(N):
        w[n, (n * 37 + 5) % K] = 1 + (n % 3)
        w[n, (n * 11 + 2) % K] -= 2
    bias = rng.integers(-3, 4, (1, N)).astype(np.float32)
    out16 = np.empty((M, N), np.uint16)
    rc = lib.opd_test_gemm_k256(_p(_h(x)[1]), _p(_h(w)[1]), _p(bias), _p(out16), None, M, N, K, 0, 0)
    _capi.check(rc, "opd_test_gemm_k256")
    np.testing.assert_array_equal(out16.view(np.float16).astype(np.float32), x @ w.T + bias)


# ---- dual-source GEMM: the shortcut convolution as extra K of the 1x1 expand (conv_gemm_dma_kernel, DUAL) ------------------------
@pytest.mark.parametrize("B,H,W,Cin,N,Cin2,stride2", [(2, 13, 11, 256, 1024, 512, 2), (1, 25, 42, 512, 2048, 1024, 2), (2, 9, 10, 256, 1024, 512, 1),
                                                      (8, 50, 84, 256, 1024, 512, 2)])
def test_conv_dual_source_matches_torch(lib, B, H, W, Cin, N, Cin2, stride2):
    """out = relu(conv1x1(x, w1) + conv1x1_stride2(x2, w2) + bias) as ONE K-concatenated GEMM vs torch (fp32 on the same fp16 operands;
    one output rounding).  x2 lives at the resolution BEFORE the stride (odd sizes: the last row / column is not sampled)."""
    rng = np.random.default_rng(B * H + W + Cin)
    H2, W2 = (H - 1) * stride2 + 1 + (stride2 - 1), (W - 1) * stride2 + 1
    x, xb = _h(np.abs(rng.standard_normal((B, H, W, Cin))))
    x2, x2b = _h(np.abs(rng.standard_normal((B, H2, W2, Cin2))))
    w1, w1b = _h(rng.standard_normal((N, Cin)) / np.sqrt(Cin))
    w2, w2b = _h(rng.standard_normal((N, Cin2)) / np.sqrt(Cin2))
    bias = (0.1 * rng.standard_normal(N)).astype(np.float32)
    out = np.empty((B * H * W, N), np.uint16)
    rc = lib.opd_test_conv_dual(_p(xb), _p(w1b), _p(x2b), _p(w2b), _p(bias), _p(out), B, H, W, Cin, 1, 1, 0, N, H2, W2, Cin2, stride2, 1)
    _capi.check(rc, "opd_test_conv_dual")
    want = np.maximum(ref_conv(x, w1[:, :, None, None], bias, 1, 0, False) +
                      ref_conv(x2, w2[:, :, None, None], np.zeros(N, np.float32), stride2, 0, False)[:, :H, :W], 0)
    got = out.view(np.float16).astype(np.float32).reshape(B, H, W, N)
    np.testing.assert_allclose(got, want, atol=4e-3, rtol=2e-3)


def test_conv_dual_source_integer_exact(lib):
    """Integer operands: bit-exact, and identical to the two-launch route (shortcut conv, then expand with that residual)."""
    rng = np.random.default_rng(3)
    B, H, W, Cin, N, Cin2 = 2, 7, 9, 256, 1024, 512
    H2, W2 = 2 * H - 1, 2 * W
    x = rng.integers(0, 3, (B, H, W, Cin)).astype(np.float32)
    x2 = rng.integers(0, 3, (B, H2, W2, Cin2)).astype(np.float32)
    w1 = np.zeros((N, Cin), np.float32)
    w2 = np.zeros((N, Cin2), np.float32)
    for n in range(N):
        w1[n, (n * 7 + 3) % Cin] = 1 + (n % 3)
        w1[n, (n * 5 + 1) % Cin] -= 1
        w2[n, (n * 11 + 2) % Cin2] = 2
        w2[n, (n * 3 + 7) % Cin2] -= 1
    b1 = rng.integers(-2, 3, N).astype(np.float32)
    b2 = rng.integers(-2, 3, N).astype(np.float32)
    _, xb = _h(x); _, x2b = _h(x2); _, w1b = _h(w1); _, w2b = _h(w2)
    out = np.empty((B * H * W, N), np.uint16)
    _capi.check(lib.opd_test_conv_dual(_p(xb), _p(w1b), _p(x2b), _p(w2b), _p(b1 + b2), _p(out), B, H, W, Cin, 1, 1, 0, N, H2, W2, Cin2, 2, 1), "dual")
    sc = ref_conv(x2, w2[:, :, None, None], b2, 2, 0, False)[:, :H, :W]
    want = np.maximum(ref_conv(x, w1[:, :, None, None], b1, 1, 0, False) + sc, 0)
    assert np.abs(want).max() < 2048 and np.abs(sc).max() < 2048
    got = out.view(np.float16).astype(np.float32).reshape(B, H, W, N)
    np.testing.assert_array_equal(got, want)
    scu = run_conv(lib, x2, w2[:, :, None, None], b2, 2, 0, False)[:, :H, :W]
    yu = run_conv(lib, x, w1[:, :, None, None], b1, 1, 0, True, np.ascontiguousarray(scu))
    np.testing.assert_array_equal(got, yu)


# ---- output kernels in isolation (kernels_misc.hip): heads, post-process, ROI features ---------------------------------------------
@pytest.mark.parametrize("split", [1, 0])
@pytest.mark.parametrize("rows,ln", [(800, True), (100, False), (37, True), (1, True)])
def test_heads_kernel_matches_torch(lib, rows, ln, split):
    """class_labels_classifier + DetrMLPPredictionHead + sigmoid (HF:models/detr/modeling_detr.py:1284-1300, 1410-1411), with and
    without the decoder's final LayerNorm inside, against float64 torch.  split = 1: kernels_dec.hip::heads2_kernel (the 256-wide layers
    on split fp16 pairs through the decoder's rings: fp32-grade products); split = 0: kernels_misc.hip::heads_kernel (fp32 matrix pipe)."""
    lib.opd_test_set_heads2(split)
    rng = np.random.default_rng(rows)
    hs = rng.standard_normal((rows, 256)).astype(np.float32) * 1.5
    g = (1.0 + 0.1 * rng.standard_normal(256)).astype(np.float32)
    b = (0.1 * rng.standard_normal(256)).astype(np.float32)
    wc = (rng.standard_normal((92, 256)) / 16 * 2).astype(np.float32); bc = rng.standard_normal(92).astype(np.float32)
    w1 = (rng.standard_normal((256, 256)) / 11).astype(np.float32); b1 = rng.standard_normal(256).astype(np.float32) * 0.1
    w2 = (rng.standard_normal((256, 256)) / 11).astype(np.float32); b2 = rng.standard_normal(256).astype(np.float32) * 0.1
    w3 = (rng.standard_normal((4, 256)) / 8).astype(np.float32); b3 = rng.standard_normal(4).astype(np.float32) * 0.1
    logits = np.empty((rows, 92), np.float32)
    boxes = np.empty((rows, 4), np.float32)
    _capi.check(lib.opd_test_heads(_p(hs), _p(g if ln else None), _p(b if ln else None), _p(wc), _p(bc), _p(w1), _p(b1), _p(w2), _p(b2), _p(w3),
                                   _p(b3), _p(logits), _p(boxes), rows, 92), "opd_test_heads")
    t = lambda a: torch.from_numpy(a).double()
    x = t(hs)
    if ln:
        x = F.layer_norm(x, (256,), t(g), t(b), 1e-5)
    want_logits = x @ t(wc).T + t(bc)
    y = F.relu(x @ t(w1).T + t(b1))
    y = F.relu(y @ t(w2).T + t(b2))
    want_boxes = torch.sigmoid(y @ t(w3).T + t(b3))
    lib.opd_test_set_heads2(1)
    np.testing.assert_allclose(logits, want_logits.numpy(), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(boxes, want_boxes.numpy(), atol=2e-6)


def test_postprocess_kernel_matches_oracle(lib):
    """softmax / max over the first C classes / cxcywh -> xyxy x (W, H) / threshold / compaction in query order against the oracle's
    restatement of HF's post_process_object_detection, incl. ties between classes (lowest index wins, like torch.max), a frame with
    no survivor and a frame where every query survives."""
    from oracle import detr_oracle as O
    rng = np.random.default_rng(5)
    B, Q, C1 = 4, 100, 92
    logits = rng.standard_normal((B, Q, C1)).astype(np.float32) * 2.0
    logits[0, :, 1] += 4.0                       # frame 0: person wins often
    logits[1, :, :] = 0.0                        # frame 1: all classes tie -> score 1/92 < threshold, nothing kept
    logits[2, :, 17] = 30.0                      # frame 2: everything kept, label 17
    logits[3, 10, 5] = logits[3, 10, 6] = 9.0    # frame 3, query 10: two classes tie for the maximum -> label 5
    boxes = rng.uniform(0.05, 0.95, (B, Q, 4)).astype(np.float32)
    hw = np.asarray([[720, 1280], [800, 1333], [333, 203], [1080, 1920]], np.int32)
    recs = np.zeros((B, Q), DET)
    counts = np.zeros(B, np.int32)
    thr = 0.3   # (a two-way tie scores just under 0.5)
    _capi.check(lib.opd_test_postprocess(_p(logits), _p(boxes), _p(hw), B, Q, C1, thr, _p(recs), _p(counts)), "opd_test_postprocess")
    want = O.post_process_object_detection(logits, boxes, thr, [tuple(int(v) for v in r) for r in hw])
    assert counts.tolist() == [len(w["scores"]) for w in want] and counts[1] == 0 and counts[2] == Q
    for b in range(B):
        n = counts[b]
        np.testing.assert_array_equal(recs[b, :n]["query_index"], want[b]["query_index"])
        np.testing.assert_array_equal(recs[b, :n]["label"], want[b]["labels"])
        assert (recs[b, :n]["frame"] == b).all()
        np.testing.assert_allclose(recs[b, :n]["score"], want[b]["scores"], rtol=2e-6, atol=1e-7)
        got_xyxy = np.stack([recs[b, :n][k] for k in ("x1", "y1", "x2", "y2")], -1) if n else np.zeros((0, 4), np.float32)
        np.testing.assert_allclose(got_xyxy, want[b]["boxes"], rtol=1e-6, atol=1e-3)
    q10 = np.flatnonzero(recs[3, :counts[3]]["query_index"] == 10)
    assert len(q10) == 1 and recs[3, q10[0]]["label"] == 5


def test_roi_features_kernel_matches_oracle(lib):
    """ROI mean-pool + L2 normalisation on the encoder map (src/tracking/feature_extractor.py:39-88) against the oracle for boxes
    given in map cells: single cell, full map, thin rows / columns."""
    from oracle import detr_oracle as O
    rng = np.random.default_rng(6)
    h, w = 25, 42
    enc = rng.standard_normal((h, w, 256)).astype(np.float32)
    rois = np.asarray([[0, 0, 1, 1], [0, 0, 42, 25], [41, 24, 42, 25], [3, 7, 30, 8], [10, 2, 11, 20], [5, 5, 17, 19]], np.int32)
    out = np.empty((len(rois), 256), np.float32)
    _capi.check(lib.opd_test_roi_features(_p(enc), _p(rois), len(rois), h, w, _p(out)), "opd_test_roi_features")
    want = []
    for x0, y0, x1, y1 in rois:
        f = enc[y0:y1, x0:x1].astype(np.float64).mean(axis=(0, 1))
        want.append(f / (np.linalg.norm(f) + 1e-8))
    np.testing.assert_allclose(out, np.asarray(want), atol=2e-6)
    # and through the oracle's own box -> cell arithmetic on an image-space box
    img = (800, 1333)
    box = (100.0, 200.0, 400.0, 300.0)
    x0, y0 = int(box[0] / img[1] * w), int(box[1] / img[0] * h)
    x1, y1 = int((box[0] + box[2]) / img[1] * w), int((box[1] + box[3]) / img[0] * h)
    r1 = np.asarray([[x0, y0, max(x0 + 1, min(x1, w)), max(y0 + 1, min(y1, h))]], np.int32)
    o1 = np.empty((1, 256), np.float32)
    _capi.check(lib.opd_test_roi_features(_p(enc), _p(r1), 1, h, w, _p(o1)), "opd_test_roi_features")
    np.testing.assert_allclose(o1, O.roi_features(enc, [box], img), atol=2e-6)
