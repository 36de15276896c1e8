"""Kernel-level parity of the FUSED DECODER (csrc/kernels_dec.hip, the key-split form of csrc/kernels_attn.hip, the heads kernel's
FFN prologue) against float64 torch on identical inputs, through the test hooks of libopd_hip_test.so.

The decoder's linear layers run on SPLIT fp16 operands (x = hi + lo / 2048, three MFMAs per product, fp32 accumulate): the weights
and the GEMM inputs here are ordinary fp32 values — NOT fp16-representable — and the tolerances are fp32-grade (1e-5 on O(1) outputs),
which a single-fp16-operand GEMM misses by two orders of magnitude.  Attention scores / P.V use single fp16 operands (q, k, v are stored as
fp16): those tests compare on the fp16-rounded q / k / v.  Follows HF:models/detr/modeling_detr.py:650-739 (decoder layer)."""

import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from office_person_detection_vit_amd import _capi

pytestmark = pytest.mark.gpu

D = 256
SCALE = 32 ** -0.5


@pytest.fixture(scope="module")
def lib():
    return _capi.load_library()


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f16(a):
    return np.ascontiguousarray(np.asarray(a, np.float32).astype(np.float16).view(np.uint16))


def _from16(a):
    return a.view(np.float16).astype(np.float64)


def t(a):
    return torch.from_numpy(np.asarray(a)).double()


def _ulp16(x):
    """Spacing of fp16 at |x| (subnormal spacing below 2^-14)."""
    with np.errstate(over="ignore"):
        return np.spacing(np.abs(x).astype(np.float16)).astype(np.float64)


def k_frag(k, pad=0.0):
    """k [B][Q][256] -> kf [B][8 heads][8 key tiles][64 lanes][8]: lane 16 g + li of tile kt holds k[16 kt + li][32 h + 8 g ..]
    (the order dec_qkv_kernel writes and dec_self_kernel streams: one KiB per MFMA A operand).  Keys >= Q: `pad`."""
    B, Q, _ = k.shape
    kp = np.full((B, 128, 256), pad, np.float64)
    kp[:, :Q] = k
    return kp.reshape(B, 8, 16, 8, 4, 8).transpose(0, 3, 1, 4, 2, 5).reshape(B, 8, 8, 64, 8)   # [b][kt][li][h][g][e] -> [b][h][kt][g][li][e]


def v_frag(v, pad=0.0):
    """v [B][Q][256] -> vf [B][8 heads][4 kb][2 dt][64 lanes][8]: lane 16 g + li holds v[key][32 h + 16 dt + li] for the keys
    kb*32 + 4g + (0..3) and kb*32 + 16 + 4g + (0..3)."""
    B, Q, _ = v.shape
    vp = np.full((B, 128, 256), pad, np.float64)
    vp[:, :Q] = v
    x = vp.reshape(B, 4, 2, 4, 4, 8, 2, 16)            # [b][kb][half][g][j4][h][dt][li]
    return x.transpose(0, 5, 1, 6, 3, 7, 2, 4).reshape(B, 8, 4, 2, 64, 8)   # -> [b][h][kb][dt][g][li][half][j4]


def _ln_params(rng):
    return (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float32), (0.1 * rng.standard_normal(D)).astype(np.float32)


@pytest.mark.parametrize("B,Q,with_partials", [(8, 100, True), (2, 100, False), (1, 100, True), (3, 40, True)])
def test_dec_qkv_kernel(lib, B, Q, with_partials):
    """[h = LN3(h_in + b2 + sum of 16 partial slabs)] ; q | k | v = h . Wqkv^T + bias[row mod Q]; v written transposed per (frame, head)."""
    rng = np.random.default_rng(B * 1000 + Q)
    M, ns = B * Q, 16
    h_in = rng.standard_normal((M, D)).astype(np.float32)
    parts = (rng.standard_normal((ns, M, D)) * 0.3).astype(np.float32)
    b2 = (rng.standard_normal(D) * 0.1).astype(np.float32)
    g, be = _ln_params(rng)
    w = (rng.standard_normal((768, D)) / 16).astype(np.float32)
    bias = (rng.standard_normal((Q, 768)) * 0.5).astype(np.float32)
    h_out = np.zeros((M, D), np.float32)
    q16 = np.zeros((M, D), np.uint16); k16 = np.zeros((B, 8, 8, 64, 8), np.uint16); vT = np.zeros((B, 8, 4, 2, 64, 8), np.uint16)
    rc = lib.opd_test_dec_qkv(_p(h_in), _p(parts) if with_partials else None, ns, _p(b2), _p(g), _p(be), _p(w), _p(bias), M, Q, _p(h_out), _p(q16), _p(k16), _p(vT))
    _capi.check(rc, "opd_test_dec_qkv")
    if with_partials:
        pre = t(h_in) + t(b2)
        for s in range(ns):
            pre = pre + t(parts[s])
        h = F.layer_norm(pre, (D,), t(g), t(be), 1e-5)
        np.testing.assert_allclose(h_out, h.numpy(), atol=3e-6, rtol=2e-6)
    else:
        h = t(h_in)
    qkv = (h @ t(w).T).reshape(B, Q, 768) + t(bias)[None]
    ulp = _ulp16   # one fp16 output rounding
    want_q = qkv[..., :256].reshape(M, D).numpy()
    assert np.all(np.abs(_from16(q16) - want_q) <= 0.51 * ulp(want_q) + 2e-6)
    # k and v arrive in MFMA-fragment order; padding keys are never written (the hook zero-fills the buffers)
    for got, want in ((_from16(k16), k_frag(qkv[..., 256:512].numpy())), (_from16(vT), v_frag(qkv[..., 512:].numpy()))):
        assert np.all(np.abs(got - want) <= 0.51 * ulp(want) + 2e-6)
    valid = k_frag(np.ones((B, Q, 256)))
    assert not _from16(k16)[valid == 0].any() and not _from16(vT)[v_frag(np.ones((B, Q, 256))) == 0].any()


@pytest.mark.parametrize("B,Q", [(8, 100), (1, 100), (2, 36)])
def test_dec_self_kernel(lib, B, Q):
    """Self-attention of every (frame, 16-query slab) with wave = head, o-proj + residual + LayerNorm, then the cross-attention query
    projection; q / k / v are the fp16 operands dec_qkv_kernel writes (garbage in v^T's padding keys must not matter)."""
    rng = np.random.default_rng(B * 77 + Q)
    M = B * Q
    q = (rng.standard_normal((B, Q, D)) * 1.5).astype(np.float32)
    k = (rng.standard_normal((B, Q, D)) * 1.5).astype(np.float32)
    v = rng.standard_normal((B, Q, D)).astype(np.float32)
    q16 = _f16(q.reshape(M, D))
    k_r, v_r = k.astype(np.float16).astype(np.float64), v.astype(np.float16).astype(np.float64)   # the fp16 operands, row-major
    k16 = np.ascontiguousarray(k_frag(k_r, np.nan).astype(np.float16).view(np.uint16))            # NaN in the padding keys
    vT = np.ascontiguousarray(v_frag(v_r, np.nan).astype(np.float16).view(np.uint16))
    h = rng.standard_normal((M, D)).astype(np.float32)
    wo = (rng.standard_normal((D, D)) / 16).astype(np.float32); bo = (rng.standard_normal(D) * 0.1).astype(np.float32)
    g, be = _ln_params(rng)
    wq = (rng.standard_normal((D, D)) / 16).astype(np.float32); rbq = (rng.standard_normal((Q, D)) * 0.5).astype(np.float32)
    h_io = h.copy()
    qc16 = np.zeros((M, D), np.uint16)
    _capi.check(lib.opd_test_dec_self(_p(q16), _p(k16), _p(vT), _p(h_io), _p(wo), _p(bo), _p(g), _p(be), _p(wq), _p(rbq), B, Q, SCALE, _p(qc16)),
                "opd_test_dec_self")
    qh = t(_from16(q16)).reshape(B, Q, 8, 32).transpose(1, 2)
    kh = t(k_r).reshape(B, Q, 8, 32).transpose(1, 2)
    vh = t(v_r).reshape(B, Q, 8, 32).transpose(1, 2)
    p = torch.softmax(qh @ kh.transpose(2, 3) * SCALE, -1)
    o = (p @ vh).transpose(1, 2).reshape(M, D)
    h1 = F.layer_norm(t(h) + o @ t(wo).T + t(bo), (D,), t(g), t(be), 1e-5)
    # P is rounded to fp16 before P.V (relative 2^-11 per weight, averaged over the keys): 2e-3 on O(1) rows after the LayerNorm's gain
    np.testing.assert_allclose(h_io, h1.numpy(), atol=2.5e-3)
    # the projection itself is fp32-grade: check it on the state the kernel actually produced
    qc = (t(h_io) @ t(wq).T).reshape(B, Q, D) + t(rbq)[None]
    want = qc.reshape(M, D).numpy()
    assert np.all(np.abs(_from16(qc16) - want) <= 0.51 * _ulp16(want) + 2e-6)


@pytest.mark.parametrize("B,Lq,Lk,splits,masked", [(8, 100, 1050, 3, False), (2, 100, 1050, 3, True), (1, 100, 100, 3, False), (2, 70, 300, 2, False),
                                                  (1, 100, 2040, 3, False), (2, 100, 1050, 4, True)])
def test_attention_key_split_partials(lib, B, Lq, Lk, splits, masked):
    """attention_kernel<SPLIT>: each key range's unnormalised sum_k p v, exponent reference and sum_k p; combined like
    dec_cross_out_kernel does they give softmax(Q K^T) V.  Incl. more splits than key tiles (an empty range carries zero weight)
    and, with a key mask, a range whose keys are all masked."""
    rng = np.random.default_rng(Lk + splits)
    heads, Dm = 8, 256
    q = (rng.standard_normal((B, Lq, Dm)) * 1.5).astype(np.float32)
    k = (rng.standard_normal((B, Lk, Dm)) * 1.5).astype(np.float32)
    v = rng.standard_normal((B, Lk, Dm)).astype(np.float32)
    q16, k16, v16 = _f16(q), _f16(k), _f16(v)
    key_valid, key_row = None, 0
    mask = torch.zeros(B, 1, 1, Lk, dtype=torch.float64)
    if masked:
        key_row = 42
        rows = (Lk + key_row - 1) // key_row
        kv = np.asarray([[rows, key_row]] + [[max(1, rows // 4), 30]] * (B - 1), np.int32)[:B]   # frame 1: only the first quarter of the rows valid
        key_valid = kv
        for b in range(B):
            idx = np.arange(Lk)
            ok = (idx // key_row < kv[b, 0]) & (idx % key_row < kv[b, 1])
            mask[b, 0, 0, ~torch.from_numpy(ok)] = -np.inf
    M = B * Lq
    part_o = np.zeros((splits, M, Dm), np.float32)
    part_ml = np.zeros((splits, M, heads, 2), np.float32)
    _capi.check(lib.opd_test_attention_split(_p(q16), _p(k16), _p(v16), B, heads, Lq, Lk, SCALE, splits, _p(key_valid), key_row, _p(part_o), _p(part_ml)),
                "opd_test_attention_split")
    m = part_ml[..., 0].astype(np.float64)                      # [S][M][heads]
    l = part_ml[..., 1].astype(np.float64)
    mmax = m.max(axis=0, keepdims=True)
    wgt = np.where(np.isinf(m), 0.0, np.exp2(m - mmax))
    L = (wgt * l).sum(axis=0)                                    # [M][heads]
    O = (np.repeat(wgt, 32, axis=2) * part_o.astype(np.float64)).sum(axis=0) / np.repeat(L, 32, axis=1)
    qh = t(_from16(q16)).reshape(B, Lq, heads, 32).transpose(1, 2)
    kh = t(_from16(k16)).reshape(B, Lk, heads, 32).transpose(1, 2)
    vh = t(_from16(v16)).reshape(B, Lk, heads, 32).transpose(1, 2)
    p = torch.softmax(qh @ kh.transpose(2, 3) * SCALE + mask, -1)
    want = (p @ vh).transpose(1, 2).reshape(M, Dm).numpy()
    np.testing.assert_allclose(O, want, atol=2e-3)               # P in fp16
    assert np.isfinite(part_o).all()


@pytest.mark.parametrize("M,splits,period", [(800, 3, 0), (800, 3, 1), (100, 2, 0), (36, 4, 0)])
def test_dec_cross_out_kernel(lib, M, splits, period):
    """Combine the key splits (incl. a split that carries no weight), o-proj + residual + LayerNorm; residual rows either per row or
    one constant row (layer 0's state)."""
    rng = np.random.default_rng(M + splits)
    part_o = rng.standard_normal((splits, M, D)).astype(np.float32) * 3.0
    m = (rng.standard_normal((splits, M, 8)) * 4.0).astype(np.float32)
    l = rng.uniform(0.5, 40.0, (splits, M, 8)).astype(np.float32)
    m[splits - 1, ::5, :] = -np.inf                                # every fifth row: the last split is empty / fully masked
    l[splits - 1, ::5, :] = 0.0
    part_o[splits - 1, ::5, :] = 0.0
    part_ml = np.ascontiguousarray(np.stack([m, l], axis=-1))
    res = rng.standard_normal((period if period else M, D)).astype(np.float32)
    wo = (rng.standard_normal((D, D)) / 16).astype(np.float32); bo = (rng.standard_normal(D) * 0.1).astype(np.float32)
    g, be = _ln_params(rng)
    h = np.zeros((M, D), np.float32)
    _capi.check(lib.opd_test_dec_cross_out(_p(part_o), _p(part_ml), splits, _p(res), period, _p(wo), _p(bo), _p(g), _p(be), M, _p(h)), "opd_test_dec_cross_out")
    m64, l64 = m.astype(np.float64), l.astype(np.float64)
    wgt = np.where(np.isinf(m64), 0.0, np.exp2(m64 - m64.max(axis=0, keepdims=True)))
    o = (np.repeat(wgt, 32, axis=2) * part_o).sum(axis=0) / np.repeat((wgt * l64).sum(axis=0), 32, axis=1)
    r = t(res) if not period else t(res)[torch.arange(M) % period]
    want = F.layer_norm(r + t(o) @ t(wo).T + t(bo), (D,), t(g), t(be), 1e-5)
    err = np.abs(h - want.numpy())
    bad = np.argwhere(err > 1e-5 + 1e-5 * np.abs(want.numpy()))
    if len(bad):   # diagnostics: which rows / columns, how large the combined attention output is there
        rows = sorted(set(int(b[0]) for b in bad))
        print("violating rows", rows[:20], "cols", sorted(set(int(b[1]) for b in bad))[:20], "max err", err.max(),
              "max|o| in those rows", [float(np.abs(o[r_]).max()) for r_ in rows[:8]], "max|o| overall", float(np.abs(o).max()))
    np.testing.assert_allclose(h, want.numpy(), atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("M,Fh", [(800, 2048), (100, 2048), (37, 256)])
def test_dec_ffn_kernel(lib, M, Fh):
    """relu(h . W1^T + b1) . W2^T as F / 128 partial slabs over 64-row slabs: the slabs' sum against float64, fp32-grade."""
    rng = np.random.default_rng(M + Fh)
    h = rng.standard_normal((M, D)).astype(np.float32)
    w1 = (rng.standard_normal((Fh, D)) / 16).astype(np.float32); b1 = (rng.standard_normal(Fh) * 0.1).astype(np.float32)
    w2 = (rng.standard_normal((D, Fh)) / 32).astype(np.float32)
    parts = np.zeros((Fh // 128, M, D), np.float32)
    _capi.check(lib.opd_test_dec_ffn(_p(h), _p(w1), _p(b1), _p(w2), M, Fh, _p(parts)), "opd_test_dec_ffn")
    hid = F.relu(t(h) @ t(w1).T + t(b1))
    want = hid @ t(w2).T
    np.testing.assert_allclose(parts.astype(np.float64).sum(axis=0), want.numpy(), atol=2e-5, rtol=1e-5)
    c = 3 % (Fh // 128)                                           # one slab on its own
    np.testing.assert_allclose(parts[c], (hid[:, c * 128:(c + 1) * 128] @ t(w2)[:, c * 128:(c + 1) * 128].T).numpy(), atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("split", [1, 0])
def test_heads_kernel_with_ffn_prologue(lib, split):
    """The heads fed by the fused decoder: rows = LN3(hs + b2 + sum of the FFN's partial slabs), final LayerNorm, heads; through
    heads2_kernel (split fp16 operands, the model's default) and heads_kernel (fp32 matrix pipe)."""
    lib.opd_test_set_heads2(split)
    rng = np.random.default_rng(9)
    rows, ns = 800, 16
    hs = rng.standard_normal((rows, D)).astype(np.float32)
    parts = (rng.standard_normal((ns, rows, D)) * 0.3).astype(np.float32)
    b2f = (rng.standard_normal(D) * 0.1).astype(np.float32)
    g3, b3_ = _ln_params(rng)
    g, b = _ln_params(rng)
    wc = (rng.standard_normal((92, D)) / 8).astype(np.float32); bc = rng.standard_normal(92).astype(np.float32)
    w1 = (rng.standard_normal((D, D)) / 11).astype(np.float32); b1 = rng.standard_normal(D).astype(np.float32) * 0.1
    w2 = (rng.standard_normal((D, D)) / 11).astype(np.float32); b2 = rng.standard_normal(D).astype(np.float32) * 0.1
    w3 = (rng.standard_normal((4, D)) / 8).astype(np.float32); b3 = rng.standard_normal(4).astype(np.float32) * 0.1
    logits = np.empty((rows, 92), np.float32); boxes = np.empty((rows, 4), np.float32)
    _capi.check(lib.opd_test_heads_fused(_p(hs), _p(parts), ns, _p(b2f), _p(g3), _p(b3_), _p(g), _p(b), _p(wc), _p(bc), _p(w1), _p(b1), _p(w2), _p(b2),
                                         _p(w3), _p(b3), rows, 92, _p(logits), _p(boxes)), "opd_test_heads_fused")
    x = t(hs) + t(b2f)
    for s in range(ns):
        x = x + t(parts[s])
    x = F.layer_norm(F.layer_norm(x, (D,), t(g3), t(b3_), 1e-5), (D,), t(g), t(b), 1e-5)
    y = F.relu(F.relu(x @ t(w1).T + t(b1)) @ t(w2).T + t(b2))
    lib.opd_test_set_heads2(1)
    np.testing.assert_allclose(logits, (x @ t(wc).T + t(bc)).numpy(), atol=3e-5, rtol=1e-5)
    np.testing.assert_allclose(boxes, torch.sigmoid(y @ t(w3).T + t(b3)).numpy(), atol=2e-6)
