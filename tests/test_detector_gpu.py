"""End-to-end parity (GPU): ``HipDetrDetector`` through the C-ABI against the committed HF golden vectors, against the
oracle run live on the same seeded inputs, and through size-independent properties at the benchmark size.

Stated tolerances (normalised cxcywh boxes / softmax probabilities), fp16 storage with fp32 accumulation:
  * BASELINE configs[1] (r50, 800x1333, batch 8) and configs[3] (r101, 1066x1920, batch 8): |dbox| <= 1e-3 = the north-star
    tolerance (measured 3.8e-4 .. 6.8e-4 and 6.8e-4 .. 7.6e-4)
  * small frames (256x320 and the like; fewer tokens, same rounding noise): "mild" weight set |dbox| <= 2e-3, |dprob| <= 4e-3
    (measured 0.6e-3 .. 1.2e-3); r101 at 256x320 |dbox| <= 3e-3 (1.2e-3 .. 2.0e-3 over six launch sequences that differ only in
    summation order: profiles/r02_drift_toggles.txt)
  * "sharp" weight set (attention gain 2): |dbox| <= 3e-2, |dprob| <= 4e-2      (logic-error catcher: box spread is 0.075)
The weight recipe makes every fp16 GEMM operand exactly representable (weights.make_device_exact), so these numbers are fp16
ACTIVATION storage only; the oracle's own storage emulation (``forward(..., emulate="f16")``, tools/drift_split.py) predicts them.
"""

import os

import numpy as np
import pytest
import torch

from office_person_detection_vit_amd import HipDetrDetector
from office_person_detection_vit_amd.data_models import Detection
from office_person_detection_vit_amd.frames import structured_frames
from office_person_detection_vit_amd.weights import DetrArch, ensure_weight_file, load_safetensors
from oracle import detr_oracle as O

pytestmark = pytest.mark.gpu

TOL = {1.0: (2e-3, 4e-3, 3e-2), 2.0: (3e-2, 4e-2, 2e-1)}  # gain -> (box, prob, encoder abs)


def _softmax(x):
    return torch.softmax(torch.from_numpy(np.asarray(x)), -1).numpy()


@pytest.fixture(scope="module")
def detectors(weight_cache):
    cache = {}

    def get(depths=(3, 4, 6, 3), ga=1.0, max_batch=2, max_size=(800, 1333)):
        key = (tuple(depths), ga, max_batch, tuple(max_size))
        if key not in cache:
            tag = "r50" if tuple(depths) == (3, 4, 6, 3) else "r" + "_".join(map(str, depths))
            path = ensure_weight_file(weight_cache, DetrArch(depths=tuple(depths)), 0, ga, tag)
            det = HipDetrDetector(model_path=path, confidence_threshold=0.5, max_batch=max_batch, max_size=max_size,
                                  resize=False)
            det.load_model()
            cache[key] = det
        return cache[key]

    yield get
    for d in cache.values():
        d.close()


def _golden_frames(g):
    return [structured_frames(1, int(h), int(w), seed=int(g["frame_seed"]) + i)[0] for i, (h, w) in enumerate(g["sizes"])]


@pytest.mark.parametrize("tag", ["r50_mild_256x320", "r50_sharp_256x320", "r50_mild_odd_203x333"])
def test_forward_matches_hf_golden(detectors, golden_dir, tag, parity_log):
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    ga = float(g["attention_gain"])
    det = detectors(ga=ga)
    logits, boxes, enc = det.forward_raw(_golden_frames(g))
    tb, tp, te = TOL[ga]
    dbox = float(np.abs(boxes - g["pred_boxes"]).max())
    dprob = float(np.abs(_softmax(logits) - _softmax(g["logits"])).max())
    denc = float(np.abs(enc - g["encoder_last_hidden_state"]).max())
    parity_log(f"{tag} vs HF golden", dbox, dprob, denc, tb)
    assert dbox <= tb and dprob <= tp and denc <= te


@pytest.mark.parametrize("tag", ["r50_mild_256x320", "r50_mild_800x1333", "r50_mild_ragged"])
def test_stage3_fused_tail_forward_matches_golden(weight_cache, golden_dir, parity_log, monkeypatch, tag):
    """Stage 3 through the eight-wave fused tail (kernels_btail3.hip) whatever the launch-cost model says (OPD_TAIL3=2; the default takes
    it for r101 at 1066x1920 and for multi-stream handles): same golden vectors, same bounds as the three-launch path; also a ragged
    batch (zero canvas + padding mask)."""
    monkeypatch.setenv("OPD_TAIL3", "2")
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    path = ensure_weight_file(weight_cache, DetrArch(), 0, 1.0, "r50")
    det = HipDetrDetector(model_path=path, confidence_threshold=0.5, max_batch=2, max_size=(800, 1333), resize=False)
    det.load_model()
    try:
        logits, boxes, enc = det.forward_raw(_golden_frames(g))
    finally:
        det.close()
    full = tag.endswith("800x1333")
    tb, tp, _ = (1e-3, 2e-3, 0) if full else TOL[1.0]
    dbox = float(np.abs(boxes - g["pred_boxes"]).max())
    dprob = float(np.abs(_softmax(logits) - _softmax(g["logits"])).max())
    parity_log(f"{tag} vs HF golden, stage 3 through the fused tail", dbox, dprob, None, tb)
    assert dbox <= tb and dprob <= tp


def test_full_resolution_matches_golden(detectors, golden_dir, parity_log):
    """One 800x1333 frame (BASELINE config resolution) against the HF golden logits / boxes, at the north-star tolerance:
    boxes within 1e-3 (normalised cxcywh = 1.3 px at width 1333)."""
    g = np.load(os.path.join(golden_dir, "r50_mild_800x1333.npz"))
    det = detectors(ga=1.0)
    logits, boxes, enc = det.forward_raw(_golden_frames(g))
    dbox = float(np.abs(boxes - g["pred_boxes"]).max())
    dprob = float(np.abs(_softmax(logits) - _softmax(g["logits"])).max())
    denc = float(np.abs(enc[:, ::97, ::13] - g["encoder_sample"]).max())
    parity_log("r50 mild 800x1333 (BASELINE configs[1] resolution) vs HF golden", dbox, dprob, denc, 1e-3)
    assert dbox <= 1e-3 and dprob <= 2e-3 and denc <= 3e-2


def test_r101_matches_golden(detectors, golden_dir, parity_log):
    """r101 (33 bottlenecks) at 256x320.  The MAXIMUM over the 400 box coordinates is a noisy statistic here: it moves between 1.2e-3
    and 2.1e-3 with the fp32 summation order alone (six launch sequences, tools/drift_toggles.py -> profiles/r02_drift_toggles.txt), so
    the assertion is on statistics whose spread is smaller — mean <= 7e-4 and 99th percentile <= 2e-3 — plus the maximum at 3e-3.  Measured
    over kernel variants of EQUAL precision (same operand types, different fp32 summation order: round 3's launch sequences; round 4's
    channel ownership in the stage 1-2 tails, the encoder FFN as one or two launches, with or without its tail projection): mean
    2.6e-4 .. 5.6e-4, p99 8.9e-4 .. 1.73e-3, max 1.34e-3 .. 1.88e-3 -- this 80-token frame through 33 random-weight bottlenecks is the
    noisiest row of the table; at its BASELINE resolution the plain 1e-3 bound holds (test_config4_r101_1080p_batch8)."""
    g = np.load(os.path.join(golden_dir, "r101_mild_256x320.npz"))
    det = detectors(depths=(3, 4, 23, 3), ga=1.0)
    logits, boxes, enc = det.forward_raw(_golden_frames(g))
    d = np.abs(boxes - g["pred_boxes"]).ravel()
    dbox, mean, p99 = float(d.max()), float(d.mean()), float(np.percentile(d, 99))
    dprob = float(np.abs(_softmax(logits) - _softmax(g["logits"])).max())
    parity_log("r101 mild 256x320 vs HF golden", dbox, dprob, None, 3e-3, f"small frame; mean {mean:.2e}, p99 {p99:.2e}")
    assert mean <= 7e-4 and p99 <= 2e-3 and dbox <= 3e-3 and dprob <= 4e-3


def test_matches_live_oracle_and_postprocess(detectors, weight_cache):
    """HIP path vs the oracle on fresh seeded frames (not in the golden set), including post-processing."""
    det = detectors(ga=1.0)
    w = O.to_torch(load_safetensors(det.model_path))
    frames = structured_frames(2, 288, 352, seed=999)
    pv, pm = O.preprocess(frames)
    lg, bx, mem = O.forward(w, pv, pm)
    logits, boxes, enc = det.forward_raw(frames)
    assert float(np.abs(boxes - bx.numpy()).max()) <= 2e-3
    want = O.post_process_object_detection(lg.numpy(), bx.numpy(), 0.5, [(288, 352)] * 2)
    dets = det.detect_batch(frames)
    for b in range(2):
        ref = O.person_detections(want[b], 0.4)
        got = dets[b]
        # scores within 4e-3 of the threshold may flip; compare the unambiguous ones
        ref_q = {d["query_index"]: d for d in ref if abs(d["confidence"] - 0.5) > 8e-3}
        got_q = {d.query_index: d for d in got if abs(d.confidence - 0.5) > 8e-3}
        assert set(ref_q) == set(got_q)
        for qi, r in ref_q.items():
            d = got_q[qi]
            assert d.class_id == 1 and d.class_name == "person"
            np.testing.assert_allclose(d.bbox, r["bbox"], atol=2e-3 * 352 * 2)
            assert abs(d.confidence - r["confidence"]) <= 4e-3
            assert d.camera_coords == (d.bbox[0] + d.bbox[2] / 2, d.bbox[1] + d.bbox[3])


def test_batch8_full_size_properties(detectors):
    """BASELINE configs[1] shape (batch 8, 800x1333): batch invariance and permutation equivariance — frame i alone
    and inside the batch give the same boxes (the reference's frames are independent)."""
    det = detectors(ga=1.0, max_batch=8)
    frames = structured_frames(8, 800, 1333, seed=4242)
    lg8, bx8, _ = det.forward_raw(frames, want_encoder=False)
    assert np.isfinite(lg8).all() and np.isfinite(bx8).all()
    assert (bx8 >= 0).all() and (bx8 <= 1).all()
    lg1, bx1, _ = det.forward_raw([frames[5]], want_encoder=False)
    np.testing.assert_allclose(bx8[5], bx1[0], atol=1e-6)   # same kernels, same order: bitwise in practice
    perm = [3, 0, 7, 1, 6, 2, 5, 4]
    lgp, bxp, _ = det.forward_raw([frames[i] for i in perm], want_encoder=False)
    np.testing.assert_allclose(bxp, bx8[perm], atol=1e-6)
    assert float(bx8.std(axis=1).mean()) > 5e-3  # queries are not collapsed


def test_graph_replay_matches_eager(weight_cache):
    """The captured hipGraph (3rd call onwards) must reproduce the eager launches bit for bit, also after the input
    frames change between replays."""
    path = ensure_weight_file(weight_cache, DetrArch(), 0, 1.0, "r50")
    outs = {}
    for use_graph in (False, True):
        det = HipDetrDetector(model_path=path, max_batch=2, max_size=(256, 320), resize=False, use_graph=use_graph)
        det.load_model()
        seq = []
        for seed in (10, 10, 10, 20, 10):   # eager, capture, replay, replay (new pixels), replay
            frames = structured_frames(2, 256, 320, seed=seed)
            lg, bx, enc = det.forward_raw(frames)
            seq.append((lg.copy(), bx.copy(), enc.copy()))
        outs[use_graph] = seq
        det.close()
    for a, b in zip(outs[False], outs[True]):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(outs[True][0][1], outs[True][4][1])
    assert np.abs(outs[True][0][1] - outs[True][3][1]).max() > 0


@pytest.mark.parametrize("src,dst", [((72, 128), (75, 133)), ((90, 160), (75, 133)), ((48, 64), (80, 107)), ((211, 97), (60, 32)),
                                     ((720, 1280), (750, 1333))])
def test_device_resize_is_bit_exact_with_pillow(detectors, src, dst):
    """the device-side resize reproduces ``DetrImageProcessor.resize`` (PIL bilinear on uint8) bit for
    bit — integer work, so the bar is exact equality.  Last case: the reference's camera frame 1280x720 -> 750x1333."""
    import ctypes as C
    from PIL import Image
    from office_person_detection_vit_amd import _capi
    det = detectors(ga=1.0, max_batch=2)
    rng = np.random.default_rng(src[0] + dst[1])
    frames = rng.integers(0, 256, (2, src[0], src[1], 3), dtype=np.uint8)
    out = np.empty((2, dst[0], dst[1], 3), np.uint8)
    rc = _capi.load_library().opd_detr_resize_u8(C.c_void_p(det.model), frames.ctypes.data_as(C.c_void_p), 2, src[0], src[1],
                                                 dst[0], dst[1], out.ctypes.data_as(C.c_void_p))
    _capi.check(rc, "opd_detr_resize_u8")
    for b in range(2):
        want = np.asarray(Image.fromarray(frames[b]).resize((dst[1], dst[0]), resample=Image.BILINEAR))
        np.testing.assert_array_equal(out[b], want)


def test_camera_resolution_frames_device_resize_equals_host_resize(weight_cache):
    """detect / forward on camera-resolution frames: resizing on the device gives the SAME model outputs as resizing with
    PIL on the host (identical uint8 pixels after the resize), and boxes come back in camera pixels."""
    path = ensure_weight_file(weight_cache, DetrArch(), 0, 1.0, "r50")
    frames = structured_frames(2, 180, 320, seed=31)           # 16:9 camera frames -> 288 x 512 under max_size (288, 512)
    outs = {}
    for dev in (True, False):
        det = HipDetrDetector(model_path=path, max_batch=2, max_size=(288, 512), resize=True, device_resize=dev, use_graph=False)
        det.load_model()
        outs[dev] = (det.forward_raw(frames), det.detect_batch(frames))
        det.close()
    for a, b in zip(outs[True][0], outs[False][0]):
        np.testing.assert_array_equal(a, b)
    for da, db in zip(outs[True][1], outs[False][1]):
        assert [(d.query_index, d.bbox, d.confidence) for d in da] == [(d.query_index, d.bbox, d.confidence) for d in db]
        for d in da:   # camera-frame pixel coordinates (a box may overhang the frame, never by more than its own size)
            assert 0 < d.bbox[2] <= 2 * 320 and 0 < d.bbox[3] <= 2 * 180 and -320 <= d.bbox[0] <= 320 and -180 <= d.bbox[1] <= 180


def test_frame_list_entry_equals_the_stacked_batch(weight_cache):
    """opd_detr_detect_frames (a LIST of frame pointers, each uploaded from where it lies -- what detect_batch / detect use for same-sized
    contiguous frames) returns exactly what the stacked-block entries return: at model resolution, with the device resize, for a single
    frame, and a non-contiguous frame still goes the stacked way."""
    path = ensure_weight_file(weight_cache, DetrArch(), 0, 1.0, "r50")
    sig = lambda res: [[(d.query_index, d.bbox, d.confidence) for d in dets] for dets in res]
    for (h, w), msz in (((288, 512), (288, 512)), ((180, 320), (288, 512))):
        frames = structured_frames(3, h, w, seed=41)
        outs = {}
        for lists in (True, False):
            det = HipDetrDetector(model_path=path, max_batch=2, max_size=msz, resize=True, frame_lists=lists)
            det.load_model()
            outs[lists] = (sig(det.detect_batch(frames)), sig([det.detect(frames[0])]))          # (3 frames = chunks of 2 + 1)
            if lists:
                wide = np.ascontiguousarray(np.concatenate([frames[1], frames[1]], axis=1))
                view = wide[:, :w]                                                                # same pixels, not contiguous
                assert not view.flags.c_contiguous and det._frame_list_target([view]) is None
                assert sig([det.detect(view)]) == sig([det.detect(frames[1])])
            det.close()
        assert outs[True] == outs[False] and any(len(d) for d in outs[True][0])


def test_detect_with_features_in_one_call_equals_the_two_call_form(weight_cache):
    """opd_detr_detect_frames_features (records + the pooled feature of every person record behind ONE host wait: what detect_with_features
    uses for a contiguous frame) against detect + opd_detr_roi_features on the surviving boxes: same detections, bit-identical features --
    at model resolution and through the device resize, with a low threshold so that suppression has something to drop."""
    path = ensure_weight_file(weight_cache, DetrArch(), 0, 1.0, "r50")
    for (h, w) in ((288, 512), (180, 320)):
        frames = structured_frames(4, h, w, seed=41)
        outs = {}
        for lists in (True, False):
            det = HipDetrDetector(model_path=path, max_batch=1, max_size=(288, 512), resize=True, frame_lists=lists, confidence_threshold=0.05)
            det.load_model()
            res = []
            for frame in frames + frames[:1]:                              # (the last call replays the graph on re-used buffers)
                dets, feats = det.detect_with_features(frame)
                assert len(feats) == len(dets) and all(np.array_equal(d.features, feats[i]) for i, d in enumerate(dets))
                res.append(([(d.query_index, d.bbox, d.confidence) for d in dets], np.asarray(feats)))
            assert res[0][0] == res[-1][0] and np.array_equal(res[0][1], res[-1][1])
            outs[lists] = res
            det.close()
        n = sum(len(r[0]) for r in outs[True])
        assert n >= 2, f"the test frames give {n} person detections at threshold 0.05: nothing to compare"
        for (da, fa), (db, fb) in zip(outs[True], outs[False]):
            assert da == db
            np.testing.assert_array_equal(fa, fb)
            if len(da):
                assert np.allclose(np.linalg.norm(fa, axis=1), 1.0, atol=1e-4)


def test_ragged_batch_matches_hf_golden(detectors, golden_dir):
    """Frames of different sizes in one batch (HF pads to the batch maximum and applies the pixel mask: nearest mask
    down-sampling, mask-dependent position embedding, key masks in encoder self-attention and decoder cross-attention).
    Golden: HF on frames 256x320 and 224x288.  The encoder map is compared on every position, padded ones included."""
    g = np.load(os.path.join(golden_dir, "r50_mild_ragged.npz"))
    det = detectors(ga=float(g["attention_gain"]))
    frames = _golden_frames(g)
    assert frames[0].shape != frames[1].shape
    logits, boxes, enc = det.forward_raw(frames)
    tb, tp, te = TOL[1.0]
    assert float(np.abs(boxes - g["pred_boxes"]).max()) <= tb
    assert float(np.abs(_softmax(logits) - _softmax(g["logits"])).max()) <= tp
    assert float(np.abs(enc - g["encoder_last_hidden_state"]).max()) <= te
    # the padded frame must NOT equal what it gives alone on its own canvas (border leakage through the padded region is
    # part of the reference's behaviour), while the full-size frame is unaffected by its smaller neighbour
    lg0, bx0, _ = det.forward_raw([frames[0]])
    np.testing.assert_allclose(bx0[0], boxes[0], atol=1e-6)
    lg1, bx1, _ = det.forward_raw([frames[1]])
    assert float(np.abs(bx1[0] - boxes[1]).max()) > 1e-5
    # detections of the ragged batch follow the golden post-processing
    dets = det.detect_batch(frames)
    for b in range(2):
        want_q = {int(i) for i in np.nonzero(np.abs(_softmax(g["logits"][b])[:, :-1].max(-1) - 0.5) > 8e-3)[0]
                  if _softmax(g["logits"][b])[i, :-1].max() > 0.5 and _softmax(g["logits"][b])[i, :-1].argmax() == 1}
        got_q = {d.query_index for d in dets[b] if abs(d.confidence - 0.5) > 8e-3}
        assert got_q <= want_q    # NMS may only remove


def test_fused_bottleneck_tail_matches_unfused(weight_cache):
    """Stages 1-2 through kernels_btail.hip (default) against the same layers as three launches each: identical fp16
    rounding points, so only fp32 summation order differs (k-permuted 1x1 operands)."""
    import ctypes as C
    from office_person_detection_vit_amd import _capi
    path = ensure_weight_file(weight_cache, DetrArch(), 0, 1.0, "r50")
    det = HipDetrDetector(model_path=path, max_batch=2, max_size=(256, 320), resize=False, use_graph=False)
    det.load_model()
    frames = structured_frames(2, 256, 320, seed=77)
    lg_f, bx_f, enc_f = det.forward_raw(frames)
    _capi.check(_capi.load_library().opd_test_set_fuse_btail(C.c_void_p(det.model), 1), "set_fuse_btail")   # tails fused, shortcut on its own
    lg_s, bx_s, enc_s = det.forward_raw(frames)
    _capi.check(_capi.load_library().opd_test_set_fuse_btail(C.c_void_p(det.model), 0), "set_fuse_btail")
    lg_u, bx_u, enc_u = det.forward_raw(frames)
    det.close()
    # (bit 1 off: the stage-1 shortcut on its own launch AND the stage 3 / 4 shortcuts as separate launches instead of extra K of the expand)
    assert 0 < np.abs(enc_f - enc_s).max() < TOL[1.0][2] and np.abs(bx_f - bx_s).max() < TOL[1.0][0]
    # (fp16 rounding flips caused by the different fp32 summation order propagate like any other fp16-storage noise:
    #  same bound as against the golden vectors)
    assert np.abs(bx_f - bx_u).max() < TOL[1.0][0] and np.abs(_softmax(lg_f) - _softmax(lg_u)).max() < TOL[1.0][1]
    assert np.abs(enc_f - enc_u).max() < TOL[1.0][2]
    assert np.abs(bx_f - bx_u).max() > 0   # the switch really changed the launch sequence


def test_preprocessing_inside_the_stem_is_bit_identical(weight_cache):
    """uint8 frames: pre-processing inside the stem kernel (default) against preprocess_u8_kernel + stem as two launches -- the same
    fp16 values reach the same MFMAs, so logits, boxes and the encoder map are BIT-identical, for a uniform and for a ragged batch."""
    path = ensure_weight_file(weight_cache, DetrArch.resnet50(), 0, 1.0, "r50")
    det = HipDetrDetector(model_path=path, confidence_threshold=0.5, max_batch=2, max_size=(320, 352), resize=False)
    det.load_model()
    import ctypes as C
    from office_person_detection_vit_amd import _capi
    lib = _capi.load_library()
    try:
        for frames in (structured_frames(2, 256, 320, seed=91),
                       [structured_frames(1, 256, 320, seed=92)[0], structured_frames(1, 224, 288, seed=93)[0]]):
            _capi.check(lib.opd_test_set_fuse_stem_pool(C.c_void_p(det.model), 3), "set_fuse_stem_pool")
            a = det.forward_raw(frames)
            _capi.check(lib.opd_test_set_fuse_stem_pool(C.c_void_p(det.model), 1), "set_fuse_stem_pool")
            b = det.forward_raw(frames)
            for x, y in zip(a, b):
                np.testing.assert_array_equal(x, y)
    finally:
        _capi.check(lib.opd_test_set_fuse_stem_pool(C.c_void_p(det.model), 3), "set_fuse_stem_pool")
        det.close()


def test_position_shadow_matches_bias_tables(weight_cache):
    """q / k projections on fp16(x + pos) (default) against x plus the row-periodic fp32 table W.pos + b (the round-1 form): the same
    linear map, with the position term rounded once to fp16 inside the operand instead of being exact in the bias -- one more fp16
    rounding of the kind every activation already has, so the difference stays inside the golden-vector bounds; uniform and ragged
    batch (per-frame position tables), and a wrong column split (v reading x + pos, or k reading x) would be far outside."""
    path = ensure_weight_file(weight_cache, DetrArch.resnet50(), 0, 1.0, "r50")
    det = HipDetrDetector(model_path=path, confidence_threshold=0.5, max_batch=2, max_size=(320, 352), resize=False)
    det.load_model()
    import ctypes as C
    from office_person_detection_vit_amd import _capi
    lib = _capi.load_library()
    try:
        for frames in (structured_frames(2, 256, 320, seed=95),
                       [structured_frames(1, 256, 320, seed=96)[0], structured_frames(1, 224, 288, seed=97)[0]]):
            _capi.check(lib.opd_test_set_pos_shadow(C.c_void_p(det.model), 1), "set_pos_shadow")
            lg_s, bx_s, enc_s = det.forward_raw(frames)
            _capi.check(lib.opd_test_set_pos_shadow(C.c_void_p(det.model), 0), "set_pos_shadow")
            lg_t, bx_t, enc_t = det.forward_raw(frames)
            assert 0 < np.abs(enc_s - enc_t).max() < TOL[1.0][2]
            assert np.abs(bx_s - bx_t).max() < TOL[1.0][0] and np.abs(_softmax(lg_s) - _softmax(lg_t)).max() < TOL[1.0][1]
    finally:
        _capi.check(lib.opd_test_set_pos_shadow(C.c_void_p(det.model), 1), "set_pos_shadow")
        det.close()


def test_fused_projection_layernorm_matches_unfused(weight_cache):
    """Attention output projections through kernels_rowln.hip (default) against GEMM -> LayerNorm (encoder) and split-K
    GEMM -> reduce + LayerNorm (decoder): same operands, fp32 statistics; only the fp32 summation order differs."""
    import ctypes as C
    from office_person_detection_vit_amd import _capi
    path = ensure_weight_file(weight_cache, DetrArch(), 0, 1.0, "r50")
    det = HipDetrDetector(model_path=path, max_batch=2, max_size=(256, 320), resize=False, use_graph=False)
    det.load_model()
    frames = structured_frames(2, 256, 320, seed=78)
    lg_f, bx_f, enc_f = det.forward_raw(frames)
    _capi.check(_capi.load_library().opd_test_set_fuse_gemm_ln(C.c_void_p(det.model), 0), "set_fuse_gemm_ln")
    lg_u, bx_u, enc_u = det.forward_raw(frames)
    det.close()
    # (an fp16 rounding flip in the x16 shadow propagates like any other fp16-storage noise: golden-vector bounds)
    assert np.abs(bx_f - bx_u).max() < TOL[1.0][0] and np.abs(_softmax(lg_f) - _softmax(lg_u)).max() < TOL[1.0][1]
    assert np.abs(enc_f - enc_u).max() < TOL[1.0][2]
    assert np.abs(enc_f - enc_u).max() > 0   # the switch really changed the launch sequence


def test_detector_surface(detectors):
    det = detectors(ga=1.0)
    frame = structured_frames(1, 256, 320, seed=31)[0]
    keep = frame.copy()
    dets, feats = det.detect_with_features(frame)
    assert (frame == keep).all()  # caller's frame is not mutated
    assert feats.shape == (len(dets), 256) if dets else feats.size == 0
    for d, f in zip(dets, feats):
        assert d.features is f or np.array_equal(d.features, f)
        assert abs(float(np.linalg.norm(f)) - 1.0) < 1e-4
        assert d.bbox[2] > 0 and d.bbox[3] > 0
    assert det.extract_features(frame, []).size == 0
    assert det._get_foot_position((100.0, 200.0, 50.0, 100.0)) == (125.0, 300.0)
    amap = det.get_attention_map(frame)
    assert amap.shape == (8, 10) and amap.dtype == np.float32 and float(amap.max()) == pytest.approx(1.0) and float(amap.min()) >= 0.0
    assert det.detect_batch([]) == []
    with pytest.raises(ValueError):
        det.detect(np.zeros((64, 64), np.uint8))


def test_attention_map_matches_oracle(detectors):
    """``opd_detr_attention_map`` (the DETR-era ``get_attention_map``): the decoder's cross-attention weights, head- and query-averaged,
    against the fp32 oracle's softmax rows (``detr_oracle.cross_attention_map``) for two frames of one batch, the last and an inner
    layer, all queries and a subset.  A row of the map sums to 1; entries are ~1e-2, the stated bound is 2e-4 absolute (measured
    <= 4e-5: fp16 q / k operands, fp32 softmax)."""
    import ctypes as C

    from office_person_detection_vit_amd import _capi
    det = detectors(ga=1.0)
    frames = structured_frames(2, 256, 320, seed=77)
    det.forward_raw(frames, want_encoder=False)
    w = O.to_torch(load_safetensors(det.model_path))
    pv, pm = O.preprocess(frames)
    taps = {}
    O.forward(w, pv, pm, taps=taps)
    lib = _capi.load_library()
    out = np.empty(80, np.float32)
    for frame, layer, queries in ((0, -1, None), (1, -1, [3, 17, 64]), (1, 2, None), (0, 0, [99])):
        q = None if queries is None else np.asarray(queries, np.int32)
        rc = lib.opd_detr_attention_map(C.c_void_p(det.model), frame, layer, None if q is None else q.ctypes.data_as(C.c_void_p),
                                        0 if q is None else len(q), out.ctypes.data_as(C.c_void_p), out.size)
        _capi.check(rc, "opd_detr_attention_map")
        ref = O.cross_attention_map(taps, frame, layer, queries)
        assert abs(float(out.sum()) - 1.0) < 1e-4
        assert float(np.abs(out - ref).max()) <= 2e-4, (frame, layer, queries, float(np.abs(out - ref).max()))
    # argument errors are reported, not executed
    assert lib.opd_detr_attention_map(C.c_void_p(det.model), 2, -1, None, 0, out.ctypes.data_as(C.c_void_p), out.size) != 0
    assert lib.opd_detr_attention_map(C.c_void_p(det.model), 0, -1, None, 0, out.ctypes.data_as(C.c_void_p), out.size - 1) != 0   # buffer too small
    assert lib.opd_detr_attention_map(C.c_void_p(det.model), 0, 6, None, 0, out.ctypes.data_as(C.c_void_p), out.size) != 0
    bad = np.asarray([100], np.int32)
    assert lib.opd_detr_attention_map(C.c_void_p(det.model), 0, -1, bad.ctypes.data_as(C.c_void_p), 1, out.ctypes.data_as(C.c_void_p), out.size) != 0


def test_attention_map_ragged_batch_ignores_padding(detectors):
    """Ragged batch: the keys of the padded area carry no weight (the key mask of HF:models/detr/modeling_detr.py:402-427), and the map of
    the smaller frame matches the oracle run on the same padded batch."""
    import ctypes as C

    from office_person_detection_vit_amd import _capi
    det = detectors(ga=1.0)
    frames = [structured_frames(1, 256, 320, seed=5)[0], structured_frames(1, 192, 256, seed=6)[0]]
    det.forward_raw(frames, want_encoder=False)
    w = O.to_torch(load_safetensors(det.model_path))
    pv, pm = O.preprocess(frames)
    taps = {}
    O.forward(w, pv, pm, taps=taps)
    out = np.empty(80, np.float32)
    _capi.check(_capi.load_library().opd_detr_attention_map(C.c_void_p(det.model), 1, -1, None, 0, out.ctypes.data_as(C.c_void_p), out.size), "attention_map")
    ref = O.cross_attention_map(taps, 1, -1, None)
    assert float(np.abs(out - ref).max()) <= 2e-4
    m = out.reshape(8, 10)
    assert float(m[6:, :].sum()) == 0.0 and float(m[:, 8:].sum()) == 0.0 and abs(float(m.sum()) - 1.0) < 1e-4


def test_roi_features_match_reference_formula(detectors):
    det = detectors(ga=1.0)
    frame = structured_frames(1, 256, 320, seed=32)[0]
    _, _, enc = det.forward_raw([frame])
    from office_person_detection_vit_amd.data_models import Detection
    boxes = [(10.0, 20.0, 100.0, 120.0), (0.0, 0.0, 320.0, 256.0), (300.0, 250.0, 50.0, 50.0)]
    dets = [Detection(bbox=b, confidence=0.9, class_id=1, class_name="person", camera_coords=(0, 0)) for b in boxes]
    got = det.extract_features(frame, dets)
    want = O.roi_features(enc[0].reshape(8, 10, 256), boxes, (256, 320))
    np.testing.assert_allclose(got, want, atol=1e-5)


def test_error_conventions(weight_cache):
    det = HipDetrDetector(model_path=os.path.join(weight_cache, "missing.safetensors"))
    with pytest.raises(RuntimeError, match="Model not loaded"):
        det.detect(np.zeros((64, 64, 3), np.uint8))
    with pytest.raises(RuntimeError, match="Failed to load DETR model"):
        det.load_model()
    bad = os.path.join(weight_cache, "garbage.safetensors")
    with open(bad, "wb") as f:
        f.write(b"\x10\x00\x00\x00\x00\x00\x00\x00{not json at all}")
    det2 = HipDetrDetector(model_path=bad)
    with pytest.raises(RuntimeError, match="Failed to load DETR model"):
        det2.load_model()


def test_similarity_matrix_on_device_matches_reference_golden(golden_dir):
    """the tracker's cost matrix on the device against the reference's own SimilarityCalculator
    (tests/golden/similarity.npz).  Tolerance 1e-6: the fp32 feature dot product is summed in a different order than
    BLAS; the IoU / weighting arithmetic is double like the reference's, so feature-less columns are exact."""
    from office_person_detection_vit_amd.similarity import SimilarityCalculator
    from oracle import similarity_oracle as SO
    g = np.load(os.path.join(golden_dir, "similarity.npz"))
    mk = lambda f, b, ok: Detection(bbox=tuple(float(v) for v in b), confidence=0.9, class_id=1, class_name="person",
                                    camera_coords=(0.0, 0.0), features=f if ok else None)
    d1 = [mk(g["f1"][i], g["b1"][i], True) for i in range(len(g["b1"]))]
    d2 = [mk(g["f2"][j], g["b2"][j], bool(g["has2"][j])) for j in range(len(g["b2"]))]
    s = SimilarityCalculator(0.7, 0.3)
    sim = s.compute_similarity_matrix(d1, d2)
    dist = s.compute_distance_matrix(d1, d2)
    np.testing.assert_allclose(sim, g["similarity"], atol=1e-6)
    np.testing.assert_allclose(dist, g["distance"], atol=1e-6)
    np.testing.assert_array_equal(sim[:, 6], g["similarity"][:, 6])    # motion term only: exact
    np.testing.assert_allclose(sim, SO.similarity_matrix(g["f1"], g["b1"], None, g["f2"], g["b2"], g["has2"]), atol=1e-6)
    # no features anywhere: pure IoU
    for d in d1 + d2:
        d.features = None
    iou = s.compute_similarity_matrix(d1, d2)
    want = np.array([[SO.iou_xywh(a.bbox, b.bbox) for b in d2] for a in d1], np.float32)
    np.testing.assert_array_equal(iou, want)


def test_config4_r101_1080p_batch8(detectors, weight_cache, parity_log):
    """BASELINE.json configs[3]: detr-resnet-101, batch 8, 1080p frames (1066x1920 after the size rule).  Parity on one
    frame against the live oracle (r101 bounds: deeper trunk, more fp16 roundings) and batch invariance at the full batch."""
    det = detectors(depths=(3, 4, 23, 3), ga=1.0, max_batch=8, max_size=(1066, 1920))
    frames = structured_frames(8, 1066, 1920, seed=606)
    lg8, bx8, _ = det.forward_raw(frames, want_encoder=False)
    assert np.isfinite(lg8).all() and (bx8 >= 0).all() and (bx8 <= 1).all()
    assert float(bx8.std(axis=1).mean()) > 5e-3
    lg1, bx1, enc1 = det.forward_raw([frames[2]])
    np.testing.assert_allclose(bx8[2], bx1[0], atol=1e-6)
    w = O.to_torch(load_safetensors(det.model_path))
    pv, pm = O.preprocess([frames[2]])
    lg, bx, mem = O.forward(w, pv, pm)
    dbox, dprob = float(np.abs(bx1[0] - bx[0].numpy()).max()), float(np.abs(_softmax(lg1[0]) - _softmax(lg[0].numpy())).max())
    parity_log("r101 mild 1066x1920 batch 8 (BASELINE configs[3]), frame 2 vs live oracle", dbox, dprob, None, 1e-3)
    assert dbox <= 1e-3 and dprob <= 2e-3


def test_config5_tiled_4k_frame(detectors):
    """BASELINE.json configs[4]: a 4K frame as 2 x 2 tiles of 1080p through detect_batch (device resize 1080x1920 -> 750x1333),
    merged on the host: equals the per-tile results shifted by the tile origins, de-duplicated across the seams."""
    from office_person_detection_vit_amd.tiling import TiledDetector, merge_tile_detections, split_tiles
    path = detectors(ga=1.0).model_path
    det = HipDetrDetector(model_path=path, max_batch=4, max_size=(800, 1333), resize=True)
    det.load_model()
    frame = structured_frames(1, 2160, 3840, seed=707)[0]
    tiled = TiledDetector(det, 2, 2, nms_threshold=0.4)
    got = tiled.detect(frame)
    tiles, origins = split_tiles(frame, 2, 2, tiled.overlap)
    assert [t.shape for t in tiles] == [(1215, 2160, 3)] * 4      # 1080p tiles + the 1/8 overlap band, same 16:9 -> 750x1333
    per_tile = [det.detect(t) for t in tiles]
    det.close()
    want = merge_tile_detections(per_tile, origins, 0.4, [t.shape[:2] for t in tiles], (2160, 3840))
    # batch of 4 tiles vs one tile at a time: same kernels per frame -> same records
    assert [(d.bbox, d.confidence) for d in got] == [(d.bbox, d.confidence) for d in want]
    for d in got:
        assert -3840 <= d.bbox[0] <= 3840 and -2160 <= d.bbox[1] <= 2160 and d.camera_coords == (d.bbox[0] + d.bbox[2] / 2, d.bbox[1] + d.bbox[3])
    assert sum(len(p) for p in per_tile) >= len(got)


def test_async_submission_equals_blocking_detect(detectors):
    """opd_detr_detect_async / opd_detr_wait (the pipelined bench loop): two submissions in flight on alternating device
    buffers give exactly the records of the blocking call, in submission order."""
    import ctypes as C
    from office_person_detection_vit_amd import _capi
    det = detectors(ga=1.0)
    lib = _capi.load_library()
    h = C.c_void_p(det.model)
    Q = 100
    sets = [np.ascontiguousarray(np.stack(structured_frames(2, 256, 320, seed=s))) for s in (41, 42)]
    d_frames = [torch.from_numpy(a).cuda() for a in sets]
    hw = np.asarray([[256, 320]] * 2, np.int32)
    want = []
    for a in sets:      # blocking reference (host buffers)
        recs, counts = (_capi.OpdDet * (2 * Q))(), (C.c_int32 * 2)()
        _capi.check(lib.opd_detr_detect(h, a.ctypes.data_as(C.c_void_p), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_HOST, 2, 256, 320,
                                        0.5, hw.ctypes.data_as(C.c_void_p), recs, counts), "detect")
        want.append((list(counts), bytes(recs)))
    bufs = [torch.zeros((2 * Q * 8 + 2,), dtype=torch.int32, device="cuda") for _ in range(2)]
    tickets = []
    for i in range(2):  # both submitted before either is waited for
        t = C.c_int()
        _capi.check(lib.opd_detr_detect_async(h, C.c_void_p(d_frames[i].data_ptr()), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_DEVICE, 2, 256, 320,
                                              0.5, hw.ctypes.data_as(C.c_void_p), C.cast(C.c_void_p(bufs[i].data_ptr()), C.POINTER(_capi.OpdDet)),
                                              C.cast(C.c_void_p(bufs[i][2 * Q * 8:].data_ptr()), C.POINTER(C.c_int32)), C.byref(t)),
                    "detect_async")
        tickets.append(t.value)
    for i in range(2):
        _capi.check(lib.opd_detr_wait(h, tickets[i]), "wait")
        host = bufs[i].cpu().numpy()
        counts = host[2 * Q * 8:].tolist()
        assert counts == want[i][0]
        got = host[:2 * Q * 8].reshape(2, Q, 8)
        ref = np.frombuffer(want[i][1], dtype=np.int32).reshape(2, Q, 8)
        for b in range(2):
            np.testing.assert_array_equal(got[b, :counts[b]], ref[b, :counts[b]])
    assert lib.opd_detr_wait(h, 7) != 0     # bad ticket -> error code, not a hang


def test_async_host_buffers_equal_blocking_detect(detectors):
    """Host-memory flavour of the asynchronous API: pixels staged by an async copy, records delivered by opd_detr_wait."""
    import ctypes as C
    from office_person_detection_vit_amd import _capi
    det = detectors(ga=1.0)
    lib = _capi.load_library()
    h = C.c_void_p(det.model)
    Q = 100
    sets = [np.ascontiguousarray(np.stack(structured_frames(2, 256, 320, seed=s))) for s in (51, 52, 53)]
    hw = np.asarray([[256, 320]] * 2, np.int32)
    want = []
    for a in sets:
        recs, counts = (_capi.OpdDet * (2 * Q))(), (C.c_int32 * 2)()
        _capi.check(lib.opd_detr_detect(h, a.ctypes.data_as(C.c_void_p), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_HOST, 2, 256, 320,
                                        0.5, hw.ctypes.data_as(C.c_void_p), recs, counts), "detect")
        want.append((list(counts), np.frombuffer(bytes(recs), dtype=np.int32).reshape(2, Q, 8).copy()))
    outs = [((_capi.OpdDet * (2 * Q))(), (C.c_int32 * 2)()) for _ in sets]
    tickets = []
    for a, (recs, counts) in zip(sets, outs):   # three in flight on one handle
        t = C.c_int()
        _capi.check(lib.opd_detr_detect_async(h, a.ctypes.data_as(C.c_void_p), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_HOST, 2, 256, 320,
                                              0.5, hw.ctypes.data_as(C.c_void_p), recs, counts, C.byref(t)), "detect_async")
        tickets.append(t.value)
    assert len(set(tickets)) == 3
    for i in (2, 0, 1):   # waiting out of order is allowed
        _capi.check(lib.opd_detr_wait(h, tickets[i]), "wait")
        recs, counts = outs[i]
        assert list(counts) == want[i][0]
        got = np.frombuffer(bytes(recs), dtype=np.int32).reshape(2, Q, 8)
        for b in range(2):
            np.testing.assert_array_equal(got[b, :counts[b]], want[i][1][b, :counts[b]])


def test_multi_handle_detect_batch_equals_serial(weight_cache, monkeypatch):
    """HipDetrDetector(streams=3): chunks of one detect_batch call overlap on three handles (worker threads, own HIP streams)
    and return exactly the serial detector's detections, frame for frame, for both the canvas and the device-resize path
    (the serial one stacks into pageable numpy memory, the other into the page-locked staging of ``opd_host_alloc``).
    A multi-stream detector may run stage 3 through other kernels than a single-stream one (OPD_FLAG_MULTI_STREAM: same arithmetic,
    another summation order inside one MFMA, i.e. last-bit differences); this test is about the plumbing, so both detectors are
    pinned to the same choice."""
    monkeypatch.setenv("OPD_TAIL3", "0")
    path = ensure_weight_file(weight_cache, DetrArch(), 0, 1.0, "r50")
    serial = HipDetrDetector(model_path=path, max_batch=2, max_size=(256, 320), resize=True, pinned_staging=False)
    multi = HipDetrDetector(model_path=path, max_batch=2, max_size=(256, 320), resize=True, streams=3)
    serial.load_model(); multi.load_model()
    try:
        for (h, w) in ((256, 320), (360, 640)):
            frames = structured_frames(11, h, w, seed=900 + h)
            want = serial.detect_batch(frames)
            got = multi.detect_batch(frames)
            assert len(got) == len(frames)
            assert [[(d.bbox, d.confidence, d.query_index) for d in f] for f in got] == \
                   [[(d.bbox, d.confidence, d.query_index) for d in f] for f in want]
    finally:
        serial.close(); multi.close()


def test_multi_stream_kernel_choice_stays_within_the_parity_bounds(weight_cache):
    """Without pinning: a ``streams=2`` detector (OPD_FLAG_MULTI_STREAM: stage 3 through the fused tail) against a single-stream one
    (three launches per block at this size).  The two differ like any two summation orders do (last bits of fp16 activations,
    amplified by the network): raw outputs within the small-frame parity bounds of each other, and not bit-identical."""
    path = ensure_weight_file(weight_cache, DetrArch(), 0, 1.0, "r50")
    serial = HipDetrDetector(model_path=path, max_batch=2, max_size=(256, 320), resize=False)
    multi = HipDetrDetector(model_path=path, max_batch=2, max_size=(256, 320), resize=False, streams=2)
    serial.load_model(); multi.load_model()
    try:
        frames = structured_frames(2, 256, 320, seed=4711)
        lg_s, bx_s, _ = serial.forward_raw(frames, want_encoder=False)
        lg_m, bx_m, _ = multi.forward_raw(frames, want_encoder=False)
        assert float(np.abs(bx_s - bx_m).max()) <= TOL[1.0][0] and float(np.abs(_softmax(lg_s) - _softmax(lg_m)).max()) <= TOL[1.0][1]
        assert float(np.abs(bx_s - bx_m).max()) > 0   # the flag really changed the launch sequence
    finally:
        serial.close(); multi.close()


def test_clone_shares_weights_and_outlives_its_source(weight_cache):
    """opd_detr_clone: a second handle on the same weights gives bit-identical records, reports the same weight bytes, and keeps
    working after the handle it was cloned from has been destroyed (shared ownership of the device buffers)."""
    import ctypes as C
    from office_person_detection_vit_amd import _capi
    lib = _capi.load_library()
    path = ensure_weight_file(weight_cache, DetrArch(), 0, 1.0, "r50")
    cfg = _capi.OpdConfig(struct_size=C.sizeof(_capi.OpdConfig), max_batch=2, max_height=256, max_width=320, flags=0)
    src, dup = C.c_void_p(), C.c_void_p()
    _capi.check(lib.opd_detr_create(C.byref(cfg), path.encode(), 0, C.byref(src)), "create")
    _capi.check(lib.opd_detr_clone(src, C.byref(dup)), "clone")
    i1, i2 = _capi.OpdModelInfo(), _capi.OpdModelInfo()
    _capi.check(lib.opd_detr_info(src, C.byref(i1)), "info"); _capi.check(lib.opd_detr_info(dup, C.byref(i2)), "info")
    assert i1.weight_bytes_device == i2.weight_bytes_device and i1.num_queries == i2.num_queries
    frames = np.ascontiguousarray(np.stack(structured_frames(2, 256, 320, seed=77)))
    hw = np.asarray([[256, 320]] * 2, np.int32)
    Q = i1.num_queries

    def run(h):
        recs, counts = (_capi.OpdDet * (2 * Q))(), (C.c_int32 * 2)()
        for _ in range(3):   # eager, capture, replay
            _capi.check(lib.opd_detr_detect(h, frames.ctypes.data_as(C.c_void_p), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_HOST, 2, 256, 320,
                                            0.5, hw.ctypes.data_as(C.c_void_p), recs, counts), "detect")
        c = list(counts)
        r = np.frombuffer(bytes(recs), dtype=np.int32).reshape(2, Q, 8)
        return c, [r[b, :c[b]].copy() for b in range(2)]

    c1, r1 = run(src)
    lib.opd_detr_destroy(src)
    c2, r2 = run(dup)
    lib.opd_detr_destroy(dup)
    assert c1 == c2 and sum(c1) > 0
    for a, b in zip(r1, r2):
        np.testing.assert_array_equal(a, b)


def test_batch8_full_size_frames_match_live_oracle(detectors, parity_log):
    """BASELINE configs[1] (batch 8, 800x1333): three frames of ONE batch-8 forward against the oracle run live on the same
    frames, at the north-star tolerance (|dbox| <= 1e-3), plus the post-processed person boxes in pixels."""
    det = detectors(ga=1.0, max_batch=8)
    frames = structured_frames(8, 800, 1333, seed=8800)
    lg8, bx8, enc8 = det.forward_raw(frames)
    dets8 = det.detect_batch(frames)
    w = O.to_torch(load_safetensors(det.model_path))
    compared = 0
    for i in (0, 3, 7):
        pv, pm = O.preprocess([frames[i]])
        lg, bx, mem = O.forward(w, pv, pm)
        dbox = float(np.abs(bx8[i] - bx[0].numpy()).max())
        dprob = float(np.abs(_softmax(lg8[i]) - _softmax(lg[0].numpy())).max())
        denc = float(np.abs(enc8[i] - mem[0].numpy()).max())
        parity_log(f"r50 mild 800x1333 batch 8 (BASELINE configs[1]), frame {i} vs live oracle", dbox, dprob, denc, 1e-3)
        assert dbox <= 1e-3 and dprob <= 2e-3 and denc <= 3e-2
        want = O.person_detections(O.post_process_object_detection(lg.numpy(), bx.numpy(), 0.5, [(800, 1333)])[0], 0.4)
        ref_q = {d["query_index"]: d for d in want if abs(d["confidence"] - 0.5) > 4e-3}
        got_q = {d.query_index: d for d in dets8[i] if abs(d.confidence - 0.5) > 4e-3}
        assert set(ref_q) == set(got_q)
        compared += len(ref_q)
        for qi, r in ref_q.items():
            np.testing.assert_allclose(got_q[qi].bbox, r["bbox"], atol=1e-3 * 1333 * 2)   # x and w each within 1e-3 of the width
    assert compared > 0


def test_export_scored_like_oracle(detectors, tmp_path):
    """SURVEY.md section 8(f) rank 3 on the GPU: detector -> `detections_to_coco` -> `write_coco` -> evaluator, scored against
    detections derived from the ORACLE on the same frames as ground truth.  Away from the score threshold every oracle box is
    matched at IoU >= 0.9 (TP) and nothing else is reported (no FP / FN); the ambiguous ones can only add one FP or FN each."""
    import json
    from evaluator_checker import DetectionEvaluator
    from office_person_detection_vit_amd import detections_to_coco, write_coco
    det = detectors(ga=1.0)
    frames = structured_frames(2, 288, 352, seed=3100)
    dets = det.detect_batch(frames)
    w = O.to_torch(load_safetensors(det.model_path))
    pv, pm = O.preprocess(frames)
    lg, bx, _ = O.forward(w, pv, pm)
    ref = [O.person_detections(r, 0.4) for r in O.post_process_object_detection(lg.numpy(), bx.numpy(), 0.5, [(288, 352)] * 2)]
    sizes = [(288, 352)] * 2
    path = tmp_path / "pred" / "detections.json"
    write_coco(str(path), detections_to_coco(dets, sizes))
    pred = json.load(open(path, encoding="utf-8"))
    gt = {"images": pred["images"], "categories": pred["categories"],
          "annotations": [{"id": k, "image_id": i, "category_id": 0, "bbox": list(d["bbox"]), "area": d["bbox"][2] * d["bbox"][3], "iscrowd": 0}
                          for i, f in enumerate(ref) for k, d in enumerate(f)]}
    for k, a in enumerate(gt["annotations"]):
        a["id"] = k
    ambiguous = sum(abs(d["confidence"] - 0.5) <= 8e-3 for f in ref for d in f) + sum(abs(d.confidence - 0.5) <= 8e-3 for f in dets for d in f)
    m = DetectionEvaluator(iou_threshold=0.9, confidence_threshold=0.0).evaluate(gt, pred)
    n_ref = sum(len(f) for f in ref)
    assert n_ref > 0 and m.true_positives >= n_ref - ambiguous
    assert m.false_positives <= ambiguous and m.false_negatives <= ambiguous
    assert len(pred["annotations"]) == sum(len(f) for f in dets)


def _to_4x_name(k):
    """HF 5.x state-dict name -> the 4.x / timm name of the same tensor (inverse of csrc/opd_loader.cpp::normalise_key;
    HF:conversion_mapping.py:1036-1041 + the timm ResNet layout of `use_timm_backbone=True` checkpoints)."""
    import re
    bb = "model.backbone.model."
    if k.startswith(bb):
        r = k[len(bb):]
        r = r.replace("embedder.embedder.convolution.", "conv1.").replace("embedder.embedder.normalization.", "bn1.")
        m = re.match(r"encoder\.stages\.(\d)\.layers\.(\d+)\.(.*)", r)
        if m:
            st, ly, rest = int(m.group(1)), m.group(2), m.group(3)
            mm = re.match(r"layer\.(\d)\.(convolution|normalization)\.(.*)", rest)
            if mm:
                rest = f"{'conv' if mm.group(2) == 'convolution' else 'bn'}{int(mm.group(1)) + 1}.{mm.group(3)}"
            else:
                rest = rest.replace("shortcut.convolution.", "downsample.0.").replace("shortcut.normalization.", "downsample.1.")
            r = f"layer{st + 1}.{ly}.{rest}"
        return "model.backbone.conv_encoder.model." + r
    k = k.replace(".o_proj.", ".out_proj.")
    if k.startswith(("model.encoder.layers.", "model.decoder.layers.")):
        k = k.replace(".mlp.fc1.", ".fc1.").replace(".mlp.fc2.", ".fc2.")
    return k


def test_checkpoint_with_4x_timm_names_gives_identical_outputs(detectors, tmp_path):
    """a checkpoint in the reference pin's naming (transformers 4.57 + timm backbone: `conv_encoder`,
    `out_proj`, `fc1/fc2`, `layer1.0.conv1`, `downsample.0`) loads as is and gives bit-identical outputs."""
    from office_person_detection_vit_amd.weights import save_safetensors
    det = detectors(ga=1.0)
    w = load_safetensors(det.model_path)
    old = {_to_4x_name(k): v for k, v in w.items()}
    assert len(old) == len(w) and "model.backbone.conv_encoder.model.layer3.5.bn2.running_var" in old
    assert "model.decoder.layers.2.encoder_attn.out_proj.weight" in old and "model.encoder.layers.0.fc1.weight" in old
    assert not any(".o_proj." in k or ".mlp." in k or "embedder" in k for k in old)
    path = str(tmp_path / "detr_4x_names.safetensors")
    save_safetensors(old, path)
    frames = structured_frames(2, 256, 320, seed=1212)
    outs = []
    for pth in (det.model_path, path):
        d = HipDetrDetector(model_path=pth, max_batch=2, max_size=(800, 1333), resize=False)
        d.load_model()
        try:
            outs.append(d.forward_raw(frames))
        finally:
            d.close()
    for x, y in zip(*outs):
        np.testing.assert_array_equal(x, y)


def test_long_lived_handle_equals_fresh_handle(detectors):
    """The module's shared detector has by now served dozens of calls (graphs captured and replayed, ragged batches, ROI pooling,
    asynchronous submissions): its outputs must still be bit-identical to a fresh handle's."""
    det = detectors(ga=1.0)
    frames = structured_frames(2, 256, 320, seed=1213)
    fresh = HipDetrDetector(model_path=det.model_path, max_batch=2, max_size=(800, 1333), resize=False)
    fresh.load_model()
    try:
        want = fresh.forward_raw(frames)
    finally:
        fresh.close()
    for rep in range(3):   # eager / captured / replayed on the shared handle (or replayed thrice if this shape was seen before)
        got = det.forward_raw(frames)
        for x, y in zip(got, want):
            np.testing.assert_array_equal(x, y)


def test_sharded_detector_device_direct_exchange(detectors):
    """`ShardedDetector.detect_batch` with the exchange buffer in HBM (world size 1 RCCL group on this GPU): the post-process kernel
    writes the records into the tensor the all-gather reads (OPD_MEM_HOST_PIXELS_DEVICE_OUT); same detections as detect_batch,
    for both the plain and the device-resize path."""
    import socket
    import torch.distributed as dist
    from office_person_detection_vit_amd.sharding import ShardedDetector
    det = detectors(ga=1.0)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sharded = ShardedDetector(det, device="cuda:0")
        for frames in (structured_frames(5, 256, 320, seed=77), ):
            want = det.detect_batch(frames)
            got = sharded.detect_batch(frames)
            assert [[(d.bbox, d.confidence, d.query_index) for d in f] for f in got] == \
                   [[(d.bbox, d.confidence, d.query_index) for d in f] for f in want]
            assert sum(len(f) for f in got) > 0
    finally:
        dist.destroy_process_group()
    cam = HipDetrDetector(model_path=det.model_path, max_batch=2, max_size=(256, 320), resize=True)
    cam.load_model()
    try:   # camera-resolution frames: resize on the device, records into a device buffer
        frames = structured_frames(2, 360, 450, seed=78)
        rec = torch.zeros((2, cam.num_queries, 8), dtype=torch.int32, device="cuda")
        cnt = torch.full((2,), -1, dtype=torch.int32, device="cuda")
        cam.detect_records_into(frames, rec, cnt)
        recs, counts, Q = cam._detect_records(frames)
        assert cnt.cpu().tolist() == list(counts)
        rec_h, cnt_h = torch.zeros((2, Q, 8), dtype=torch.int32), torch.zeros((2,), dtype=torch.int32)   # host exchange buffer (gloo)
        cam.detect_records_into(frames, rec_h, cnt_h)
        assert cnt_h.tolist() == list(counts) and torch.equal(rec_h[0, :counts[0]], rec[0, :counts[0]].cpu())
        host = np.frombuffer(bytes(recs), dtype=np.int32).reshape(2, Q, 8)
        for b in range(2):
            np.testing.assert_array_equal(rec[b, :counts[b]].cpu().numpy(), host[b, :counts[b]])
    finally:
        cam.close()


def test_fifth_async_submission_is_refused(detectors):
    """a submission slot is not reused before its ticket has been waited for."""
    import ctypes as C
    from office_person_detection_vit_amd import _capi
    det = detectors(ga=1.0)
    lib = _capi.load_library()
    h = C.c_void_p(det.model)
    Q = det.num_queries
    frames = np.ascontiguousarray(np.stack(structured_frames(1, 256, 320, seed=5)))
    hw = np.asarray([[256, 320]], np.int32)
    outs = [((_capi.OpdDet * Q)(), (C.c_int32 * 1)()) for _ in range(5)]
    tickets = []
    for i, (recs, counts) in enumerate(outs):
        t = C.c_int(-1)
        rc = lib.opd_detr_detect_async(h, frames.ctypes.data_as(C.c_void_p), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_HOST, 1, 256, 320,
                                       0.5, hw.ctypes.data_as(C.c_void_p), recs, counts, C.byref(t))
        if i < 4:
            assert rc == 0
            tickets.append(t.value)
        else:
            assert rc == -6 and "outstanding" in _capi.last_error()     # OPD_ESTATE
    assert sorted(tickets) == [0, 1, 2, 3]
    _capi.check(lib.opd_detr_wait(h, tickets[0]), "wait")
    assert lib.opd_detr_wait(h, tickets[0]) == -6                       # waiting twice for one ticket
    t = C.c_int(-1)
    _capi.check(lib.opd_detr_detect_async(h, frames.ctypes.data_as(C.c_void_p), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_HOST, 1, 256, 320,
                                          0.5, hw.ctypes.data_as(C.c_void_p), outs[4][0], outs[4][1], C.byref(t)), "detect_async")
    for tk in tickets[1:] + [t.value]:
        _capi.check(lib.opd_detr_wait(h, tk), "wait")
    assert [list(c) for _, c in outs] == [list(outs[0][1])] * 5         # five runs of the same frame


def test_portrait_frames_are_accepted(weight_cache, parity_log):
    """the HF size rule maps a portrait camera frame to a portrait model input; a handle configured for 256 x 320 takes
    320 x 256 too, and the result matches the oracle like the landscape case."""
    path = ensure_weight_file(weight_cache, DetrArch(), 0, 1.0, "r50")
    det = HipDetrDetector(model_path=path, max_batch=1, max_size=(256, 320), resize=False)
    det.load_model()
    try:
        frame = structured_frames(1, 320, 256, seed=640)[0]
        logits, boxes, enc = det.forward_raw([frame])
        w = O.to_torch(load_safetensors(path))
        pv, pm = O.preprocess([frame])
        lg, bx, mem = O.forward(w, pv, pm)
        dbox = float(np.abs(boxes - bx.numpy()).max())
        parity_log("r50 mild 320x256 (portrait) vs live oracle", dbox, None, float(np.abs(enc - mem.numpy()).max()), 2e-3)
        assert dbox <= 2e-3
        with pytest.raises(RuntimeError):
            det.forward_raw([structured_frames(1, 330, 256, seed=1)[0]])     # more pixels than the handle was sized for
    finally:
        det.close()


def test_graph_replay_survives_handle_churn(weight_cache):
    """Regression (round 2): a handle's captured graph must still be right after OTHER handles have been destroyed and created with
    different weights in the memory they returned.  On ROCm 7.2 the replay then gave NaN although the same launches issued eagerly
    stayed bit-exact; the library now re-captures its graphs whenever a handle has come or gone (csrc/opd_model.cpp::g_handle_epoch)."""
    mild = ensure_weight_file(weight_cache, DetrArch(), 0, 1.0, "r50")
    sharp = ensure_weight_file(weight_cache, DetrArch(), 0, 2.0, "r50")
    mk = lambda p: HipDetrDetector(model_path=p, max_batch=2, max_size=(256, 320), resize=False)
    frames = structured_frames(2, 256, 320, seed=4321)
    a = mk(mild); a.load_model()
    try:
        a.forward_raw(structured_frames(2, 256, 320, seed=1))   # eager
        ref = a.forward_raw(frames)                             # captured
        f = mk(mild); f.load_model(); f.forward_raw(frames); f.close()
        b = mk(sharp); b.load_model()
        try:
            b.forward_raw(frames)
            for _ in range(2):
                got = a.forward_raw(frames)
                for x, y in zip(got, ref):
                    np.testing.assert_array_equal(x, y)
        finally:
            b.close()
    finally:
        a.close()
