"""CPU-only tests: the C-ABI library loads and exports every symbol ``include/opd_detr.h`` declares, host-side loader
logic (fp16 conversion, checkpoint key normalisation, native safetensors parsing + schema check), the detector's
host pre-processing against HF golden vectors, and the reference's error conventions.  No compute calls."""

import ctypes as C
import os
import re

import numpy as np
import pytest

from office_person_detection_vit_amd import HipDetrDetector, _capi, model_input_size
from office_person_detection_vit_amd.detector import resize_frame
from office_person_detection_vit_amd.frames import structured_frames
from office_person_detection_vit_amd.weights import (DetrArch, param_specs, rename_4x_key, save_safetensors,
                                                     synth_weights)
from oracle import detr_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "opd_detr.h")).read()
    declared = set(re.findall(r"\b(opd_[a-z0-9_]+)\s*\(", header))
    declared -= {"opd_config", "opd_det", "opd_model_info"}
    assert len(declared) >= 12
    lib = _capi.load_library()
    for name in sorted(declared):
        assert hasattr(lib, name), f"libopd_hip.so does not export {name}"
        assert name in _capi.API, f"_capi.API lacks a prototype for {name}"
    assert set(_capi.API) == declared
    assert b"gfx950" in lib.opd_version()


def test_product_library_exports_exactly_the_header():
    """libopd_hip.so (the product) exports the functions of include/opd_detr.h and nothing else of ours: no opd_test_* hook, no
    launcher, no C++ helper (the test API used to ship inside the product library).  The hooks live in
    libopd_hip_test.so, which tests/ and tools/ load instead (conftest.py sets OPD_TEST_HOOKS=1)."""
    import subprocess
    header = open(os.path.join(ROOT, "include", "opd_detr.h")).read()
    declared = set(re.findall(r"\b(opd_[a-z0-9_]+)\s*\(", header)) - {"opd_config", "opd_det", "opd_model_info"}

    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
        return {l.split()[2] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] == "T"}

    prod = exported(_capi.LIB_PATH)
    assert {n for n in prod if "opd" in n.lower()} == declared, sorted(prod ^ declared)[:10]
    test = exported(_capi.TEST_LIB_PATH)
    assert declared <= test and set(_capi.TEST_API) <= test
    raw = C.CDLL(_capi.LIB_PATH)   # (a second, prototype-less handle: does not disturb the process's _capi library)
    assert not hasattr(raw, "opd_test_set_graph_guard") and not hasattr(raw, "opd_launch_conv_gemm")


def test_weight_rounding_by_error_diffusion():
    """csrc/opd_host.cpp::round_f16_diffused: every value lands on an fp16 value next to it, the errors of a row sum to at most
    half an ulp of its largest weight (round-to-nearest: a random walk), fp16-exact input is untouched, and the result is the
    sequential carry algorithm visiting (channel outer, tap inner)."""
    lib = _capi.load_library()
    rng = np.random.default_rng(5)
    rows, taps, cin = 7, 9, 64
    w = (rng.standard_normal((rows, taps, cin)) * 0.02).astype(np.float32)
    got = w.copy()
    assert lib.opd_test_round_f16_diffused(got.ctypes.data_as(C.c_void_p), rows, taps, cin) == 0
    want = np.empty_like(w)
    for r in range(rows):
        carry = 0.0
        for c in range(cin):
            for t in range(taps):
                target = float(w[r, t, c]) + carry
                q = float(np.float32(target).astype(np.float16))
                want[r, t, c] = q
                carry = target - q
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(got, got.astype(np.float16).astype(np.float32))                    # fp16 values
    ulp_max = np.spacing(np.abs(w).max(axis=(1, 2)).astype(np.float16)).astype(np.float64)
    assert np.all(np.abs(got.astype(np.float64) - w) <= ulp_max[:, None, None])   # own half-ulp + the carried half-ulp of a larger neighbour
    assert np.all(np.abs((got.astype(np.float64) - w).sum(axis=(1, 2))) <= 0.5 * ulp_max + 1e-12)
    rtn = w.astype(np.float16).astype(np.float32)
    assert np.abs((rtn.astype(np.float64) - w).sum(axis=(1, 2))).mean() > 4 * np.abs((got.astype(np.float64) - w).sum(axis=(1, 2))).mean()
    exact = rtn.copy()
    assert lib.opd_test_round_f16_diffused(exact.ctypes.data_as(C.c_void_p), rows, taps, cin) == 0
    np.testing.assert_array_equal(exact, rtn)


def test_struct_layouts_match_header():
    assert C.sizeof(_capi.OpdDet) == 32
    assert C.sizeof(_capi.OpdConfig) == 32
    assert C.sizeof(_capi.OpdModelInfo) == 4 * 4 + 7 * 4 + 3 * 4 + 4 + 4 + 16  # incl. 4 bytes padding before int64


def test_f16_conversion_matches_numpy():
    lib = _capi.load_library()
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.standard_normal(2000).astype(np.float32) * s for s in (1e-7, 1e-4, 1.0, 300.0, 7e4)] +
                          [np.array([0.0, -0.0, 65504.0, 65520.0, 65519.9, 6.1e-5, 5.96e-8, 2.98e-8, 2.99e-8], np.float32)])
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).view(np.uint16)
    got = np.array([lib.opd_test_f32_to_f16(float(v)) for v in vals], np.uint16)
    np.testing.assert_array_equal(got, want)
    back = np.array([lib.opd_test_f16_to_f32(int(h)) for h in want[:3000]], np.float32)
    np.testing.assert_array_equal(back, want[:3000].view(np.float16).astype(np.float32))


def test_checkpoint_key_normalisation():
    lib = _capi.load_library()
    buf = C.create_string_buffer(512)
    cases = {
        "model.backbone.conv_encoder.model.conv1.weight": "model.backbone.model.embedder.embedder.convolution.weight",
        "model.backbone.conv_encoder.model.layer3.5.bn2.running_mean":
            "model.backbone.model.encoder.stages.2.layers.5.layer.1.normalization.running_mean",
        "model.backbone.conv_encoder.model.layer2.0.downsample.0.weight":
            "model.backbone.model.encoder.stages.1.layers.0.shortcut.convolution.weight",
        "model.decoder.layers.2.encoder_attn.out_proj.bias": "model.decoder.layers.2.encoder_attn.o_proj.bias",
        "model.encoder.layers.4.fc2.weight": "model.encoder.layers.4.mlp.fc2.weight",
        "bbox_predictor.layers.1.weight": "bbox_predictor.layers.1.weight",
    }
    for k, want in cases.items():
        assert lib.opd_test_normalise_key(k.encode(), buf, 512) == 0
        assert buf.value.decode() == want
    # the python mirror used by tools agrees on the 4.x transformer renames
    assert rename_4x_key("model.decoder.layers.2.encoder_attn.out_proj.bias") == "model.decoder.layers.2.encoder_attn.o_proj.bias"
    # 5.x names are fixed points
    for name, _, _ in list(param_specs(DetrArch()))[::37]:
        lib.opd_test_normalise_key(name.encode(), buf, 512)
        assert buf.value.decode() == name


def test_native_checkpoint_parse_and_schema(tmp_path):
    lib = _capi.load_library()
    arch = DetrArch(depths=(1, 2, 1, 1), encoder_layers=2, decoder_layers=1, num_queries=20)
    w = synth_weights(arch, 3, 1.0, calibrate=False)
    path = str(tmp_path / "tiny.safetensors")
    save_safetensors(w, path)
    info = (C.c_int32 * 8)()
    assert lib.opd_test_inspect_checkpoint(path.encode(), info) == 0, _capi.last_error()
    assert list(info) == [1, 2, 1, 1, 2, 1, 20, 92]
    # a missing tensor is a schema error naming the tensor
    w2 = dict(w)
    del w2["model.decoder.layers.0.encoder_attn.k_proj.bias"]
    bad = str(tmp_path / "bad.safetensors")
    save_safetensors(w2, bad)
    assert lib.opd_test_inspect_checkpoint(bad.encode(), info) == -3
    assert "encoder_attn.k_proj.bias" in _capi.last_error()
    # wrong shape
    w3 = dict(w)
    w3["model.input_projection.weight"] = np.zeros((256, 1024, 1, 1), np.float32)
    bad3 = str(tmp_path / "bad3.safetensors")
    save_safetensors(w3, bad3)
    assert lib.opd_test_inspect_checkpoint(bad3.encode(), info) == -3
    assert lib.opd_test_inspect_checkpoint(str(tmp_path / "nope.safetensors").encode(), info) == -2


def test_size_rule_and_resize_match_hf(golden_dir):
    """``model_input_size`` + PIL bilinear ``resize_frame`` + the oracle's normalisation reproduce HF's image processor."""
    g = np.load(os.path.join(golden_dir, "hf_resize.npz"))
    for tag, (h, w) in {"720x1280": (720, 1280), "480x640": (480, 640), "1080x1920": (1080, 1920), "900x700": (900, 700)}.items():
        th, tw = model_input_size(h, w)
        assert (th, tw) == tuple(int(v) for v in g[f"{tag}_shape"][2:])
        frame = structured_frames(1, h, w, seed=555)[0]
        pv, _ = O.preprocess([resize_frame(frame, th, tw)])
        np.testing.assert_allclose(pv[:, :, ::53, ::59].numpy(), g[f"{tag}_sample"], atol=1e-6)
    assert model_input_size(800, 1333) == (800, 1333)  # the benchmark frames need no resize


def test_error_conventions_without_gpu(tmp_path):
    """Reference conventions (tests/test_yolov8_detector.py:108-119): detect before load -> 'Model not loaded';
    load failure -> RuntimeError 'Failed to load ... model'.  On a GPU-less host load_model must FAIL, not fall back."""
    det = HipDetrDetector(model_path=str(tmp_path / "missing.safetensors"), confidence_threshold=0.5)
    assert det.model is None and det.feature_extractor is not None and det.confidence_threshold == 0.5
    with pytest.raises(RuntimeError, match="Model not loaded"):
        det.detect(np.zeros((720, 1280, 3), np.uint8))
    with pytest.raises(RuntimeError, match="Model not loaded"):
        det.detect_with_features(np.zeros((720, 1280, 3), np.uint8))
    with pytest.raises(RuntimeError, match="Failed to load DETR model"):
        det.load_model()
    assert det._get_foot_position((100.0, 200.0, 50.0, 100.0)) == (125.0, 300.0)
    with pytest.raises(ValueError):
        HipDetrDetector(device="cpu")
    assert HipDetrDetector(device="cuda").device == "hip:0" and HipDetrDetector(device="hip:3").device_ordinal == 3


def test_no_cpu_fallback_on_gpuless_host(tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    w = synth_weights(DetrArch(depths=(1, 1, 1, 1), encoder_layers=1, decoder_layers=1, num_queries=20), 0, 1.0, calibrate=False)
    path = str(tmp_path / "tiny.safetensors")
    save_safetensors(w, path)
    det = HipDetrDetector(model_path=path)
    with pytest.raises(RuntimeError, match="Failed to load DETR model.*(no HIP device|no CPU fallback|hip)"):
        det.load_model()


def test_person_nms_host_routine():
    """``opd_person_nms`` against the oracle's ``person_detections`` on random boxes (host code, no GPU needed)."""
    from office_person_detection_vit_amd.sharding import DET_DTYPE

    lib = _capi.load_library()
    rng = np.random.default_rng(12)
    for trial in range(20):
        n = int(rng.integers(0, 60))
        xy = rng.uniform(0, 500, (n, 2)).astype(np.float32)
        wh = rng.uniform(20, 300, (n, 2)).astype(np.float32)
        recs = np.zeros(max(n, 1), DET_DTYPE)
        recs["x1"][:n], recs["y1"][:n] = xy[:, 0], xy[:, 1]
        recs["x2"][:n], recs["y2"][:n] = xy[:, 0] + wh[:, 0], xy[:, 1] + wh[:, 1]
        recs["score"][:n] = rng.uniform(0.5, 1.0, n).astype(np.float32)
        recs["label"][:n] = rng.choice([1, 1, 1, 3, 17], n)
        recs["query_index"][:n] = np.arange(n)
        res = {"scores": recs["score"][:n].copy(), "labels": recs["label"][:n].copy(),
               "boxes": np.stack([recs["x1"][:n], recs["y1"][:n], recs["x2"][:n], recs["y2"][:n]], 1),
               "query_index": np.arange(n)}
        want = [d["query_index"] for d in O.person_detections(res, 0.4)]
        kept = lib.opd_person_nms(recs.ctypes.data_as(C.POINTER(_capi.OpdDet)), n, 1, 0.4)
        assert kept == len(want)
        assert list(recs["query_index"][:kept]) == want


# ---- ragged batches (padding mask path): host-side pieces ------------------------------------------------------------------
def test_ragged_batch_canvas_and_valid_sizes():
    """Frames of different model-input sizes -> one zero canvas of the batch maximum + per-frame valid sizes, like HF's
    ``DetrImageProcessor.pad`` (image_processing_detr.py:639-668); equal sizes -> plain stack, no mask."""
    det = HipDetrDetector(model_path="unused.safetensors", resize=False, max_size=(64, 96))
    a, b = structured_frames(1, 40, 60, seed=1)[0], structured_frames(1, 33, 72, seed=2)[0]
    canvas, orig, valid, target = det._preprocess_batch([a, b])
    assert target is None
    assert canvas.shape == (2, 40, 72, 3) and orig == [(40, 60), (33, 72)]
    np.testing.assert_array_equal(valid, [[40, 60], [33, 72]])
    np.testing.assert_array_equal(canvas[0, :, :60], a)
    np.testing.assert_array_equal(canvas[1, :33], b)
    assert not canvas[0, :, 60:].any() and not canvas[1, 33:].any()
    same, _, none, target = det._preprocess_batch([a, a])
    assert none is None and target is None and same.shape == (2, 40, 60, 3)
    # one camera size that needs resizing: the batch stays at camera resolution, the target size follows the HF rule
    cam = HipDetrDetector(model_path="unused.safetensors", resize=True)
    f = structured_frames(1, 72, 128, seed=4)[0]
    batch, orig, valid, target = cam._preprocess_batch([f, f])
    assert batch.shape == (2, 72, 128, 3) and valid is None and target == model_input_size(72, 128) == (750, 1333)
    host = HipDetrDetector(model_path="unused.safetensors", resize=True, device_resize=False)
    batch, _, _, target = host._preprocess_batch([f])
    assert target is None and batch.shape == (1, 750, 1333, 3)
    with pytest.raises(ValueError):
        det._preprocess_batch([a, structured_frames(1, 100, 40, seed=3)[0]])   # canvas edge longer than max(max_size)
    with pytest.raises(ValueError):
        det._preprocess_batch([b, structured_frames(1, 90, 60, seed=3)[0]])    # canvas 90 x 72: more pixels than 64 x 96


def test_mask_downsampling_matches_torch_nearest():
    """valid_prefix == number of True positions of F.interpolate(mask, size) (nearest), HF:modeling_detr.py:283-289."""
    import torch
    import torch.nn.functional as F
    lib = _capi.load_library()
    for size, out in [(256, 8), (224, 7), (800, 25), (1333, 42), (750, 24), (203, 7), (333, 11), (1066, 34)]:
        for valid in {1, 2, size // 3, size // 2, size - 33, size - 1, size}:
            if valid < 1:
                continue
            m = torch.zeros(1, 1, size, 1)
            m[:, :, :valid] = 1
            want = int(F.interpolate(m, size=(out, 1)).bool().sum())
            assert lib.opd_test_valid_prefix(valid, size, out) == want, (size, out, valid)


def test_masked_sine_position_embedding_matches_oracle():
    """Position embedding of a frame whose valid region is a top-left rectangle of the map: equals the oracle's
    cumulative-sum formulation (HF:modeling_detr.py:294-368) at EVERY position, padded ones included."""
    import torch
    lib = _capi.load_library()
    for h, w, vh, vw in [(8, 10, 8, 10), (8, 10, 7, 9), (25, 42, 24, 42), (25, 42, 25, 30), (7, 11, 1, 1)]:
        mask = torch.zeros(1, h, w, dtype=torch.bool)
        mask[:, :vh, :vw] = True
        want = O.sine_position_embedding(mask, 256)[0].numpy()
        got = np.empty((h * w, 256), np.float32)
        _capi.check(lib.opd_test_sine_pos_embed(h, w, vh, vw, 256, got.ctypes.data_as(C.c_void_p)), "sine_pos_embed")
        np.testing.assert_allclose(got, want, atol=2e-6)


# ---- device-side resize: the integer algorithm and its coefficient tables against Pillow (host emulation) ---------------------
def _resize_tables(lib, n_in, n_out):
    bounds = np.zeros((n_out, 2), np.int32)
    coeffs = np.zeros(n_out * 64, np.int32)
    ks = lib.opd_test_resize_coeffs(n_in, n_out, bounds.ctypes.data_as(C.c_void_p), coeffs.ctypes.data_as(C.c_void_p), coeffs.size)
    assert ks > 0
    return bounds, coeffs[:n_out * ks].reshape(n_out, ks)


def _emulate_resize(lib, img, oh, ow):
    """numpy restatement of resize_bilinear_u8_kernel: horizontal pass rounded to uint8, then vertical pass."""
    h, w, _ = img.shape
    bh, kh = _resize_tables(lib, w, ow)
    bv, kv = _resize_tables(lib, h, oh)
    half = 1 << 21
    tmp = np.empty((h, ow, 3), np.int64)
    for xo in range(ow):
        x0, n = int(bh[xo, 0]), int(bh[xo, 1])
        tmp[:, xo] = np.clip((half + (img[:, x0:x0 + n].astype(np.int64) * kh[xo, :n, None]).sum(1)) >> 22, 0, 255)
    out = np.empty((oh, ow, 3), np.uint8)
    for yo in range(oh):
        y0, n = int(bv[yo, 0]), int(bv[yo, 1])
        out[yo] = np.clip((half + (tmp[y0:y0 + n] * kv[yo, :n, None, None]).sum(0)) >> 22, 0, 255)
    return out


@pytest.mark.parametrize("src,dst", [((72, 128), (75, 133)), ((90, 160), (75, 133)), ((48, 64), (80, 107)), ((75, 133), (75, 133)),
                                     ((100, 37), (133, 49)), ((211, 97), (60, 28))])
def test_resize_algorithm_is_bit_exact_with_pillow(src, dst):
    """Up- and down-scaling, odd sizes, identity: the fixed-point two-pass algorithm of the device kernel, driven by the
    library's own coefficient tables, reproduces PIL.Image.resize(BILINEAR) bit for bit."""
    from PIL import Image
    lib = _capi.load_library()
    rng = np.random.default_rng(src[0] * 7 + dst[1])
    img = rng.integers(0, 256, (src[0], src[1], 3), dtype=np.uint8)
    want = np.asarray(Image.fromarray(img).resize((dst[1], dst[0]), resample=Image.BILINEAR))
    np.testing.assert_array_equal(_emulate_resize(lib, img, dst[0], dst[1]), want)


def test_constructor_takes_the_detection_phase_keywords():
    """``DetectionPhase.initialize`` builds its detector with exactly these keywords (reference
    ``src/pipeline/phases/detection.py:47-52``): model_path, confidence_threshold, device, iou_threshold."""
    det = HipDetrDetector(model_path="weights/model.safetensors", confidence_threshold=0.6, device="cuda", iou_threshold=0.45)
    assert det.nms_threshold == 0.45 and det.iou_threshold == 0.45 and det.confidence_threshold == 0.6 and det.device == "hip:0"
    assert HipDetrDetector(nms_threshold=0.3).iou_threshold == 0.3                     # the DETR-era keyword still works
    assert HipDetrDetector(nms_threshold=0.3, iou_threshold=0.5).nms_threshold == 0.5  # iou_threshold wins when both are given


def test_schema_check_covers_every_tensor_the_loader_reads(tmp_path):
    """a checkpoint that lacks a stem / shortcut normalisation tensor must be OPD_ESCHEMA (-3) from the schema check,
    not a C++ exception inside opd_detr_create; malformed shapes in the header are refused before anything is allocated."""
    import json
    import struct

    lib = _capi.load_library()
    arch = DetrArch(depths=(1, 1, 1, 1), encoder_layers=1, decoder_layers=1, num_queries=20)
    w = synth_weights(arch, 3, 1.0, calibrate=False)
    info = (C.c_int32 * 8)()
    for missing in ("model.backbone.model.embedder.embedder.normalization.running_mean",
                    "model.backbone.model.encoder.stages.2.layers.0.shortcut.normalization.weight",
                    "model.backbone.model.encoder.stages.0.layers.0.shortcut.normalization.bias"):
        w2 = dict(w)
        del w2[missing]
        bad = str(tmp_path / "missing.safetensors")
        save_safetensors(w2, bad)
        assert lib.opd_test_inspect_checkpoint(bad.encode(), info) == -3
        assert missing in _capi.last_error()
    for shape in ([-4, 4], [1 << 40, 1 << 30]):
        hdr = json.dumps({"t": {"dtype": "F32", "shape": shape, "data_offsets": [0, 64]}}).encode()
        bad = str(tmp_path / "shape.safetensors")
        with open(bad, "wb") as f:
            f.write(struct.pack("<Q", len(hdr)) + hdr + b"\0" * 64)
        assert lib.opd_test_inspect_checkpoint(bad.encode(), info) == -2
        assert "invalid shape" in _capi.last_error()


def test_portrait_frames_fit_the_configured_maximum():
    """the HF size rule maps a portrait camera frame to about 1333 x 750; the handle accepts either orientation
    (include/opd_detr.h, opd_config), and the shim's ragged-canvas check uses the same rule."""
    assert model_input_size(1280, 720) == (1333, 750)
    det = HipDetrDetector(max_size=(800, 1333), pinned_staging=False)
    frames = [np.zeros((1280, 720, 3), np.uint8), np.zeros((1000, 720, 3), np.uint8)]   # -> 1333x750 and 1111x800
    canvas, orig, valid, target = det._preprocess_batch(frames)
    assert canvas.shape == (2, 1333, 800, 3) and valid.tolist() == [[1333, 750], [1111, 800]] and target is None
    with pytest.raises(ValueError, match="exceeds the configured maximum"):
        HipDetrDetector(max_size=(640, 640), pinned_staging=False, resize=False)._preprocess_batch(
            [np.zeros((700, 300, 3), np.uint8), np.zeros((300, 500, 3), np.uint8)])


def test_feature_extractor_matches_reference_fixture(golden_dir):
    """The product-side ``FeatureExtractor`` (summed-area-table pooling) against the reference class's captured outputs."""
    from office_person_detection_vit_amd import FeatureExtractor
    g = np.load(os.path.join(golden_dir, "feature_extractor.npz"))
    rng = np.random.default_rng(int(g["rng_seed"]))
    enc = rng.standard_normal((25, 42, 256)).astype(np.float32)
    raw = rng.standard_normal((5, 256)).astype(np.float32)
    raw[3] = 0.0
    fe = FeatureExtractor()
    roi = fe.extract_roi_features(enc, [tuple(b) for b in g["bboxes"]], tuple(int(v) for v in g["image_shape"]))
    assert roi.dtype == np.float32
    np.testing.assert_allclose(roi, g["roi"], atol=2e-6)
    np.testing.assert_allclose(fe.normalize_features(raw), g["norm"], atol=1e-7)
    assert fe.extract_roi_features(enc, [], (800, 1333)).shape == tuple(g["empty_shape"])
    with pytest.raises(ValueError, match="Expected 3D encoder features"):
        fe.extract_roi_features(enc[0], [], (800, 1333))


def test_stem_normalisation_constants_reproduce_the_reference_arithmetic():
    """stem_pool2_kernel<U8> normalises a pixel byte as fp16(fma(v, A_c, -B_c)) (kernels_gemm.hip: STEM_NA / STEM_NB) where the
    pre-processing kernel and the oracle compute (float(v) * (1/255) - mean_c) / std_c.  256 x 3 inputs: enumerate -- the fp16 results
    must agree for every one (the GPU test test_stem_pool_with_preprocessing_inside_is_bit_identical checks the device's own fma)."""
    import re

    src = open(os.path.join(os.path.dirname(__file__), "..", "office_person_detection_vit_amd", "csrc", "kernels_gemm.hip")).read()
    na = [np.float32(x) for x in re.search(r"STEM_NA\[3\] = \{([^}]*)\}", src).group(1).replace("f", "").split(",")]
    nb = [np.float32(x) for x in re.search(r"STEM_NB\[3\] = \{([^}]*)\}", src).group(1).replace("f", "").split(",")]
    mean = np.array([0.485, 0.456, 0.406], np.float32)
    std = np.array([0.229, 0.224, 0.225], np.float32)
    k = np.float32(1.0) / np.float32(255.0)
    v = np.arange(256, dtype=np.float32)
    for c in range(3):
        ref = ((v * k - mean[c]) / std[c]).astype(np.float32).astype(np.float16)
        fma = (v.astype(np.float64) * np.float64(na[c]) - np.float64(nb[c])).astype(np.float32).astype(np.float16)   # one rounding to fp32
        np.testing.assert_array_equal(ref.view(np.uint16), fma.view(np.uint16))
