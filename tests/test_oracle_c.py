"""The plain-C restatement (``oracle/detr_ref.c``) against the same HF golden vectors as the torch-CPU oracle: two
independent restatements of the path have to agree with the captured outputs of the real module."""

import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from office_person_detection_vit_amd.frames import structured_frames
from office_person_detection_vit_amd.weights import DetrArch, synth_weights
from oracle import detr_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "libdetr_ref.so")


class RefTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float))]


@pytest.fixture(scope="module")
def clib():
    if not os.path.exists(LIB):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    lib = C.CDLL(LIB)
    lib.detr_ref_forward.restype = C.c_int
    lib.detr_ref_postprocess.restype = C.c_int
    return lib


def test_c_oracle_matches_hf_golden(clib, golden_dir):
    g = np.load(os.path.join(golden_dir, "r50_mild_odd_203x333.npz"))
    arch = DetrArch()
    w = synth_weights(arch, int(g["seed"]), float(g["attention_gain"]))
    keep = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in w.items()}
    tensors = (RefTensor * len(keep))()
    for i, (k, v) in enumerate(keep.items()):
        tensors[i].name = k.encode()
        tensors[i].data = v.ctypes.data_as(C.POINTER(C.c_float))
    h, wd = (int(v) for v in g["sizes"][0])
    frame = structured_frames(1, h, wd, seed=int(g["frame_seed"]))[0]
    pv, _ = O.preprocess([frame])
    pix = np.ascontiguousarray(pv[0].numpy())
    fh = fw = None
    fh, fw = h, wd
    for _ in range(5):
        fh, fw = (fh - 1) // 2 + 1, (fw - 1) // 2 + 1
    Q, ncls = arch.num_queries, arch.num_labels + 1
    logits = np.empty((Q, ncls), np.float32)
    boxes = np.empty((Q, 4), np.float32)
    enc = np.empty((fh * fw, 256), np.float32)
    depths = (C.c_int * 4)(*arch.depths)
    rc = clib.detr_ref_forward(tensors, len(keep), pix.ctypes.data_as(C.POINTER(C.c_float)), h, wd, depths, arch.encoder_layers,
                               arch.decoder_layers, Q, ncls, arch.ffn_dim, logits.ctypes.data_as(C.POINTER(C.c_float)),
                               boxes.ctypes.data_as(C.POINTER(C.c_float)), enc.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == 0
    # fp32 summation order differs from torch's blocked kernels: 1e-4-level agreement is what two correct fp32
    # implementations of this (sensitive) network give
    np.testing.assert_allclose(boxes, g["pred_boxes"][0], atol=3e-4)
    np.testing.assert_allclose(enc, g["encoder_last_hidden_state"][0], atol=5e-3)
    np.testing.assert_allclose(logits, g["logits"][0], atol=5e-3)
    # post-process against HF's own post_process_object_detection output
    out = np.empty((Q, 7), np.float32)
    n = clib.detr_ref_postprocess(g["logits"][0].ctypes.data_as(C.POINTER(C.c_float)),
                                  np.ascontiguousarray(g["pred_boxes"][0]).ctypes.data_as(C.POINTER(C.c_float)), Q, ncls,
                                  C.c_float(0.5), h, wd, out.ctypes.data_as(C.POINTER(C.c_float)))
    assert n == len(g["post0_scores"])
    np.testing.assert_allclose(out[:n, 4], g["post0_scores"], atol=1e-6)
    np.testing.assert_array_equal(out[:n, 5].astype(np.int64), g["post0_labels"])
    np.testing.assert_allclose(out[:n, :4], g["post0_boxes"], atol=1e-3)
