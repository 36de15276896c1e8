"""BASELINE.json configs[2] and configs[4] as WORKLOADS through the real sharding / tiling code on the one GPU a builder box has
(world size 1 RCCL group, device-direct exchange), and the memory hygiene of the forward (poisoned buffers, red zones, foreign
allocations between graph capture and replay).

configs[2]: 64 frames of 800x1333 "sharded 8 ways" = eight chunks of 8 through ``ShardedDetector``: on an 8-GPU node each rank
runs exactly one of these chunks; here rank 0 of a world-size-1 group runs all eight, through the same ``detect_shard`` code, the
same exchange buffer in HBM and the same RCCL all-gather.  Sampled frames are checked against the live oracle at 1e-3.
configs[4]: 8 4K frames -> 32 overlapping 1080p tiles through ``TiledDetector(ShardedDetector)``.
"""

import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

from office_person_detection_vit_amd import HipDetrDetector, _capi
from office_person_detection_vit_amd.frames import structured_frames
from office_person_detection_vit_amd.weights import DetrArch, ensure_weight_file, load_safetensors
from oracle import detr_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mild_path(weight_cache):
    return ensure_weight_file(weight_cache, DetrArch(), 0, 1.0, "r50")


@pytest.fixture()
def rccl_world1():
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def _sig(frames_dets):
    return [[(d.bbox, d.confidence, d.query_index) for d in f] for f in frames_dets]


def test_config2_64_frames_through_sharded_detector(mild_path, rccl_world1, parity_log):
    """64 x 800x1333 frames -> ShardedDetector (RCCL, records written by the post-process kernel into the tensor the all-gather
    reads) = 8 chunks of 8.  Frames repeat with period 16 in a shuffled order, so the workload also checks that a frame's records
    do not depend on the chunk or slot it travels in; four frames against the live oracle at the north-star tolerance."""
    from office_person_detection_vit_amd.sharding import ShardedDetector
    det = HipDetrDetector(model_path=mild_path, max_batch=8, max_size=(800, 1333), resize=False)
    det.load_model()
    try:
        base = structured_frames(16, 800, 1333, seed=6400)
        order = np.random.default_rng(64).permutation(64) % 16
        frames = [base[i] for i in order]
        got = ShardedDetector(det, device="cuda:0").detect_batch(frames)
        assert len(got) == 64
        first = {}
        for pos, i in enumerate(order):     # the same frame gives the same records wherever it sits in the 64
            first.setdefault(int(i), pos)
            assert _sig([got[pos]]) == _sig([got[first[int(i)]]])
        assert sum(len(f) for f in got) > 0
        w = O.to_torch(load_safetensors(mild_path))
        compared = 0
        for pos in (0, 21, 42, 63):
            fr = frames[pos]
            pv, pm = O.preprocess([fr])
            lg, bx, _ = O.forward(w, pv, pm)
            lg1, bx1, _ = det.forward_raw([fr], want_encoder=False)
            dbox = float(np.abs(bx1[0] - bx[0].numpy()).max())
            parity_log(f"configs[2] workload: frame {pos} of 64 (800x1333) vs live oracle", dbox, None, None, 1e-3)
            assert dbox <= 1e-3
            want = O.person_detections(O.post_process_object_detection(lg.numpy(), bx.numpy(), 0.5, [(800, 1333)])[0], 0.4)
            ref_q = {d["query_index"]: d for d in want if abs(d["confidence"] - 0.5) > 4e-3}
            got_q = {d.query_index: d for d in got[pos] if abs(d.confidence - 0.5) > 4e-3}
            assert set(ref_q) == set(got_q)
            compared += len(ref_q)
            for qi, r in ref_q.items():
                np.testing.assert_allclose(got_q[qi].bbox, r["bbox"], atol=1e-3 * 1333 * 2)
        assert compared > 0
    finally:
        det.close()


def test_config4_eight_4k_frames_as_32_tiles(mild_path, rccl_world1):
    """8 4K frames -> 32 overlapping 1080p tiles (device resize 1215x2160 -> 750x1333) through TiledDetector(ShardedDetector):
    equal to the single-process TiledDetector on the same detector, frame for frame."""
    from office_person_detection_vit_amd.sharding import ShardedDetector
    from office_person_detection_vit_amd.tiling import TiledDetector
    det = HipDetrDetector(model_path=mild_path, max_batch=4, max_size=(800, 1333), resize=True)
    det.load_model()
    try:
        frames = structured_frames(8, 2160, 3840, seed=3200)
        got = TiledDetector(ShardedDetector(det, device="cuda:0"), 2, 2, nms_threshold=0.4).detect_batch(frames)
        want = TiledDetector(det, 2, 2, nms_threshold=0.4).detect_batch(frames)
        assert len(got) == 8 and _sig(got) == _sig(want)
        assert sum(len(f) for f in got) > 0
        for f in got:
            for d in f:
                assert -3840 <= d.bbox[0] <= 3840 and -2160 <= d.bbox[1] <= 2160
    finally:
        det.close()


def _poisoned(path, poison, **kw):
    lib = _capi.load_library()
    lib.opd_test_set_alloc_poison(poison)
    try:
        det = HipDetrDetector(model_path=path, **kw)
        det.load_model()
    finally:
        lib.opd_test_set_alloc_poison(-1)
    return det


def test_forward_does_not_depend_on_memory_it_has_not_written(mild_path):
    """Handles whose every device buffer is pre-filled with 0x00 / 0xFF (fp16 and fp32 NaN patterns) and fenced by 256-KiB red
    zones of the same byte give bit-identical outputs to an unpoisoned handle — eager, captured and replayed; uniform, ragged, odd
    and device-resized batches; ROI features — and leave every red zone intact: no kernel consumes workspace it has not written,
    reads next to a buffer in a way that matters, or writes next to one."""
    lib = _capi.load_library()
    uni = structured_frames(2, 256, 320, seed=4321)
    rag = [structured_frames(1, 256, 320, seed=5)[0], structured_frames(1, 224, 288, seed=6)[0]]
    odd = structured_frames(1, 203, 333, seed=7)
    cam = structured_frames(1, 720, 1280, seed=8)

    def run(poison):
        det = _poisoned(mild_path, poison, max_batch=2, max_size=(800, 1333), resize=True)
        try:
            out = []
            det.resize = False
            for _ in range(3):
                out.append(det.forward_raw(uni))
            out.append(det.forward_raw(rag))
            out.append(det.forward_raw(odd))
            det.resize = True
            for _ in range(2):
                out.append(det.forward_raw(cam))
            dets, feats = det.detect_with_features(cam[0])
            out.append((np.asarray([d.bbox + (d.confidence,) for d in dets], np.float64).reshape(-1, 5), np.asarray(feats, np.float32)))
            if poison >= 0:
                bad = lib.opd_test_check_redzones(C.c_void_p(det.model))
                assert bad == 0, f"poison {poison:#x}: {bad} buffers with a damaged red zone; first: {_capi.last_error()}"
            return out
        finally:
            det.close()

    ref = run(-1)
    for poison in (0x00, 0xFF):
        got = run(poison)
        for k, (a, b) in enumerate(zip(got, ref)):
            for x, y in zip(a, b):
                assert not np.isnan(np.asarray(x, np.float64)).any(), f"poison {poison:#x}, step {k}: NaN in the output"
                np.testing.assert_array_equal(x, y, err_msg=f"poison {poison:#x}, step {k}")


def test_batch8_forward_on_poisoned_buffers(mild_path):
    """The same at the benchmark shape (8 x 800x1333, eager + captured + replayed + detect_batch)."""
    lib = _capi.load_library()
    frames = structured_frames(8, 800, 1333, seed=99)
    outs, sigs = {}, {}
    for poison in (-1, 0xFF):
        det = _poisoned(mild_path, poison, max_batch=8, max_size=(800, 1333), resize=False)
        try:
            outs[poison] = [det.forward_raw(frames) for _ in range(3)]
            sigs[poison] = _sig(det.detect_batch(frames))
            if poison >= 0:
                assert lib.opd_test_check_redzones(C.c_void_p(det.model)) == 0, _capi.last_error()
        finally:
            det.close()
    assert sigs[0xFF] == sigs[-1] and sum(len(f) for f in sigs[-1]) > 0
    for k, (a, b) in enumerate(zip(outs[0xFF], outs[-1])):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y, err_msg=f"step {k}")


def test_graph_replay_survives_foreign_allocations_and_handle_churn(mild_path, weight_cache):
    """A graph captured BEFORE other users of the device allocate, write and free memory is replayed AFTER it, with the library's
    own re-capture guard switched off: torch's caching allocator (1 GiB of NaNs allocated, freed back to the driver, 768 MiB
    allocated again and kept), a world-size-1 RCCL group created and destroyed, another handle created / run / destroyed and a
    handle with different weights created in the memory it returned.  Replays must stay bit-identical."""
    import torch.distributed as dist
    lib = _capi.load_library()
    sharp = ensure_weight_file(weight_cache, DetrArch(), 0, 2.0, "r50")
    probe = structured_frames(2, 256, 320, seed=4321)
    other = structured_frames(2, 256, 320, seed=1234)
    mk = lambda p: HipDetrDetector(model_path=p, max_batch=2, max_size=(800, 1333), resize=False)
    lib.opd_test_set_graph_guard(0)
    A = mk(mild_path)
    A.load_model()
    extra = []
    try:
        eager = A.forward_raw(probe)
        A.forward_raw(other)                      # capture + first launch
        ref = A.forward_raw(probe)                # replay
        for x, y in zip(ref, eager):
            np.testing.assert_array_equal(x, y)
        # (1) torch allocator churn
        x = torch.full((1 << 28,), float("nan"), device="cuda"); torch.cuda.synchronize(); del x
        torch.cuda.empty_cache()
        keep = torch.full((3 << 26,), float("nan"), device="cuda"); torch.cuda.synchronize()
        for x, y in zip(A.forward_raw(probe), ref):
            np.testing.assert_array_equal(x, y)
        # (2) RCCL communicator setup / teardown
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        t = torch.ones(1024, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
        dist.destroy_process_group()
        for x, y in zip(A.forward_raw(probe), ref):
            np.testing.assert_array_equal(x, y)
        # (3) handle churn: the round-2 scenario (tools/graph_churn_probe.py)
        f = mk(mild_path); f.load_model(); f.forward_raw(probe); f.close()
        B = mk(sharp); B.load_model(); extra.append(B); B.forward_raw(other)
        for _ in range(2):
            for x, y in zip(A.forward_raw(probe), ref):
                np.testing.assert_array_equal(x, y)
        del keep
    finally:
        lib.opd_test_set_graph_guard(1)
        A.close()
        for d in extra:
            d.close()


@pytest.mark.parametrize("size,frames_n", [((800, 1333), 2), ((256, 320), 2)])
def test_parity_on_weights_that_are_not_device_exact(weight_cache, parity_log, size, frames_n):
    """The unfavourable operating point: the same seeded recipe WITHOUT `make_device_exact`, i.e. ordinary fp32
    tensors as a real checkpoint has them — the device rounds every folded conv kernel and every linear weight to fp16 itself,
    the fp32 oracle does not round at all.  Stated bound: the north-star 1e-3 on the boxes at 800x1333 (measured 6.4e-4 .. 6.7e-4 in
    round 4; round 3: 9.2e-4 .. 1.2e-3 and a 2e-3 bound.  What changed: the decoder's linear layers run on split fp16 operands, i.e. its
    weights and activations are no longer rounded to fp16 at all (tools/drift_split.py: 4.9e-4 of drift), and the folded convolution
    kernels are rounded by error diffusion instead of to nearest), 2e-3 at 256x320 (measured 1.2e-3)."""
    H, W = size
    path = ensure_weight_file(weight_cache, DetrArch(), 0, 1.0, "r50", device_exact=False)
    det = HipDetrDetector(model_path=path, max_batch=2, max_size=(800, 1333), resize=False)
    det.load_model()
    try:
        frames = structured_frames(frames_n, H, W, seed=5150)
        lg, bx, enc = det.forward_raw(frames)
    finally:
        det.close()
    w = O.to_torch(load_safetensors(path))
    pv, pm = O.preprocess(frames)
    lg0, bx0, mem0 = O.forward(w, pv, pm)
    sm = lambda t: torch.softmax(torch.as_tensor(t), -1).numpy()
    dbox = float(np.abs(bx - bx0.numpy()).max())
    dprob = float(np.abs(sm(lg) - sm(lg0.numpy())).max())
    denc = float(np.abs(enc - mem0.numpy()).max())
    bound = 1e-3 if H >= 800 else 2e-3
    parity_log(f"r50 mild {H}x{W}, weights NOT device-exact (raw fp32 checkpoint) vs live oracle", dbox, dprob, denc, bound,
               "weight rounding included")
    assert dbox <= bound and dprob <= 2 * bound


def test_stage3_frame_split_is_bit_identical(mild_path):
    """Single-stream handles run the frames of stage 3 beyond whole rounds of its fused tail as a second chain on a forked stream,
    captured into the graph as a fork / join (csrc/opd_model.cpp::enqueue_forward).  Per-row arithmetic does not depend on the
    tiling, so the split must be invisible: batch 8 at 800x1333, eager launch, capture and replay, each against a handle created with
    OPD_TAIL3_SPLIT=0 (one launch per tail) -- equal bits."""
    frames = structured_frames(8, 800, 1333, seed=808)
    outs = {}
    for split in ("1", "0"):
        os.environ["OPD_TAIL3_SPLIT"] = split
        try:
            det = HipDetrDetector(model_path=mild_path, max_batch=8, max_size=(800, 1333), resize=False)
            det.load_model()
        finally:
            del os.environ["OPD_TAIL3_SPLIT"]
        try:
            outs[split] = [det.forward_raw(frames) for _ in range(3)]   # eager, capture + first launch, replay
        finally:
            det.close()
    for call in range(3):
        for x, y in zip(outs["1"][call], outs["0"][call]):
            np.testing.assert_array_equal(x, y)
        for x, y in zip(outs["1"][call], outs["1"][0]):
            np.testing.assert_array_equal(x, y)


def test_native_exchange_world_size_1(mild_path):
    """The C-ABI's own exchange step (csrc/opd_comm.cpp: ncclAllGather from librccl on the detector handle's stream, one host wait) over a
    communicator of ONE rank — all a one-GPU box can run: (a) opd_comm_begin / detect / exchange / wait deliver exactly the records and
    counts of opd_detr_detect on the same frames, trailing slots at count -1; (b) ShardedDetector(exchange="native") — no
    torch.distributed process group needed beyond the unique id, which rank 0 generates itself here — equals detect_batch, also for a
    shard larger than max_batch (chunks into consecutive slots) and through the device-resize path."""
    import torch.distributed as dist
    from office_person_detection_vit_amd.sharding import NativeExchange, ShardedDetector
    lib = _capi.load_library()
    det = HipDetrDetector(model_path=mild_path, max_batch=4, max_size=(256, 320), resize=False)
    det.load_model()
    try:
        frames = np.stack(structured_frames(3, 256, 320, seed=91))
        Q = det.num_queries
        hw = np.asarray([[256, 320]] * 3, np.int32)
        want = np.zeros((3, Q, 8), np.int32); want_c = np.zeros(3, np.int32)
        _capi.check(lib.opd_detr_detect(C.c_void_p(det.model), frames.ctypes.data_as(C.c_void_p), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_HOST, 3, 256, 320,
                                        0.5, hw.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.POINTER(_capi.OpdDet)), want_c.ctypes.data_as(C.POINTER(C.c_int32))),
                    "opd_detr_detect")
        uid = NativeExchange.unique_id()
        assert len(uid) == _capi.OPD_COMM_ID_BYTES
        comm = C.c_void_p()
        _capi.check(lib.opd_comm_create(uid, 0, 1, C.c_void_p(det.model), C.byref(comm)), "opd_comm_create")
        try:
            r, w = C.c_int(-1), C.c_int(-1)
            _capi.check(lib.opd_comm_info(comm, C.byref(r), C.byref(w)), "opd_comm_info")
            assert (r.value, w.value) == (0, 1)
            for _ in range(2):   # (twice: buffers are reused, the second exchange starts from a clean slate)
                _capi.check(lib.opd_comm_begin(comm, 4), "opd_comm_begin")
                _capi.check(lib.opd_comm_detect(comm, 0, frames.ctypes.data_as(C.c_void_p), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_HOST, 3, 256, 320, 0.5,
                                                hw.ctypes.data_as(C.c_void_p)), "opd_comm_detect")
                _capi.check(lib.opd_comm_exchange(comm), "opd_comm_exchange")
                # a lane holds TWO exchanges (round 5): the next one may be begun and issued while this one travels; a third is refused
                _capi.check(lib.opd_comm_begin(comm, 4), "opd_comm_begin")
                assert lib.opd_comm_begin(comm, 4) == _capi.OPD_ESTATE          # the one just begun has not been issued
                _capi.check(lib.opd_comm_detect(comm, 1, frames[:2].ctypes.data_as(C.c_void_p), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_HOST, 2, 256, 320, 0.5,
                                                hw.ctypes.data_as(C.c_void_p)), "opd_comm_detect")
                _capi.check(lib.opd_comm_exchange(comm), "opd_comm_exchange")
                assert lib.opd_comm_begin(comm, 4) == _capi.OPD_ESTATE          # two outstanding
                got = np.zeros((1, 4, Q, 8), np.int32); got_c = np.zeros((1, 4), np.int32)
                _capi.check(lib.opd_comm_wait(comm, got.ctypes.data_as(C.POINTER(_capi.OpdDet)), got_c.ctypes.data_as(C.POINTER(C.c_int32))), "opd_comm_wait")
                assert got_c[0].tolist() == want_c.tolist() + [-1]              # the OLDER exchange comes first
                for f in range(3):
                    np.testing.assert_array_equal(got[0, f, :want_c[f]], want[f, :want_c[f]])
                _capi.check(lib.opd_comm_wait(comm, got.ctypes.data_as(C.POINTER(_capi.OpdDet)), got_c.ctypes.data_as(C.POINTER(C.c_int32))), "opd_comm_wait")
                assert got_c[0].tolist() == [-1] + want_c[:2].tolist() + [-1]   # frames 0, 1 in slots 1, 2
                for f in range(2):
                    np.testing.assert_array_equal(got[0, 1 + f, :want_c[f]], want[f, :want_c[f]])
            assert lib.opd_comm_wait(comm, got.ctypes.data_as(C.POINTER(_capi.OpdDet)), got_c.ctypes.data_as(C.POINTER(C.c_int32))) != 0   # nothing outstanding
            assert lib.opd_comm_detect(comm, 2, frames.ctypes.data_as(C.c_void_p), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_HOST, 3, 256, 320, 0.5,
                                       hw.ctypes.data_as(C.c_void_p)) != 0                                                                  # slots 2..4 of 4
        finally:
            lib.opd_comm_destroy(comm)
        # (b) the Python layer: world size 1 needs no process group beyond what dist.get_rank() answers
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        dist.init_process_group("gloo", rank=0, world_size=1)   # (gloo: torch.distributed carries the unique id only, never the records)
        try:
            sharded = ShardedDetector(det, exchange="native")
            many = structured_frames(9, 256, 320, seed=92)   # 9 frames through a max_batch = 4 handle: three chunks
            assert _sig(sharded.detect_batch(many)) == _sig(det.detect_batch(many))
            assert sum(len(f) for f in sharded.detect_batch(many)) > 0
            sharded.close()
        finally:
            dist.destroy_process_group()
    finally:
        det.close()


def test_communicator_lanes_share_one_communicator_and_survive_their_handles(mild_path):
    """Round 5: ONE RCCL communicator per rank with a LANE per detector handle (opd_comm_attach): own buffers and events, all all-gathers on
    the communicator's own stream in submission order.  (a) three lanes (handle + two clones), exchanges submitted on all three before any
    is waited for, waited in another order: every lane delivers exactly what opd_detr_detect gives for ITS frames; (b) (a lane's own depth of two
    exchanges: test_native_exchange_world_size_1); (c) lifetime: destroying a handle detaches its lanes -- they
    answer OPD_ESTATE from then on and opd_comm_destroy still frees them; lanes may be destroyed in any order, the parent first."""
    from office_person_detection_vit_amd.sharding import NativeExchange
    lib = _capi.load_library()
    det = HipDetrDetector(model_path=mild_path, max_batch=2, max_size=(256, 320), resize=False)
    det.load_model()
    Q = det.num_queries
    hw = np.asarray([[256, 320]] * 2, np.int32)
    DetP, I32P = C.POINTER(_capi.OpdDet), C.POINTER(C.c_int32)
    handles = [C.c_void_p(det.model)]
    for _ in range(2):
        hx = C.c_void_p()
        _capi.check(lib.opd_detr_clone(handles[0], C.byref(hx)), "opd_detr_clone")
        handles.append(hx)
    batches = [np.stack(structured_frames(2, 256, 320, seed=300 + k)) for k in range(3)]
    want = []
    for k in range(3):
        r = np.zeros((2, Q, 8), np.int32); c = np.zeros(2, np.int32)
        _capi.check(lib.opd_detr_detect(handles[k], batches[k].ctypes.data_as(C.c_void_p), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_HOST, 2, 256, 320, 0.5,
                                        hw.ctypes.data_as(C.c_void_p), r.ctypes.data_as(DetP), c.ctypes.data_as(I32P)), "opd_detr_detect")
        want.append((r, c))
    assert lib.opd_comm_available() == 0
    lanes = [C.c_void_p()]
    _capi.check(lib.opd_comm_create(NativeExchange.unique_id(), 0, 1, handles[0], C.byref(lanes[0])), "opd_comm_create")
    for k in (1, 2):
        lx = C.c_void_p()
        _capi.check(lib.opd_comm_attach(lanes[0], handles[k], C.byref(lx)), "opd_comm_attach")
        lanes.append(lx)
    try:
        for rnd in range(3):
            for k in range(3):
                _capi.check(lib.opd_comm_begin(lanes[k], 2), "opd_comm_begin")
                _capi.check(lib.opd_comm_detect(lanes[k], 0, batches[k].ctypes.data_as(C.c_void_p), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_HOST, 2, 256, 320, 0.5,
                                                hw.ctypes.data_as(C.c_void_p)), "opd_comm_detect")
                _capi.check(lib.opd_comm_exchange(lanes[k]), "opd_comm_exchange")
            for k in ((2, 0, 1), (0, 1, 2), (1, 2, 0))[rnd]:
                got = np.zeros((1, 2, Q, 8), np.int32); got_c = np.zeros((1, 2), np.int32)
                _capi.check(lib.opd_comm_wait(lanes[k], got.ctypes.data_as(DetP), got_c.ctypes.data_as(I32P)), "opd_comm_wait")
                assert got_c[0].tolist() == want[k][1].tolist()
                for f in range(2):
                    np.testing.assert_array_equal(got[0, f, :want[k][1][f]], want[k][0][f, :want[k][1][f]])
        # (c) handle 2 goes first: its lane is detached, not dangling
        lib.opd_detr_destroy(handles[2])
        assert lib.opd_comm_begin(lanes[2], 2) == _capi.OPD_ESTATE and b"destroyed" in lib.opd_last_error()
        lib.opd_comm_destroy(lanes[2])
        lib.opd_comm_destroy(lanes[0])   # the parent lane before its sibling: the communicator lives on in lane 1
        _capi.check(lib.opd_comm_begin(lanes[1], 2), "opd_comm_begin")
        _capi.check(lib.opd_comm_detect(lanes[1], 0, batches[1].ctypes.data_as(C.c_void_p), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_HOST, 2, 256, 320, 0.5,
                                        hw.ctypes.data_as(C.c_void_p)), "opd_comm_detect")
        _capi.check(lib.opd_comm_exchange(lanes[1]), "opd_comm_exchange")
        got = np.zeros((1, 2, Q, 8), np.int32); got_c = np.zeros((1, 2), np.int32)
        _capi.check(lib.opd_comm_wait(lanes[1], got.ctypes.data_as(DetP), got_c.ctypes.data_as(I32P)), "opd_comm_wait")
        assert got_c[0].tolist() == want[1][1].tolist()
        lib.opd_comm_destroy(lanes[1])
        lib.opd_detr_destroy(handles[1])
    finally:
        det.close()


def test_bench_pipelined_loop_through_a_one_rank_communicator():
    """with the native exchange bench.py's pipelined loop submitted step i on a communicator whose step i - NS was still
    outstanding -- OPD_ESTATE at step NS on every rank, and nothing ever ran that path (communicators existed only for world > 1).  The test
    switch OPD_BENCH_FORCE_COMM=1 drives the same loop over a ONE-rank communicator with three lanes: more than 2 NS steps, warm-up included,
    then the sustained leg; the line must say which exchange ran and report detections."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "OPD_TEST_HOOKS")}
    env.update({"OPD_BENCH_FORCE_COMM": "1", "OPD_BENCH_SUSTAINED": "0"})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for comms in ("shared", "per-handle"):
        env["OPD_BENCH_COMMS"] = comms
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "11", "--warmup", "4", "--batch", "2", "--height", "256", "--width", "320",
                            "--no-cpu-baseline", "--serial-steps", "0"], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        out = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
        assert out["exchange"].startswith("native") and ("ONE communicator" in out["exchange"]) == (comms == "shared"), out["exchange"]
        assert out["steps"] == 11 and out["value"] > 0 and out["detections_last_step"] is not None


def test_measurement_abi_modes_are_consistent(mild_path):
    """opd_detr_set_profiling / stage_times / kernel_times / kernel_table had no test.  (a) outputs are bit-identical under
    modes 0, 1 and 2; (b) mode 2: every stage mark > 0 and their sum <= the wall time of the call; (c) mode 1: flops4 sums to the
    architecture's algorithmic FLOPs within 1 %, the kernel table accounts for every launch and every millisecond of kernel_times and its
    rows carry real kernel names; (d) switching modes back and forth on one handle re-captures cleanly (the `invalid resource handle` seen in
    round 4 at opd_model.cpp:660 was exactly this path)."""
    lib = _capi.load_library()
    H, W, B = 256, 320, 2
    cfg = _capi.OpdConfig(struct_size=C.sizeof(_capi.OpdConfig), max_batch=B, max_height=H, max_width=W, flags=0)
    h = C.c_void_p()
    _capi.check(lib.opd_detr_create(C.byref(cfg), mild_path.encode(), 0, C.byref(h)), "opd_detr_create")
    try:
        frames = np.stack(structured_frames(B, H, W, seed=77))
        logits = np.zeros((B, 100, 92), np.float32); boxes = np.zeros((B, 100, 4), np.float32)
        F32P = C.POINTER(C.c_float)

        def fwd():
            t0 = time.perf_counter()
            _capi.check(lib.opd_detr_forward(h, frames.ctypes.data_as(C.c_void_p), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_HOST, B, H, W,
                                             logits.ctypes.data_as(F32P), boxes.ctypes.data_as(F32P), None), "opd_detr_forward")
            return 1e3 * (time.perf_counter() - t0), logits.copy(), boxes.copy()

        hw = np.asarray([[H, W]] * B, np.int32)
        recs = np.zeros((B, 100, 8), np.int32); cnts = np.zeros(B, np.int32)

        def det():   # (the stage / kernel times are those of the last DETECT call: forward + post-process)
            t0 = time.perf_counter()
            _capi.check(lib.opd_detr_detect(h, frames.ctypes.data_as(C.c_void_p), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_HOST, B, H, W, 0.5,
                                            hw.ctypes.data_as(C.c_void_p), recs.ctypes.data_as(C.POINTER(_capi.OpdDet)), cnts.ctypes.data_as(C.POINTER(C.c_int32))),
                        "opd_detr_detect")
            return 1e3 * (time.perf_counter() - t0), recs.copy(), cnts.copy()

        ref = ref_det = None
        for mode in (0, 2, 1, 0, 1, 2, 2, 0):   # back and forth; each mode runs eager / capture / replay
            _capi.check(lib.opd_detr_set_profiling(h, mode), "opd_detr_set_profiling")
            for it in range(3):
                _, lg, bx = fwd()
                if ref is None:
                    ref = (lg, bx)
                np.testing.assert_array_equal(lg, ref[0])
                np.testing.assert_array_equal(bx, ref[1])
            for it in range(3):
                wall_ms, rr, cc = det()
                if ref_det is None:
                    ref_det = (rr, cc)
                    assert cc.sum() > 0
                np.testing.assert_array_equal(cc, ref_det[1])
                for f in range(B):
                    np.testing.assert_array_equal(rr[f, :cc[f]], ref_det[0][f, :cc[f]])
                s8 = (C.c_float * 8)()
                _capi.check(lib.opd_detr_stage_times(h, s8), "opd_detr_stage_times")
                st = np.asarray(list(s8))
                if mode == 2 and it == 2:      # a replay with the marks inside the graph
                    assert (st[:7] > 0).all(), st
                    assert st.sum() <= wall_ms, (st.sum(), wall_ms)
                if mode == 1:
                    ms4, l4, f4 = (C.c_float * 4)(), (C.c_int32 * 4)(), (C.c_double * 4)()
                    _capi.check(lib.opd_detr_kernel_times(h, ms4, l4, f4), "opd_detr_kernel_times")
                    # algorithmic FLOPs of the architecture at this size: the oracle's layer list is the independent count
                    from tools_flops import detr_r50_flops
                    want = detr_r50_flops(H, W) * B
                    assert abs(sum(f4) - want) <= 0.01 * want, (sum(f4), want)
                    tab = (_capi.OpdKernelStat * 64)(); n = C.c_int(0)
                    _capi.check(lib.opd_detr_kernel_table(h, tab, 64, C.byref(n)), "opd_detr_kernel_table")
                    rows = [(tab[i].name.decode(), tab[i].launches, tab[i].ms, tab[i].flops) for i in range(n.value)]
                    assert sum(r[1] for r in rows) == sum(l4) and abs(sum(r[2] for r in rows) - sum(ms4)) <= 1e-3 * sum(ms4) + 1e-4
                    assert abs(sum(r[3] for r in rows) - sum(f4)) <= 1e-6 * sum(f4)
                    names = " ".join(r[0] for r in rows)
                    assert "conv_gemm_dma_kernel<" in names and "attention_kernel<" in names and "stem_pool2_kernel<true>" in names, names
                    assert all(rows[i][2] >= rows[i + 1][2] for i in range(len(rows) - 1))   # longest first
    finally:
        lib.opd_detr_destroy(h)


def test_device_mem_kinds_refuse_pointers_that_are_not_device_memory(mild_path):
    """A host pointer handed over under OPD_MEM_DEVICE / OPD_MEM_HOST_PIXELS_DEVICE_OUT would make a kernel fault the GPU (for every process
    on it); the C-ABI returns OPD_EINVAL instead, and the handle keeps working.  (Found by doing it: tools/host_b1_probe.py, round 5.)"""
    lib = _capi.load_library()
    H, W, B = 256, 320, 1
    cfg = _capi.OpdConfig(struct_size=C.sizeof(_capi.OpdConfig), max_batch=B, max_height=H, max_width=W, flags=0)
    h = C.c_void_p()
    _capi.check(lib.opd_detr_create(C.byref(cfg), mild_path.encode(), 0, C.byref(h)), "opd_detr_create")
    try:
        frames = np.stack(structured_frames(B, H, W, seed=5))
        d_frames = torch.from_numpy(frames).cuda()
        hw = np.asarray([[H, W]] * B, np.int32)
        recs = np.zeros((B, 100, 8), np.int32); cnts = np.zeros(B, np.int32)
        d_out = torch.zeros((B * 100 * 8 + B,), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        DetP, I32P = C.POINTER(_capi.OpdDet), C.POINTER(C.c_int32)
        h_recs, h_cnts = recs.ctypes.data_as(DetP), cnts.ctypes.data_as(I32P)
        d_recs, d_cnts = C.cast(C.c_void_p(d_out.data_ptr()), DetP), C.cast(C.c_void_p(d_out[B * 800:].data_ptr()), I32P)

        def detect(pix, kind, r, c):
            return lib.opd_detr_detect(h, pix, _capi.OPD_PIXELS_U8_BGR_HWC, kind, B, H, W, 0.5, hw.ctypes.data_as(C.c_void_p), r, c)

        host_pix, dev_pix = frames.ctypes.data_as(C.c_void_p), C.c_void_p(d_frames.data_ptr())
        assert detect(host_pix, _capi.OPD_MEM_DEVICE, d_recs, d_cnts) == _capi.OPD_EINVAL          # host pixels as device pixels
        assert b"not device-accessible" in lib.opd_last_error()
        assert detect(dev_pix, _capi.OPD_MEM_DEVICE, h_recs, h_cnts) == _capi.OPD_EINVAL          # host outputs as device outputs
        assert detect(host_pix, _capi.OPD_MEM_HOST_PIXELS_DEVICE_OUT, h_recs, h_cnts) == _capi.OPD_EINVAL
        t = C.c_int()
        assert lib.opd_detr_detect_async(h, dev_pix, _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_DEVICE, B, H, W, 0.5, hw.ctypes.data_as(C.c_void_p),
                                         h_recs, h_cnts, C.byref(t)) == _capi.OPD_EINVAL
        # the right pointers: both memory kinds give the same records
        assert detect(host_pix, _capi.OPD_MEM_HOST, h_recs, h_cnts) == 0
        assert detect(dev_pix, _capi.OPD_MEM_DEVICE, d_recs, d_cnts) == 0
        got = d_out.cpu().numpy()
        n = max(int(cnts[0]), 0)
        assert np.array_equal(got[:B * 800].reshape(B, 100, 8)[0, :n], recs[0, :n]) and got[B * 800] == cnts[0]
    finally:
        lib.opd_detr_destroy(h)


def test_stage1_residual_rebuild_is_invisible_end_to_end(mild_path):
    """OPD_TAIL_RC / OPD_Y_STRIDE2 (round 5: stage 1's first tail stores a1 instead of its output, the second rebuilds it, the last stores its
    output only where the next stage reads it): logits, boxes and the encoder map of a batch are bit-identical with the switches on and off,
    eager, captured and replayed, also for a ragged last tile (203 x 333) and through a clone (the switches travel with opd_detr_clone)."""
    for (H, W, B) in ((256, 320, 3), (203, 333, 2)):
        frames = structured_frames(B, H, W, seed=515)
        outs = {}
        for flag in ("1", "0"):
            os.environ["OPD_TAIL_RC"] = flag; os.environ["OPD_Y_STRIDE2"] = flag
            try:
                det = HipDetrDetector(model_path=mild_path, max_batch=B, max_size=(H, W), resize=False, streams=2)
                det.load_model()
            finally:
                del os.environ["OPD_TAIL_RC"], os.environ["OPD_Y_STRIDE2"]
            try:
                outs[flag] = [det.forward_raw(frames) for _ in range(3)]
            finally:
                det.close()
        for call in range(3):
            for x, y in zip(outs["1"][call], outs["0"][call]):
                np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_tail_workgroup_shape_is_invisible_end_to_end(mild_path, monkeypatch, dtype):
    """OPD_TAIL_NW = 8 (the stage 1-2 fused tails as eight-wave workgroups on 256-pixel tiles: one set of weight tiles staged per 256 pixels)
    against the four-wave form: a wave's arithmetic is the same, so logits, boxes and the encoder map are bit-identical -- with the
    residual rebuild on and off (all tail variants), ragged last tiles (203 x 333: 4 250 / 1 092 pixels at stage 1 / 2 are not multiples of 256),
    eager, captured and replayed."""
    for (H, W, B) in ((256, 320, 3), (203, 333, 2)):
        frames = structured_frames(B, H, W, seed=616)
        for rc in ("1", "0"):
            monkeypatch.setenv("OPD_TAIL_RC", rc)
            outs = {}
            for nw in ("4", "8"):
                monkeypatch.setenv("OPD_TAIL_NW", nw)
                det = HipDetrDetector(model_path=mild_path, max_batch=B, max_size=(H, W), resize=False, dtype=dtype)
                det.load_model()
                try:
                    outs[nw] = [det.forward_raw(frames) for _ in range(3)]
                finally:
                    det.close()
            for call in range(3):
                for x, y in zip(outs["4"][call], outs["8"][call]):
                    np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("flag_bf16", [0, 1])
def test_multi_stream_eight_wave_3x3_is_invisible_end_to_end(mild_path, monkeypatch, flag_bf16):
    """OPD_FLAG_MULTI_STREAM handles run stage 4's 3x3 convolutions through the eight-wave kernel (kernels_w8.hip; OPD_W8 = -1 = by the flags):
    a speed choice only -- logits, boxes and the encoder map are bit-identical with OPD_W8=0, for fp16 and for the bf16 instantiation."""
    lib = _capi.load_library()
    H, W, B = 384, 512, 2
    frames = np.stack(structured_frames(B, H, W, seed=19))
    monkeypatch.setenv("OPD_SMALL_SPLITK", "0")   # (a handle this small would split the reduction of its deep convolutions instead: another plan)
    outs = []
    for w8 in ("0", None):
        if w8 is None:
            monkeypatch.delenv("OPD_W8", raising=False)
        else:
            monkeypatch.setenv("OPD_W8", w8)
        cfg = _capi.OpdConfig(struct_size=C.sizeof(_capi.OpdConfig), max_batch=B, max_height=H, max_width=W,
                              flags=_capi.OPD_FLAG_MULTI_STREAM | (_capi.OPD_FLAG_BF16 if flag_bf16 else 0))
        h = C.c_void_p()
        _capi.check(lib.opd_detr_create(C.byref(cfg), mild_path.encode(), 0, C.byref(h)), "opd_detr_create")
        try:
            logits = np.zeros((B, 100, 92), np.float32); boxes = np.zeros((B, 100, 4), np.float32); enc = np.zeros((B, 12 * 16, 256), np.float32)
            F32P = C.POINTER(C.c_float)
            _capi.check(lib.opd_detr_set_profiling(h, 1), "opd_detr_set_profiling")      # (the kernel table tells which kernels ran)
            hw = np.asarray([[H, W]] * B, np.int32)
            recs = np.zeros((B, 100, 8), np.int32); cnts = np.zeros(B, np.int32)
            _capi.check(lib.opd_detr_detect(h, frames.ctypes.data_as(C.c_void_p), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_HOST, B, H, W, 0.5,
                                            hw.ctypes.data_as(C.c_void_p), recs.ctypes.data_as(C.POINTER(_capi.OpdDet)), cnts.ctypes.data_as(C.POINTER(C.c_int32))),
                        "opd_detr_detect")
            ktab = (_capi.OpdKernelStat * 64)(); kc = C.c_int(0)
            _capi.check(lib.opd_detr_kernel_table(h, ktab, 64, C.byref(kc)), "opd_detr_kernel_table")
            names = [ktab[i].name.decode() for i in range(kc.value)]
            _capi.check(lib.opd_detr_set_profiling(h, 0), "opd_detr_set_profiling")
            for _ in range(3):
                _capi.check(lib.opd_detr_forward(h, frames.ctypes.data_as(C.c_void_p), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_HOST, B, H, W,
                                                 logits.ctypes.data_as(F32P), boxes.ctypes.data_as(F32P), enc.ctypes.data_as(F32P)), "opd_detr_forward")
            outs.append((logits.copy(), boxes.copy(), enc.copy(), names))
        finally:
            lib.opd_detr_destroy(h)
    assert not any("conv_w8" in n for n in outs[0][3]) and any("conv_w8" in n for n in outs[1][3]), outs[1][3]
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_small_handle_split_k_plan(mild_path, parity_log):
    """Round 5: a max_batch = 1 handle splits the reduction of its deep convolutions over workgroups (csrc/opd_model.cpp::conv_splits: stage 3 / 4's
    3x3 and stage 4's 2048 -> 512 reduce become split-K launches + reduce_act16_kernel) because unsplit they occupy a quarter of the CUs for
    72 k-steps.  (a) the plan engages at 800x1333 batch 1 (the kernel table of a profiled forward shows the reduction kernel) and not at batch 8;
    (b) its boxes agree with the unsplit plan (OPD_SMALL_SPLITK=0) far inside the north-star tolerance and with the live oracle at it.
    The same handle runs the encoder side's deep linears as split-K GEMMs + reduce / LayerNorm instead of the row-owner launches
    (OPD_SMALL_ENC: 22 workgroups would each stream 2.2 MB of weights): the kernel table shows no enc_ffn_kernel then."""
    lib = _capi.load_library()
    H, W = 800, 1333
    frames = structured_frames(1, H, W, seed=4242)
    outs = {}
    for flag in ("1", "0"):
        os.environ["OPD_SMALL_SPLITK"] = flag; os.environ["OPD_SMALL_ENC"] = flag
        try:
            det = HipDetrDetector(model_path=mild_path, max_batch=1, max_size=(H, W), resize=False)
            det.load_model()
        finally:
            del os.environ["OPD_SMALL_SPLITK"], os.environ["OPD_SMALL_ENC"]
        try:
            outs[flag] = det.forward_raw(frames)
            det.set_profiling(1)
            det.detect_batch(frames)
            tab = (_capi.OpdKernelStat * 64)(); n = C.c_int(0)
            _capi.check(lib.opd_detr_kernel_table(C.c_void_p(det.model), tab, 64, C.byref(n)), "opd_detr_kernel_table")
            names = [tab[i].name.decode() for i in range(n.value)]
            assert any("reduce_act16_kernel" in k for k in names) == (flag == "1"), names
            assert any("enc_ffn_kernel" in k for k in names) == (flag == "0"), names
        finally:
            det.close()
    dbox = float(np.abs(outs["1"][1] - outs["0"][1]).max())
    w = O.to_torch(load_safetensors(mild_path))
    pv, pm = O.preprocess(frames)
    lg0, bx0, _ = O.forward(w, pv, pm)
    d_oracle = float(np.abs(outs["1"][1] - bx0.numpy()).max())
    parity_log("r50 mild 800x1333, max_batch = 1 handle (split-K plan) vs live oracle", d_oracle, None, None, 1e-3, f"split vs unsplit plan {dbox:.1e}")
    assert dbox <= 4e-4 and d_oracle <= 1e-3


def test_multi_stream_plan_matches_live_oracle_at_batch8(mild_path, parity_log):
    """The BENCHMARKED kernel plan: a `streams=3` detector creates its handles with OPD_FLAG_MULTI_STREAM, which takes
    stage 3 through the fused eight-wave tail for every call — other kernels than a single-stream handle runs at this shape.  The
    batch-8 rows of test_batch8_full_size_frames_match_live_oracle again on that plan: three frames of one batch-8 forward of handle 0
    and of handle 2 against the oracle run live, at the north-star tolerance; the two handles agree bit for bit."""
    det = HipDetrDetector(model_path=mild_path, max_batch=8, max_size=(800, 1333), resize=False, streams=3)
    det.load_model()
    try:
        frames = structured_frames(8, 800, 1333, seed=8800)
        lg8, bx8, enc8 = det.forward_raw(frames)
        w = O.to_torch(load_safetensors(mild_path))
        sm = lambda t: torch.softmax(torch.as_tensor(t), -1).numpy()
        for i in (0, 3, 7):
            pv, pm = O.preprocess([frames[i]])
            lg, bx, mem = O.forward(w, pv, pm)
            dbox = float(np.abs(bx8[i] - bx[0].numpy()).max())
            dprob = float(np.abs(sm(lg8[i]) - sm(lg[0].numpy())).max())
            parity_log(f"r50 mild 800x1333 batch 8, MULTI-STREAM plan (streams=3), frame {i} vs live oracle", dbox, dprob,
                       float(np.abs(enc8[i] - mem[0].numpy()).max()), 1e-3)
            assert dbox <= 1e-3 and dprob <= 2e-3
        # every handle of the detector runs the same plan on the same weights
        B, H, W = 8, 800, 1333
        batch = np.stack(frames)
        outs = []
        for hx in (det._handles[0], det._handles[2]):
            lgh = np.empty((B, 100, 92), np.float32); bxh = np.empty((B, 100, 4), np.float32)
            _capi.check(_capi.load_library().opd_detr_forward(C.c_void_p(hx), batch.ctypes.data_as(C.c_void_p), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_HOST,
                                                              B, H, W, lgh.ctypes.data_as(C.c_void_p), bxh.ctypes.data_as(C.c_void_p), None), "opd_detr_forward")
            outs.append((lgh, bxh))
        np.testing.assert_array_equal(outs[0][0], outs[1][0])
        np.testing.assert_array_equal(outs[0][1], outs[1][1])
        np.testing.assert_array_equal(outs[0][1], bx8)
    finally:
        det.close()


def test_r50_tile_1080x1920_matches_live_oracle(mild_path, parity_log):
    """a 4K frame's tile at MODEL input size 1080x1920 (r50: 34 x 60 = 2040 tokens, 405 GFLOP) — benchmarked since
    round 1 (profiles/*_bench_r50_tile1080p_b4.json) but until now only compared with itself.  One frame of a batch-2 forward against
    the oracle run live at that size, at the north-star tolerance, plus batch invariance."""
    det = HipDetrDetector(model_path=mild_path, max_batch=2, max_size=(1080, 1920), resize=False)
    det.load_model()
    try:
        frames = structured_frames(2, 1080, 1920, seed=4160)
        lg2, bx2, enc2 = det.forward_raw(frames)
        assert enc2.shape == (2, 34 * 60, 256)
        w = O.to_torch(load_safetensors(mild_path))
        pv, pm = O.preprocess([frames[1]])
        lg, bx, mem = O.forward(w, pv, pm)
        sm = lambda t: torch.softmax(torch.as_tensor(t), -1).numpy()
        dbox = float(np.abs(bx2[1] - bx[0].numpy()).max())
        dprob = float(np.abs(sm(lg2[1]) - sm(lg[0].numpy())).max())
        parity_log("r50 mild 1080x1920 (BASELINE configs[4] tile at model size), frame 1 of 2 vs live oracle", dbox, dprob,
                   float(np.abs(enc2[1] - mem[0].numpy()).max()), 1e-3)
        assert dbox <= 1e-3 and dprob <= 2e-3
        lg1, bx1, _ = det.forward_raw([frames[1]], want_encoder=False)
        assert float(np.abs(bx1[0] - bx2[1]).max()) <= 1e-6   # a frame does not depend on the batch it travels in
    finally:
        det.close()


@pytest.mark.parametrize("env", [{"OPD_ENC_FRONT": "0"}, {"OPD_FUSED_ENC_FFN": "0"}, {"OPD_ENC_TAIL": "1"}, {"OPD_FUSED_DEC": "0"}])
def test_alternative_launch_plans_agree_with_the_default(mild_path, parity_log, monkeypatch, env):
    """Every encoder / decoder launch plan the library can take for the same weights — the attention output projection as its own launch,
    the FFN block as two launches, the next layer's q / k / v (and the decoder's memory k / v) inside the FFN launch, the decoder as the
    round-3 chain — computes the same function: each against the live oracle at the north-star tolerance (800x1333, ragged second frame:
    padding mask and per-frame position tables go through every plan), and against the default plan within the fp16 noise of two equally
    precise summation orders."""
    frames = structured_frames(2, 800, 1333, seed=5151)
    frames[1] = np.ascontiguousarray(frames[1][:640, :1100])

    def run():
        det = HipDetrDetector(model_path=mild_path, max_batch=2, max_size=(800, 1333), resize=False)
        det.load_model()
        try:
            return det.forward_raw(frames)
        finally:
            det.close()

    lg0, bx0, enc0 = run()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    lg1, bx1, enc1 = run()
    w = O.to_torch(load_safetensors(mild_path))
    pv, pm = O.preprocess(frames)
    lg, bx, mem = O.forward(w, pv, pm)
    sm = lambda t: torch.softmax(torch.as_tensor(t), -1).numpy()
    dbox = float(np.abs(bx1 - bx.numpy()).max())
    parity_log(f"r50 mild 800x1333 + ragged 640x1100, plan {env} vs live oracle", dbox, float(np.abs(sm(lg1) - sm(lg.numpy())).max()), None, 1e-3,
               f"default plan on the same frames {float(np.abs(bx0 - bx.numpy()).max()):.1e}; plans apart {float(np.abs(bx1 - bx0).max()):.1e}")
    assert dbox <= 1e-3
    assert float(np.abs(bx1 - bx0).max()) <= 8e-4
