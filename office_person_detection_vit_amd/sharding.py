"""Frame-sharded data parallelism over the GPUs of one node: one process per GPU, RCCL all-gather of detections.

The reference has no distributed mode (single process, per-frame loop,
``src/pipeline/phases/detection.py:91-94``); DETR inference has no cross-frame state, so the path shards by frame
(SURVEY.md §8e): rank r of R detects frames ``[r*ceil(n/R), ...)`` on its own GPU, and ONE exchange step — an
all-gather of fixed-size detection records (32 bytes x ``num_queries`` per frame, 3.2 KB/frame) plus per-frame counts —
returns everything to every rank (rank 0 is the orchestrator that feeds tracking/transform).  The payload is tiny, so the
collective is latency-bound; xGMI link bandwidth is irrelevant at this size.

Two exchanges behind one interface (``ShardedDetector(exchange=...)``):

* ``"native"`` — the C-ABI's own step (``opd_comm_*``, ``csrc/opd_comm.cpp``): ``ncclAllGather`` from librccl enqueued on the detector
  handle's stream right behind the post-process kernel, one host wait per exchange.  ``torch.distributed`` only carries the 128-byte
  unique id from rank 0 to the others at set-up (any launcher could: the library does no rendezvous).
* ``"torch"`` — ``dist.all_gather_into_tensor`` on a torch tensor: backend ``nccl`` (= RCCL) with device tensors, or ``gloo`` with CPU
  tensors in the CPU tests of the exchange / assembly logic (no GPU there, hence no native path).
"""

from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _capi
from .data_models import Detection

# numpy view of `opd_det` (include/opd_detr.h)
DET_DTYPE = np.dtype([("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4"), ("score", "<f4"),
                      ("label", "<i4"), ("query_index", "<i4"), ("frame", "<i4")])
assert DET_DTYPE.itemsize == C.sizeof(_capi.OpdDet) == 32


def shard_bounds(n_frames: int, rank: int, world: int) -> Tuple[int, int, int]:
    """(start, stop, per_rank): contiguous shards of ``ceil(n/world)`` frames; trailing ranks may get fewer or none."""
    per = -(-n_frames // world) if n_frames else 0
    start = min(n_frames, rank * per)
    return start, min(n_frames, start + per), per


def assemble(g_rec: np.ndarray, g_cnt: np.ndarray, n_frames: int, person_label: int = 1, nms_threshold: float = 0.4,
             foot=lambda b: (b[0] + b[2] / 2, b[1] + b[3])) -> List[List[Detection]]:
    """Orchestrator side: gathered records -> ``list[list[Detection]]`` in global frame order (person filter + NMS via
    the C-ABI's host routine ``opd_person_nms``; xyxy -> xywh; foot point)."""
    lib = _capi.load_library()
    world, per, Q, _ = g_rec.shape
    recs = np.ascontiguousarray(g_rec, dtype=np.int32).reshape(world * per * Q, 8).view(DET_DTYPE).reshape(world * per, Q).copy()
    counts = np.ascontiguousarray(g_cnt, dtype=np.int32).reshape(world * per).copy()
    rc = lib.opd_person_nms_batch(recs.ctypes.data_as(C.POINTER(_capi.OpdDet)), counts.ctypes.data_as(C.POINTER(C.c_int32)),
                                  world * per, Q, person_label, float(nms_threshold))   # padding slots (count -1) are skipped
    _capi.check(rc, "opd_person_nms_batch")
    out: List[List[Detection]] = []
    for f in range(world * per):
        if counts[f] < 0:
            continue  # padding slot of an uneven shard
        dets = []
        for d in recs[f, :counts[f]]:
            x1, y1, x2, y2 = float(d["x1"]), float(d["y1"]), float(d["x2"]), float(d["y2"])
            bbox = (x1, y1, x2 - x1, y2 - y1)   # (double arithmetic on the fp32 corners, like detector._postprocess_batch)
            dets.append(Detection(bbox=bbox, confidence=float(d["score"]), class_id=person_label, class_name="person",
                                  camera_coords=foot(bbox), query_index=int(d["query_index"])))
        out.append(dets)
    if len(out) != n_frames:
        raise RuntimeError(f"gathered {len(out)} frames, expected {n_frames}")
    return out


class NativeExchange:
    """One ``opd_comm`` (``include/opd_detr.h``) bound to handle 0 of a loaded ``HipDetrDetector``: the records of this rank's frames
    go from the post-process kernel into the communicator's send buffer, ``ncclAllGather`` runs on the handle's stream, and ``wait``
    returns every rank's records from page-locked host memory."""

    SETUP_TIMEOUT_S = 120.0   # ncclCommInitRank blocks until every rank has arrived: a rank whose peers never do must not hang forever

    def __init__(self, detector, rank: int, world: int, unique_id: bytes):
        import os
        import sys
        import threading

        self._lib = _capi.load_library()
        self.detector, self.rank, self.world = detector, rank, world
        self._comm = C.c_void_p()

        def stalled():   # (a collective initialisation cannot be cancelled from outside: the process is the unit that fails)
            print(f"NativeExchange: rank {rank}: communicator set-up did not finish within {self.SETUP_TIMEOUT_S:.0f} s", file=sys.stderr, flush=True)
            os._exit(3)

        dog = threading.Timer(float(os.environ.get("OPD_COMM_TIMEOUT", self.SETUP_TIMEOUT_S)), stalled)
        dog.daemon = True
        dog.start()
        try:
            _capi.check(self._lib.opd_comm_create(unique_id, rank, world, C.c_void_p(detector.model), C.byref(self._comm)), "opd_comm_create")
        finally:
            dog.cancel()
        if not hasattr(detector, "_exchanges"):
            detector._exchanges = []
        detector._exchanges.append(self)   # HipDetrDetector.close() closes its exchanges before it destroys the handles

    @staticmethod
    def available() -> bool:
        """librccl can be resolved in this process (every rank asks, and the ranks agree, BEFORE any of them enters the collective set-up)."""
        return _capi.load_library().opd_comm_available() == 0

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(_capi.OPD_COMM_ID_BYTES)
        _capi.check(_capi.load_library().opd_comm_unique_id(buf), "opd_comm_unique_id")
        return buf.raw

    def gather(self, local_frames: Sequence[np.ndarray], per: int) -> Tuple[np.ndarray, np.ndarray]:
        """-> (records int32 [world][per][Q][8], counts int32 [world][per]) of all ranks; ``per`` frame slots per rank."""
        det, lib = self.detector, self._lib
        Q = det.num_queries
        # Failure symmetry: the all-gather is a collective -- a rank that raised between begin and exchange would leave its peers blocked in
        # it.  So (1) everything that can be checked about the local frames is checked BEFORE the exchange is begun; (2) once begun, the
        # exchange is ALWAYS issued and waited for, whatever happens locally: slots this rank could not fill keep the count -1 that
        # opd_comm_begin wrote, the peers return, and the local error is raised afterwards.
        if len(local_frames) > per:
            raise ValueError(f"{len(local_frames)} local frames do not fit {per} slots per rank")
        for f in local_frames:
            if not isinstance(f, np.ndarray) or f.ndim != 3 or f.shape[2] != 3 or f.dtype != np.uint8:
                raise ValueError("frames must be HxWx3 uint8 BGR arrays")
        _capi.check(lib.opd_comm_begin(self._comm, per), "opd_comm_begin")
        failure = None
        try:
            for s0 in range(0, len(local_frames), det.max_batch):   # the handle's workspace holds max_batch frames
                rec, cnt = C.c_void_p(), C.c_void_p()
                _capi.check(lib.opd_comm_buffers(self._comm, s0, C.byref(rec), C.byref(cnt)), "opd_comm_buffers")
                det.detect_records_at(list(local_frames[s0:s0 + det.max_batch]), rec.value, cnt.value)
        except Exception as e:   # noqa: BLE001 -- re-raised below, after the collective every rank takes part in
            failure = e
        _capi.check(lib.opd_comm_exchange(self._comm), "opd_comm_exchange")
        recs = np.empty((self.world, per, Q, 8), np.int32)
        counts = np.empty((self.world, per), np.int32)
        _capi.check(lib.opd_comm_wait(self._comm, recs.ctypes.data_as(C.POINTER(_capi.OpdDet)), counts.ctypes.data_as(C.POINTER(C.c_int32))),
                    "opd_comm_wait")
        if failure is not None:
            raise failure
        return recs, counts

    def close(self) -> None:
        if self._comm:
            self._lib.opd_comm_destroy(self._comm)
            self._comm = C.c_void_p()
        ex = getattr(self.detector, "_exchanges", None)
        if ex is not None and self in ex:
            ex.remove(self)


class ShardedDetector:
    """``detect_batch`` over all ranks of an initialised ``torch.distributed`` process group (one rank per GPU).

    ``detector``: anything with ``max_batch``, ``num_queries``, ``nms_threshold``, ``_get_foot_position`` and
    ``detect_records_into(frames, records, counts)`` — a loaded ``HipDetrDetector`` bound to this rank's GPU (the CPU tests
    inject a stand-in for the compute).  ``device``: torch device of the exchange buffer: this rank's GPU under RCCL — the
    post-process kernel then writes the records straight into the tensor the all-gather reads — or None (host memory, gloo)."""

    def __init__(self, detector, device: Optional[str] = None, exchange: str = "torch"):
        """``exchange``: "torch" (``dist.all_gather_into_tensor``: RCCL with device tensors, gloo with host tensors) or "native" (the
        C-ABI's ``opd_comm_*``: RCCL on the detector handle's own stream; the unique id travels through the process group once)."""
        if exchange not in ("torch", "native"):
            raise ValueError("exchange must be 'torch' or 'native'")
        self.detector = detector
        self.device = device
        self.exchange = exchange
        self._native: Optional[NativeExchange] = None

    def _native_exchange(self) -> NativeExchange:
        if self._native is None:
            import torch.distributed as dist
            rank, world = dist.get_rank(), dist.get_world_size()
            # every rank or none: agree on availability before any rank enters ncclCommInitRank (a rank without librccl would otherwise
            # raise here while its peers block in the collective)
            flags = [None] * world
            if world > 1:
                dist.all_gather_object(flags, bool(NativeExchange.available()))
            else:
                flags = [bool(NativeExchange.available())]
            if not all(flags):
                raise RuntimeError(f"native exchange unavailable: librccl could not be resolved on rank(s) {[r for r, ok in enumerate(flags) if not ok]}")
            box = [NativeExchange.unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(box, src=0)   # (set-up only: 128 bytes)
            self._native = NativeExchange(self.detector, rank, world, box[0])
        return self._native

    def close(self) -> None:
        if self._native is not None:
            self._native.close()
            self._native = None

    def detect_batch(self, frames: Sequence[np.ndarray]) -> List[List[Detection]]:
        """Every rank passes the SAME global frame list; every rank returns the full result."""
        import torch.distributed as dist

        start, stop, _ = shard_bounds(len(frames), dist.get_rank(), dist.get_world_size())
        return self.detect_shard(list(frames[start:stop]), len(frames))

    def detect_shard(self, local_frames: Sequence[np.ndarray], n_frames: int) -> List[List[Detection]]:
        """A rank passes ONLY its own frames — global frames ``shard_bounds(n_frames, rank, world)[0:2]`` — plus the global frame
        count; every rank returns the full result in global frame order (a rank never needs the other ranks' pixels)."""
        import torch
        import torch.distributed as dist

        rank, world = dist.get_rank(), dist.get_world_size()
        start, stop, per = shard_bounds(n_frames, rank, world)
        if len(local_frames) != stop - start:
            raise ValueError(f"rank {rank} of {world} owns frames [{start}, {stop}) of {n_frames} but was given {len(local_frames)} frames")
        det = self.detector
        Q = det.num_queries
        if self.exchange == "native":
            g_rec, g_cnt = self._native_exchange().gather(local_frames, per)
            return assemble(g_rec, g_cnt, n_frames, nms_threshold=det.nms_threshold, foot=det._get_foot_position)
        nrec = per * Q * 8
        # ONE flat int32 buffer per rank: per x Q records (8 words each), then per counts; -1 marks the padding slots of an
        # uneven shard.  The collective is latency bound: one launch, not two.
        flat = torch.zeros((nrec + per,), dtype=torch.int32, device=self.device or "cpu")
        flat[nrec:] = -1
        if flat.is_cuda:
            # the fills above run on torch's stream, the post-process kernel writes the same tensor from the handle's own
            # (non-blocking) stream: finish the fills first
            torch.cuda.current_stream(flat.device).synchronize()
        rec_view, cnt_view = flat[:nrec].view(per, Q, 8), flat[nrec:]
        for s0 in range(0, stop - start, det.max_batch):   # the handle's workspace holds max_batch frames
            s1 = min(stop - start, s0 + det.max_batch)
            det.detect_records_into(list(local_frames[s0:s1]), rec_view[s0:s1], cnt_view[s0:s1])   # (blocking: returns with the records written)
        gathered = torch.empty((world * flat.shape[0],), dtype=torch.int32, device=flat.device)
        dist.all_gather_into_tensor(gathered, flat)   # concatenated form: accepted by both the RCCL and the gloo backend
        g = gathered.cpu().numpy().reshape(world, flat.shape[0])
        g_rec = np.ascontiguousarray(g[:, :nrec]).reshape(world, per, Q, 8)
        g_cnt = np.ascontiguousarray(g[:, nrec:])
        return assemble(g_rec, g_cnt, n_frames, nms_threshold=det.nms_threshold, foot=det._get_foot_position)
