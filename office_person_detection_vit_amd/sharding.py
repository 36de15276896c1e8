"""Frame-sharded data parallelism over the GPUs of one node: one process per GPU, RCCL all-gather of detections.

The reference has no distributed mode (single process, per-frame loop,
``src/pipeline/phases/detection.py:91-94``); DETR inference has no cross-frame state, so the path shards by frame
(SURVEY.md §8e): rank r of R detects frames ``[r*ceil(n/R), ...)`` on its own GPU, and ONE exchange step — an
all-gather of fixed-size detection records (32 bytes x ``num_queries`` per frame, 3.2 KB/frame) plus per-frame counts —
returns everything to every rank (rank 0 is the orchestrator that feeds tracking/transform).  The payload is tiny, so the
collective is latency-bound; xGMI link bandwidth is irrelevant at this size.

``torch.distributed`` is plumbing only: backend ``nccl`` (= RCCL) with device tensors on the GPU box, ``gloo`` with CPU
tensors in the CPU tests of the exchange/assembly logic.
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


def pack_local(records: np.ndarray, counts: np.ndarray, per_rank: int, num_queries: int) -> Tuple[np.ndarray, np.ndarray]:
    """Pad this rank's ``[n_local, Q]`` records / ``[n_local]`` counts to the fixed per-rank shape (int32 words);
    padding frames carry count -1 so the orchestrator can drop them."""
    rec = np.zeros((per_rank, num_queries, 8), np.int32)
    cnt = np.full((per_rank,), -1, np.int32)
    n = len(counts)
    if n:
        rec[:n] = np.ascontiguousarray(records).view(np.int32).reshape(n, num_queries, 8)
        cnt[:n] = counts
    return rec, cnt


def exchange(rec: np.ndarray, cnt: np.ndarray, device=None):
    """The path's one collective: all-gather of the packed records and counts.  Returns ([R, per, Q, 8], [R, per])."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    # records and counts travel in ONE flat int32 tensor per rank (the collective is latency bound: one launch, not two)
    nrec = rec.size
    flat = torch.from_numpy(np.concatenate([np.ascontiguousarray(rec, dtype=np.int32).reshape(-1),
                                            np.ascontiguousarray(cnt, dtype=np.int32).reshape(-1)]))
    if device is not None:
        flat = flat.to(device)
    # concatenated-along-dim-0 output form: accepted by both the RCCL and the gloo backend
    gathered = torch.empty((world * flat.shape[0],), dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(gathered, flat)
    g = gathered.cpu().numpy().reshape(world, flat.shape[0])
    return (np.ascontiguousarray(g[:, :nrec]).reshape((world,) + tuple(rec.shape)),
            np.ascontiguousarray(g[:, nrec:]).reshape(world, cnt.shape[0]))


def assemble(g_rec: np.ndarray, g_cnt: np.ndarray, n_frames: int, person_label: int = 1, nms_threshold: float = 0.4,
             foot=lambda b: (b[0] + b[2] / 2, b[1] + b[3])) -> List[List[Detection]]:
    """Orchestrator side: gathered records -> ``list[list[Detection]]`` in global frame order (person filter + NMS via
    the C-ABI's host routine ``opd_person_nms``; xyxy -> xywh; foot point)."""
    lib = _capi.load_library()
    world, per, Q, _ = g_rec.shape
    out: List[List[Detection]] = []
    for r in range(world):
        for i in range(per):
            n = int(g_cnt[r, i])
            if n < 0:
                continue  # padding slot of an uneven shard
            recs = np.ascontiguousarray(g_rec[r, i]).view(DET_DTYPE).reshape(Q).copy()
            kept = lib.opd_person_nms(recs.ctypes.data_as(C.POINTER(_capi.OpdDet)), n, person_label, float(nms_threshold))
            if kept < 0:
                _capi.check(kept, "opd_person_nms")
            dets = []
            for k in range(kept):
                d = recs[k]
                bbox = (float(d["x1"]), float(d["y1"]), float(d["x2"] - d["x1"]), float(d["y2"] - d["y1"]))
                dets.append(Detection(bbox=bbox, confidence=float(d["score"]), class_id=person_label, class_name="person",
                                      camera_coords=foot(bbox), query_index=int(d["query_index"])))
            out.append(dets)
    if len(out) != n_frames:
        raise RuntimeError(f"gathered {len(out)} frames, expected {n_frames}")
    return out


class ShardedDetector:
    """``detect_batch`` over all ranks of an initialised ``torch.distributed`` process group (one rank per GPU)."""

    def __init__(self, detector, device: Optional[str] = None):
        self.detector = detector          # a loaded HipDetrDetector bound to this rank's GPU
        self.device = device              # torch device of the collective's tensors (None = CPU/gloo)

    def detect_batch(self, frames: Sequence[np.ndarray]) -> List[List[Detection]]:
        """Every rank passes the SAME global frame list; every rank returns the full result."""
        import torch.distributed as dist

        rank, world = dist.get_rank(), dist.get_world_size()
        start, stop, per = shard_bounds(len(frames), rank, world)
        det = self.detector
        Q = det._info.num_queries
        if stop > start:
            parts, cparts = [], []
            for s0 in range(start, stop, det.max_batch):  # the handle's workspace holds max_batch frames
                chunk = list(frames[s0:min(stop, s0 + det.max_batch)])
                recs, counts, _ = det._detect_records(chunk)
                parts.append(np.frombuffer(recs, dtype=DET_DTYPE).reshape(len(chunk), Q).copy())
                cparts.append(np.frombuffer(counts, dtype=np.int32).copy())
            local, local_counts = np.concatenate(parts), np.concatenate(cparts)
        else:
            local, local_counts = np.zeros((0, Q), DET_DTYPE), np.zeros((0,), np.int32)
        rec, cnt = pack_local(local, local_counts, per, Q)
        g_rec, g_cnt = exchange(rec, cnt, self.device)
        return assemble(g_rec, g_cnt, len(frames), nms_threshold=det.nms_threshold, foot=det._get_foot_position)
