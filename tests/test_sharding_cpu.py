"""Multi-rank path on CPU: world_size 2 and 3 ``gloo`` process groups run ``ShardedDetector.detect_batch`` itself — shard
bounds, the per-rank exchange buffer filled through ``detect_records_into``, the ONE all-gather, the orchestrator-side
assembly (person filter + NMS through the C-ABI host routine) — and, on top of it, the tile bookkeeping of BASELINE configs[4]
(the tiles of one 4K frame land on different ranks; every rank merges them back).  Only the per-rank GPU compute is replaced:
a stand-in detector writes records that are a deterministic function of each frame's pixels."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from office_person_detection_vit_amd.sharding import DET_DTYPE, ShardedDetector, assemble, shard_bounds
from office_person_detection_vit_amd.tiling import TiledDetector

Q = 100


def test_shard_bounds_cover_all_frames():
    for n in (0, 1, 7, 8, 9, 64):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                a, b, per = shard_bounds(n, r, world)
                assert 0 <= b - a <= per
                seen += list(range(a, b))
            assert seen == list(range(n))


def _frame_records(frame: np.ndarray):
    """Deterministic fake detections of one frame: a function of its pixels only (frame size, first pixel = a seed)."""
    h, w = frame.shape[:2]
    rng = np.random.default_rng(int(frame[0, 0, 0]) * 65536 + int(frame[0, 0, 1]) * 256 + int(frame[0, 0, 2]))
    n = int(rng.integers(0, Q // 2))
    recs = np.zeros(Q, DET_DTYPE)
    x = rng.uniform(0, 0.8 * w, n)
    y = rng.uniform(0, 0.6 * h, n)
    recs["x1"][:n], recs["y1"][:n] = x, y
    recs["x2"][:n], recs["y2"][:n] = x + rng.uniform(0.02, 0.2, n) * w, y + rng.uniform(0.05, 0.4, n) * h
    recs["score"][:n] = rng.uniform(0.5, 1, n)
    recs["label"][:n] = rng.choice([1, 1, 1, 2], n)
    recs["query_index"][:n] = np.sort(rng.choice(Q, n, replace=False))
    return recs, n


class StandInDetector:
    """The surface ``ShardedDetector`` needs from a detector; the compute is ``_frame_records``."""
    max_batch, num_queries, nms_threshold = 3, Q, 0.4
    calls = 0

    def _get_foot_position(self, b):
        return (b[0] + b[2] / 2, b[1] + b[3])

    def detect_records_into(self, frames, records, counts):
        assert 0 < len(frames) <= self.max_batch and records.shape == (len(frames), Q, 8) and counts.shape == (len(frames),)
        assert records.is_contiguous() and not records.is_cuda
        self.calls += 1
        for i, f in enumerate(frames):
            r, n = _frame_records(f)
            records[i] = torch.from_numpy(r.view(np.int32).reshape(Q, 8))
            counts[i] = n

    def detect_batch(self, frames):   # the single-process expectation: same records, no collective
        rec = np.stack([_frame_records(f)[0] for f in frames]).view(np.int32).reshape(1, len(frames), Q, 8)
        cnt = np.array([[_frame_records(f)[1] for f in frames]], np.int32)
        return assemble(rec, cnt, len(frames), nms_threshold=self.nms_threshold)


def _frames(n, h=48, w=64):
    out = []
    for i in range(n):
        f = np.zeros((h, w, 3), np.uint8)
        f[0, 0] = (i + 1, 7 * i % 251, 3)
        out.append(f)
    return out


def _sig(frames_dets):
    return [[(round(d.bbox[0], 3), round(d.bbox[2], 3), round(d.confidence, 5), d.query_index, d.camera_coords[1]) for d in f] for f in frames_dets]


def _worker(rank, world, port, n_frames, tiled, out_dir, local_only=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stand_in = StandInDetector()
        sharded = ShardedDetector(stand_in)
        if tiled:   # configs[4]: each 4K-like frame -> 2 x 2 overlapping tiles; a frame's tiles straddle the rank boundary
            frames = [np.random.default_rng(50 + i).integers(0, 255, (96, 128, 3), dtype=np.uint8) for i in range(n_frames)]
            dets = TiledDetector(sharded, 2, 2, nms_threshold=0.4).detect_batch(frames)
        elif local_only:   # a rank holds ONLY its own shard's pixels (the other ranks' frames never reach this process)
            a, b, _ = shard_bounds(n_frames, rank, world)
            dets = sharded.detect_shard(_frames(n_frames)[a:b], n_frames)
            with pytest.raises(ValueError):
                sharded.detect_shard(_frames(n_frames)[a:b] + _frames(1), n_frames)   # not this rank's shard size
        else:
            dets = sharded.detect_batch(_frames(n_frames))
        a, b, _ = shard_bounds(n_frames * (4 if tiled else 1), rank, world)
        assert stand_in.calls == -(-(b - a) // stand_in.max_batch)   # this rank computed ITS shard only, in max_batch chunks
        torch.save(_sig(dets), os.path.join(out_dir, f"rank{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n_frames", [(2, 8), (2, 5), (3, 7), (3, 2)])
def test_sharded_detect_batch_gloo(tmp_path, world, n_frames):
    mp.spawn(_worker, args=(world, _free_port(), n_frames, False, str(tmp_path)), nprocs=world, join=True)
    want = _sig(StandInDetector().detect_batch(_frames(n_frames)))
    for r in range(world):
        assert torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) == want   # every rank holds the full, ordered result
    assert any(len(f) for f in want)


@pytest.mark.parametrize("world,n_frames", [(2, 7), (3, 4)])
def test_sharded_detect_shard_local_frames_only_gloo(tmp_path, world, n_frames):
    """`detect_shard`: every rank passes only the frames it owns; every rank still returns the full, ordered result."""
    mp.spawn(_worker, args=(world, _free_port(), n_frames, False, str(tmp_path), True), nprocs=world, join=True)
    want = _sig(StandInDetector().detect_batch(_frames(n_frames)))
    for r in range(world):
        assert torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) == want


@pytest.mark.parametrize("world,n_frames", [(2, 1), (2, 3), (3, 2)])
def test_tiled_frames_over_ranks_gloo(tmp_path, world, n_frames):
    """BASELINE configs[4] bookkeeping: 4 tiles per frame sharded over the ranks (world 2, one frame: tiles 0-1 on rank 0, 2-3 on
    rank 1; world 3, two frames: 8 tiles as 3 + 3 + 2), gathered, re-assembled per frame and merged across the seams — equal to
    the single-process TiledDetector on the same stand-in."""
    mp.spawn(_worker, args=(world, _free_port(), n_frames, True, str(tmp_path)), nprocs=world, join=True)
    frames = [np.random.default_rng(50 + i).integers(0, 255, (96, 128, 3), dtype=np.uint8) for i in range(n_frames)]
    want = _sig(TiledDetector(StandInDetector(), 2, 2, nms_threshold=0.4).detect_batch(frames))
    assert len(want) == n_frames
    for r in range(world):
        assert torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) == want
