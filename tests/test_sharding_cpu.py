"""Multi-rank path on CPU: world_size 2 and 3 ``gloo`` process groups exercise shard bounds, the packed all-gather
exchange and the orchestrator-side assembly (person filter + NMS through the C-ABI host routine).  The per-rank GPU
compute is replaced by seeded synthetic records — the collective and the bookkeeping are what is under test."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from office_person_detection_vit_amd.sharding import DET_DTYPE, assemble, exchange, pack_local, shard_bounds


def test_shard_bounds_cover_all_frames():
    for n in (0, 1, 7, 8, 9, 64):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                a, b, per = shard_bounds(n, r, world)
                assert 0 <= b - a <= per
                seen += list(range(a, b))
            assert seen == list(range(n))


def _frame_records(frame_idx: int, Q: int):
    """Deterministic fake per-frame detections: count and boxes are functions of the global frame index."""
    rng = np.random.default_rng(1000 + frame_idx)
    n = int(rng.integers(0, Q // 2))
    recs = np.zeros(Q, DET_DTYPE)
    x = rng.uniform(0, 1000, n)
    y = rng.uniform(0, 600, n)
    recs["x1"][:n], recs["y1"][:n] = x, y
    recs["x2"][:n], recs["y2"][:n] = x + rng.uniform(30, 200, n), y + rng.uniform(60, 300, n)
    recs["score"][:n] = rng.uniform(0.5, 1, n)
    recs["label"][:n] = rng.choice([1, 1, 1, 2], n)
    recs["query_index"][:n] = np.sort(rng.choice(Q, n, replace=False))
    recs["frame"][:n] = frame_idx
    return recs, n


def _worker(rank, world, port, n_frames, Q, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, b, per = shard_bounds(n_frames, rank, world)
        local = [(_frame_records(i, Q)) for i in range(a, b)]
        recs = np.stack([r for r, _ in local]) if local else np.zeros((0, Q), DET_DTYPE)
        counts = np.array([n for _, n in local], np.int32)
        rec, cnt = pack_local(recs, counts, per, Q)
        g_rec, g_cnt = exchange(rec, cnt)
        dets = assemble(g_rec, g_cnt, n_frames)
        sig = [[(round(d.bbox[0], 3), round(d.confidence, 5), d.query_index) for d in f] for f in dets]
        torch.save(sig, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n_frames", [(2, 8), (2, 5), (3, 7)])
def test_gather_and_assemble_gloo(tmp_path, world, n_frames):
    Q = 100
    mp.spawn(_worker, args=(world, _free_port(), n_frames, Q, str(tmp_path)), nprocs=world, join=True)
    # single-process expectation: the same records assembled without any collective
    recs = np.stack([_frame_records(i, Q)[0] for i in range(n_frames)])
    counts = np.array([_frame_records(i, Q)[1] for i in range(n_frames)], np.int32)
    rec, cnt = pack_local(recs, counts, n_frames, Q)
    want = assemble(rec[None], cnt[None], n_frames)
    want_sig = [[(round(d.bbox[0], 3), round(d.confidence, 5), d.query_index) for d in f] for f in want]
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"))
        assert got == want_sig  # every rank holds the full, ordered result
    assert any(len(f) for f in want_sig)
