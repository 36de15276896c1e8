#!/usr/bin/env python3
"""Headline benchmark: frames/sec of the Phase-2 DETR detect path (BASELINE.json: detr-resnet-50, batch 8, 800x1333).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = one pass of the hot path over one batch of synthetic frames per rank: uint8 BGR frames already resident in
HBM -> preprocess -> ResNet-50 -> encoder/decoder -> heads -> device post-process -> detection records; with N > 1
every rank detects its own shard of the global batch (8 frames per rank, weak scaling) and the records are
all-gathered over RCCL/xGMI to every rank (rank 0 = the orchestrator) inside the step.  Rank 0 prints ONE JSON line.

Steps are submitted asynchronously to `--streams` detector handles per GPU (default 3: own stream / workspace / graph, one
shared copy of the weights), so up to three batches of 8 are in flight per GPU; every step's records reach host memory inside
the timed region, which is bracketed by barrier + device synchronisation.  `--streams 1 --sync-steps` is the strictly serial
form (one blocking call per step).  DESIGN.md section 5 has the numbers for both.
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_TFLOPS = 2500.0  # dense fp16/bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md §Chip-level parameters
PEAK_HBM_GBS = 8000.0


def cpu_baseline(path: str, H: int, W: int, batch: int = 4, iters: int = 2) -> dict:
    """The oracle (CPU fp32 restatement, torch CPU backend) on a bounded sample of the same workload."""
    import torch

    from office_person_detection_vit_amd.frames import structured_frame
    from office_person_detection_vit_amd.weights import load_safetensors
    from oracle import detr_oracle as O

    # use the host cores this process may actually run on (the GPU box shares its CPUs between boxes)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            ncpu = max(1, min(ncpu, int(int(quota) / int(period))))
    except Exception:
        pass
    torch.set_num_threads(ncpu)
    w = O.to_torch(load_safetensors(path))
    frames = [structured_frame(H, W, 1234 + i) for i in range(batch)]
    times = []
    for it in range(iters + 1):  # first iteration is the warm-up
        t0 = time.perf_counter()
        pv, pm = O.preprocess(frames)
        lg, bx, _ = O.forward(w, pv, pm)
        O.post_process_object_detection(lg.numpy(), bx.numpy(), 0.5, [(H, W)] * batch)
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    return {"value": round(batch / best, 4), "unit": "frames/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": f"oracle/detr_oracle.py (torch CPU fp32), detr-resnet-50, {batch}x{H}x{W} frames, 1 warm-up + "
                      f"{iters} timed iterations, best iteration {best:.2f} s"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # 0.65 s timed; 20 steps carry ~4 % of pipeline fill / drain
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="frames per GPU per step")
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    ap.add_argument("--arch", choices=["r50", "r101"], default="r50", help="backbone depth (r101 = BASELINE configs[3]; not the headline workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("OPD_BENCH_STREAMS", "3")),
                    help="detector handles (own stream, workspace and graph each) that take the steps in turn: with 2 the low-"
                         "occupancy tail of step i (decoder, heads) overlaps the trunk of step i+1")
    ap.add_argument("--sync-steps", action="store_true",
                    help="one blocking detect call per step (default: steps are submitted asynchronously, results of step i-1 are "
                         "fetched while step i computes)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from office_person_detection_vit_amd import _capi
    from office_person_detection_vit_amd.frames import structured_frame
    from office_person_detection_vit_amd.weights import DetrArch, ensure_weight_file

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # OPD_BENCH_BACKEND=gloo is a REHEARSAL mode for a one-GPU box: every rank uses device 0 and the records travel through
    # host memory; it exercises the multi-rank control flow (barriers, max-over-ranks timing, rank-0 JSON), not RCCL.
    backend = os.environ.get("OPD_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else 0
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    B, H, W = args.batch, args.height, args.width
    cache = os.environ.get("OPD_WEIGHT_CACHE", "/tmp/opd_weights")
    arch = DetrArch.resnet101() if args.arch == "r101" else DetrArch.resnet50()
    if rank == 0:
        path = ensure_weight_file(cache, arch, 0, 1.0, args.arch)
    if world > 1:
        dist.barrier()
    path = ensure_weight_file(cache, arch, 0, 1.0, args.arch)

    lib = _capi.load_library()
    cfg = _capi.OpdConfig(struct_size=C.sizeof(_capi.OpdConfig), max_batch=B, max_height=H, max_width=W, flags=0)
    handle = C.c_void_p()
    _capi.check(lib.opd_detr_create(C.byref(cfg), path.encode(), device_index, C.byref(handle)), "opd_detr_create")
    info = _capi.OpdModelInfo()
    _capi.check(lib.opd_detr_info(handle, C.byref(info)), "opd_detr_info")
    handles = [handle]
    for _ in range(1, max(1, args.streams)):
        hx = C.c_void_p()   # same weights in HBM, own stream / workspace / graph
        _capi.check(lib.opd_detr_clone(handle, C.byref(hx)), "opd_detr_clone")
        handles.append(hx)
    NS = len(handles)
    Q = info.num_queries

    # synthetic office-camera frames, resident in HBM before the timed region (torch = device memory plumbing only)
    frames = np.stack([structured_frame(H, W, 1234 + rank * B + i) for i in range(B)])
    d_frames = torch.from_numpy(frames).cuda()
    # one flat int32 buffer per rank: B*Q records (opd_det = 8 x 4 bytes) followed by the B per-frame counts, so that the
    # path's exchange step is ONE all-gather
    NREC = B * Q * 8
    d_flats = [torch.zeros((NREC + B,), dtype=torch.int32, device="cuda") for _ in range(2 * NS)]   # rotating output buffers
    d_flat = d_flats[0]
    d_records, d_counts = d_flat[:NREC].view(B, Q, 8), d_flat[NREC:]
    gdev = "cuda" if backend == "nccl" else "cpu"
    g_flat = torch.zeros((world * (NREC + B),), dtype=torch.int32, device=gdev) if world > 1 else None
    hw = np.asarray([[H, W]] * B, dtype=np.int32)

    def local_detect():
        rc = lib.opd_detr_detect(handle, C.c_void_p(d_frames.data_ptr()), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_DEVICE,
                                 B, H, W, 0.5, hw.ctypes.data_as(C.c_void_p),
                                 C.cast(C.c_void_p(d_records.data_ptr()), C.POINTER(_capi.OpdDet)),
                                 C.cast(C.c_void_p(d_counts.data_ptr()), C.POINTER(C.c_int32)))
        _capi.check(rc, "opd_detr_detect")

    def collect(buf):
        """records (+ counts) of one finished step to the orchestrator's host memory (the path's one exchange step for N > 1)"""
        if world > 1:
            dist.all_gather_into_tensor(g_flat, buf if backend == "nccl" else buf.cpu())
            g = g_flat.cpu().view(world, NREC + B)
            return g[:, NREC:].reshape(-1), (g[:, :NREC].reshape(world * B, Q, 8) if rank == 0 else None)
        h = buf.cpu()
        return h[NREC:], h[:NREC].view(B, Q, 8)

    def step():
        local_detect()
        return collect(d_flat)

    def submit(i):
        """enqueue step i on the library's stream (forward + post-process into buffer i & 1) and return its ticket"""
        buf = d_flats[i % (2 * NS)]
        ticket = C.c_int()
        rc = lib.opd_detr_detect_async(handles[i % NS], C.c_void_p(d_frames.data_ptr()), _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_DEVICE,
                                       B, H, W, 0.5, hw.ctypes.data_as(C.c_void_p), C.cast(C.c_void_p(buf.data_ptr()), C.POINTER(_capi.OpdDet)),
                                       C.cast(C.c_void_p(buf[NREC:].data_ptr()), C.POINTER(C.c_int32)), C.byref(ticket))
        _capi.check(rc, "opd_detr_detect_async")
        return ticket.value

    def run_steps(n):
        """n steps; every step's records reach host memory.  Pipelined form: step i is submitted before step i-1 is collected."""
        if args.sync_steps:
            out = None
            for _ in range(n):
                out = step()
            return out
        # NS steps stay submitted ahead (one per handle); step i - NS is collected right after step i has been submitted
        out, tickets = None, {}
        for i in range(n + NS):
            if i < n:
                tickets[i] = submit(i)
            j = i - NS
            if j >= 0:
                _capi.check(lib.opd_detr_wait(handles[j % NS], tickets.pop(j)), "opd_detr_wait")
                out = collect(d_flats[j % (2 * NS)])
        return out

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if args.warmup:
        run_steps(args.warmup)
    sync()
    t0 = time.perf_counter()
    counts, _ = run_steps(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=gdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel: HIP event pairs around every launch on the library's stream ------------------
    roof = None
    stage_ms = None
    if rank == 0:
        _capi.check(lib.opd_detr_set_profiling(handle, 1), "opd_detr_set_profiling")
        ms_acc = np.zeros(4)
        fl = np.zeros(4)
        ln = np.zeros(4, dtype=np.int64)
        st_acc = np.zeros(8)
        reps = 3
        for _ in range(reps):
            local_detect()  # rank-local: no collective here (the other ranks have left the timed loop)
            ms4, l4, f4, s8 = (C.c_float * 4)(), (C.c_int32 * 4)(), (C.c_double * 4)(), (C.c_float * 8)()
            _capi.check(lib.opd_detr_kernel_times(handle, ms4, l4, f4), "opd_detr_kernel_times")
            _capi.check(lib.opd_detr_stage_times(handle, s8), "opd_detr_stage_times")
            ms_acc += np.asarray(list(ms4)); fl = np.asarray(list(f4)); ln = np.asarray(list(l4)); st_acc += np.asarray(list(s8))
        ms_avg = ms_acc / reps
        stage_ms = [round(float(v), 4) for v in st_acc / reps]
        _capi.check(lib.opd_detr_set_profiling(handle, 0), "opd_detr_set_profiling")
        # dominant kernel = the implicit-GEMM family (backbone convolutions incl. the fused tails + transformer linears): classes 0 and 1
        k_ms = float(ms_avg[0] + ms_avg[1])
        k_fl = float(fl[0] + fl[1])
        k_n = int(ln[0] + ln[1])
        achieved = k_fl / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        # HBM bytes per launch of the same kernel family from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE
        # in separate runs, gfx950 x2 fetch correction: tools/pmc_traffic.py) — valid for the default 8 x 800x1333 workload
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath) and (B, H, W) == (8, 800, 1333) and args.arch == "r50":
            traffic = round(json.load(open(tpath))["conv_gemm_family"]["bytes_per_launch"])
        roof = {"bound": "mfma", "kernel": "implicit-GEMM family (conv_gemm_dma_kernel, btail_kernel, gemm_ln256_kernel, gemm_k256_kernel, stem_pool2_kernel)", "achieved": round(achieved, 2), "peak": PEAK_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / PEAK_MFMA_TFLOPS, 4), "traffic": traffic,
                "launches_per_step": k_n, "avg_launch_us": round(1e3 * k_ms / max(k_n, 1), 2),
                "flops_per_launch": round(k_fl / max(k_n, 1)),
                "by_class_ms": {"conv": round(float(ms_avg[0]), 4), "linear": round(float(ms_avg[1]), 4),
                                "attention": round(float(ms_avg[2]), 4)}}

    total_frames = B * world * args.steps
    fps = total_frames / elapsed
    if rank == 0:
        out = {
            "metric": f"frames/sec (Phase-2 DETR detect) at {H}x{W} batch {B}",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"facebook/detr-resnet-{50 if args.arch == 'r50' else 101} architecture (seeded synthetic weights), batch {B} per GPU, "
                                   f"{H}x{W} uint8 BGR frames resident in HBM, forward + device post-process, "
                                   + ("blocking steps" if args.sync_steps else
                                      f"steps submitted asynchronously over {NS} detector handle(s), every step's records fetched to host")
                                   + ((", RCCL all-gather of detection records" if backend == "nccl" else f", {backend} REHEARSAL (ranks share device 0)")
                                      if world > 1 else ""),
                       "global_batch": B * world, "parallelism": f"frame-sharded dp{world}",
                       "batches_in_flight_per_gpu": 1 if args.sync_steps else NS},
            "roofline": roof,
            "stage_ms": stage_ms,
            "detections_last_step": int(np.asarray(counts).sum()),
        }
        if (H, W) == (800, 1333) and args.arch == "r50" and roof is not None:
            # SURVEY.md section 8(d) headline: the whole path's algorithmic FLOPs (203.2 GFLOP per r50 frame at 800x1333) x frames/s
            # per GPU against the dense MFMA peak; this one includes every gap, the attention, pre- and post-processing
            path_tf = 203.2e9 * fps / world / 1e12
            roof["whole_path"] = {"flops_per_frame": 203.2e9, "achieved": round(path_tf, 2), "frac": round(path_tf / PEAK_MFMA_TFLOPS, 4)}
        if world == 1 and not args.no_cpu_baseline and args.arch == "r50":
            out["cpu_baseline"] = cpu_baseline(path, H, W)
            out["speedup_vs_cpu_baseline"] = round(fps / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    for hx in handles:
        lib.opd_detr_destroy(hx)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
