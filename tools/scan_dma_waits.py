#!/usr/bin/env python3
"""Static check of the compiled kernels (gfx950 ISA): every s_barrier that is reached with LDS-DMA requests issued since the last
`s_waitcnt vmcnt(0)` must be preceded by a wait that counts at most the YOUNGER LDS-DMA requests.  Stores and loads into registers retire
out of order with respect to an older LDS-DMA request (tools/microbench/vmorder.hip), but hipcc assumes one in-order queue and, for a
`__syncthreads()` behind [LDS-DMA requests, register loads], emits e.g. vmcnt(2): "everything but the two youngest loads".  The kernels
therefore drain with an explicit vmcnt(0) (or a hand-counted wait among LDS-DMA requests only) before such barriers; this script lists the
barriers where the last wait in front of them allows more operations in flight than LDS-DMA requests were issued behind the data the barrier
publishes -- conservatively: any non-zero wait with register loads / stores among the operations issued since the last full drain.

What is scanned is what ships: the flag sets come from csrc/build.py (COMMON_FLAGS, the per-file EXTRA_FLAGS, the -DOPD_ELEM_BF16 second
instantiation of every ELEM_SOURCES file), and hipcc is resolved the way build.py resolves it.

No kernel and no barrier is exempted (round 4 skipped the wave-private ring kernels by NAME; round 5 checks them like any other: the one barrier
that then stood out sat behind a tools-only trace store in dec_self_kernel, which now keeps its stamps in registers until the kernel ends).
usage: scan_dma_waits.py [kernels_*.hip ...]   (default: every kernel file; exit code 1 if a suspicious barrier is found)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.environ.get("OPD_SCAN_CSRC") or os.path.join(ROOT, "office_person_detection_vit_amd", "csrc")   # (another revision's sources: a worktree's csrc)
sys.path.insert(0, ROOT)
from office_person_detection_vit_amd.csrc import build as B   # noqa: E402

files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.startswith("kernels_") and f.endswith(".hip"))
bad = checked = 0
for f in files:
    path = f if os.path.isabs(f) else os.path.join(CSRC, f)
    base = os.path.basename(path)
    variants = [("f16", [])] + ([("bf16", B.BF16_FLAGS)] if base in B.ELEM_SOURCES and os.path.exists(os.path.join(CSRC, "opd_elem.h")) else [])
    for tag, vflags in variants:
        with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
            cmd = [B.hipcc_path()] + B.COMMON_FLAGS + B.EXTRA_FLAGS.get(base, []) + vflags + ["-S", "--cuda-device-only", "-I" + CSRC, path, "-o", tmp.name]
            subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
            src = open(tmp.name).read().split("\n")
        fn, seq, verdict = None, [], None   # seq: classes of the vector-memory operations that may be in flight ('D' LDS-DMA request, 'o' load into registers, 's' store)
        for i, l in enumerate(src):
            m = re.match(r"^(_Z\w+):", l)
            if m:
                fn, seq, verdict = m.group(1), [], None
                continue
            if fn is None:
                continue
            t = l.strip()
            if re.match(r"(buffer_load|global_load)\w*\s.*\blds\b", t) or t.startswith("global_load_lds"):
                seq.append("D")
            elif re.match(r"(buffer_load|global_load|flat_load)", t):
                seq.append("o")          # a load into registers
            elif re.match(r"(buffer_store|global_store|buffer_atomic|global_atomic|flat_store)", t):
                seq.append("s")
            mm = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", t)
            if mm:
                n = int(mm.group(1))
                # the wait lets the n youngest operations stay in flight: it proves the older LDS-DMA requests only if those n are all LDS-DMA requests
                young = seq[len(seq) - n:] if n else []
                older_dma = "D" in seq[:len(seq) - n] if n else "D" in seq
                verdict = (n, "o" not in young and "s" not in young, older_dma)
                if n == 0:
                    seq = []
                else:   # register loads older than the n youngest operations have returned (they are not overtaken by younger requests): forget them
                    seq = [c for c in seq[:len(seq) - n] if c != "o"] + young
            if t.startswith("s_barrier"):
                if "D" in seq:
                    n, pure, _ = verdict if verdict else (None, False, True)
                    ok = verdict is not None and pure
                    checked += 1
                    print(f"{'ok ' if ok else 'BAD'} {base} [{tag}] {fn[:70]}: barrier at line {i}, operations since the last drain {''.join(seq)[-40:]}, "
                          f"last wait vmcnt({n})")
                    bad += 0 if ok else 1
                if verdict and verdict[0] == 0:
                    seq = []
print(f"{checked} barrier(s) with LDS-DMA requests in flight checked (fp16 and bf16 instantiations, the shipped flags), none exempted")
print(f"{bad} suspicious barrier(s)")
sys.exit(1 if bad else 0)
