#!/usr/bin/env python3
"""Static check of the compiled kernels (gfx950 ISA): every s_barrier that is reached with LDS-DMA requests issued since the last
`s_waitcnt vmcnt(0)` must be preceded by a wait that counts at most the YOUNGER LDS-DMA requests.  Stores and loads into registers retire
out of order with respect to an older LDS-DMA request (tools/microbench/vmorder.hip), but hipcc assumes one in-order queue and, for a
`__syncthreads()` behind [LDS-DMA requests, register loads], emits e.g. vmcnt(2): "everything but the two youngest loads".  The kernels
therefore drain with an explicit vmcnt(0) (or a hand-counted wait among LDS-DMA requests only) before such barriers; this script lists the
barriers where the last wait in front of them allows more operations in flight than LDS-DMA requests were issued behind the data the barrier
publishes -- conservatively: any non-zero wait with register loads / stores among the operations issued since the last full drain.
Wave-private rings (kernels_dec.hip, enc_ffn_kernel) wait by hand and publish nothing through barriers: their barriers are listed as
"private" when the file says so (-p pattern).
usage: scan_dma_waits.py [kernels_*.hip ...]   (default: every kernel file; exit code 1 if a suspicious barrier is found)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "office_person_detection_vit_amd", "csrc")
PRIVATE = re.compile(r"enc_ffn_kernel|dec_qkv_kernel|dec_self_kernel|dec_cross_out_kernel|dec_ffn_kernel")   # wave-private rings: hand-counted waits
files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.startswith("kernels_") and f.endswith(".hip"))
bad = 0
for f in files:
    path = f if os.path.isabs(f) else os.path.join(CSRC, f)
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I" + CSRC, path, "-o", tmp.name],
                              stderr=subprocess.DEVNULL)
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
            if "D" in seq and not PRIVATE.search(fn):
                n, pure, _ = verdict if verdict else (None, False, True)
                ok = verdict is not None and pure
                print(f"{'ok ' if ok else 'BAD'} {os.path.basename(path)} {fn[:70]}: barrier at line {i}, operations since the last drain {''.join(seq)[-40:]}, "
                      f"last wait vmcnt({n})")
                bad += 0 if ok else 1
            if verdict and verdict[0] == 0:
                seq = []
print(f"{bad} suspicious barrier(s)")
sys.exit(1 if bad else 0)
