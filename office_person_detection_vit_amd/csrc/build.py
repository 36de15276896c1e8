"""Build the native libraries in-tree with hipcc for gfx950 (cross-compiles without a GPU):

  libopd_hip.so       the PRODUCT: exports exactly the functions of include/opd_detr.h (-fvisibility=hidden + OPD_API)
  libopd_hip_test.so  the same objects + opd_test_api.o (kernel-level hooks, fusion switches, poison allocator): what tests/ and
                      tools/ load (`_capi.load_library(test_hooks=True)`); never loaded by the package on its own
"""

from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
SOURCES = ["kernels_gemm.hip", "kernels_w8.hip", "kernels_btail.hip", "kernels_btail3.hip", "kernels_rowln.hip", "kernels_attn.hip", "kernels_misc.hip", "kernels_dec.hip", "opd_loader.cpp", "opd_host.cpp", "opd_model.cpp", "opd_comm.cpp", "opd_dispatch.cpp", "opd_test_api.cpp"]
TEST_ONLY = {"opd_test_api.cpp"}
# kernel files with 16-bit operands: ONE source, compiled for fp16 and (-DOPD_ELEM_BF16) for bf16 (opd_elem.h)
ELEM_SOURCES = ["kernels_gemm.hip", "kernels_w8.hip", "kernels_btail.hip", "kernels_btail3.hip", "kernels_rowln.hip", "kernels_attn.hip", "kernels_misc.hip"]
HEADERS = ["opd_kernels.h", "opd_elem.h", "opd_loader.h", "opd_host.h", "opd_model.h", os.path.join("..", "..", "include", "opd_detr.h")]
# code-generation flags of every translation unit, and per file: the attention kernel consumes its S = K.Q^T accumulators with VALU right
# away, so its MFMAs should write VGPRs (the default AGPR form costs 56 v_accvgpr moves per key tile in a VALU-bound loop).
# tools/scan_dma_waits.py imports these: the ISA it checks must be the ISA that ships.
COMMON_FLAGS = ["-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function", "-fvisibility=hidden"]
EXTRA_FLAGS = {"kernels_attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
BF16_FLAGS = ["-DOPD_ELEM_BF16=1"]


def hipcc_path() -> str:
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


LIB = os.path.join(PKG, "libopd_hip.so")
TEST_LIB = os.path.join(PKG, "libopd_hip_test.so")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) >= t for d in deps)


def build(verbose: bool = False, force: bool = False) -> str:
    hipcc = hipcc_path()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    objs = []
    common, extra = COMMON_FLAGS, EXTRA_FLAGS
    jobs = []
    for src in SOURCES:
        sp = os.path.join(HERE, src)
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [sp] + hdrs):
            jobs.append([hipcc] + common + extra.get(src, []) + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", sp, "-o", obj])
        if src in ELEM_SOURCES:   # the bf16 instantiation of the same kernels
            obj16 = os.path.join(objdir, os.path.splitext(src)[0] + "_bf16.o")
            objs.append(obj16)
            if force or _stale(obj16, [sp] + hdrs):
                jobs.append([hipcc] + common + extra.get(src, []) + BF16_FLAGS + ["-c", sp, "-o", obj16])

    def compile_one(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if jobs:   # independent translation units: compile them side by side (bounded by the host's cores)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as pool:
            list(pool.map(compile_one, jobs))
    test_objs = {os.path.join(objdir, os.path.splitext(s)[0] + ".o") for s in TEST_ONLY}
    for lib, members in ((LIB, [o for o in objs if o not in test_objs]), (TEST_LIB, objs)):
        if force or _stale(lib, members):
            cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib] + members + ["-ldl"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
