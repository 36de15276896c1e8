#!/usr/bin/env python3
"""GPU: time the attention kernel on the encoder / decoder shapes of the benchmark (batch 8, 1050 tokens, fused QKV buffer
with leading dimension 768).  Usage: bench_attn.py [iters]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from office_person_detection_vit_amd import _capi  # noqa: E402

lib = _capi.load_library(test_hooks=True)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(0)
_p = lambda a: a.ctypes.data_as(C.c_void_p)
for name, B, Lq, Lk, sharp in (("encoder self 1050x1050", 8, 1050, 1050, 1.0), ("encoder, sharp scores", 8, 1050, 1050, 3.0),
                               ("decoder cross 100x1050", 8, 100, 1050, 1.0), ("decoder self 100x100", 8, 100, 100, 1.0),
                               ("r101 1080p 2040x2040", 8, 2040, 2040, 1.0)):
    qkv = (rng.standard_normal((B, max(Lq, Lk), 768)) * 1.2).astype(np.float16)
    qkv[:, :, :512] *= np.float16(sharp)
    buf = np.ascontiguousarray(qkv.view(np.uint16))
    q = np.ascontiguousarray(buf[:, :Lq, :])
    us = C.c_float()
    # q / k / v are column blocks 0 / 256 / 512 of the fused buffer: pass offset views as separate uploads with ld 768
    k = np.ascontiguousarray(np.roll(buf[:, :Lk, :], -256, axis=2))
    v = np.ascontiguousarray(np.roll(buf[:, :Lk, :], -512, axis=2))
    _capi.check(lib.opd_test_bench_attention(_p(q), _p(k), _p(v), B, 8, Lq, Lk, 768, 768, 32 ** -0.5, iters, C.byref(us)), "bench")
    print(f"{name:26s} {us.value:7.2f} us", flush=True)
