"""ctypes binding of libopd_hip.so — the only way the package reaches the GPU.

There is NO CPU fallback: if the shared library is missing (not built) or no HIP device is visible, the calls raise.
The declarations mirror ``include/opd_detr.h`` one to one.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libopd_hip.so")
TEST_LIB_PATH = os.path.join(_HERE, "libopd_hip_test.so")   # the same objects + csrc/opd_test_api.cpp: tests/ and tools/ only

OPD_PIXELS_U8_BGR_HWC = 0
OPD_PIXELS_F32_NCHW = 1
OPD_MEM_HOST = 0
OPD_MEM_DEVICE = 1
OPD_MEM_HOST_PIXELS_DEVICE_OUT = 2
OPD_FLAG_NO_GRAPH = 1
OPD_FLAG_MULTI_STREAM = 2
OPD_FLAG_BF16 = 4
OPD_COMM_ID_BYTES = 128
OPD_OK, OPD_EINVAL, OPD_EIO, OPD_ESCHEMA, OPD_EHIP, OPD_ENOMEM, OPD_ESTATE = 0, -1, -2, -3, -4, -5, -6   # include/opd_detr.h


class OpdConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("max_batch", C.c_int32), ("max_height", C.c_int32),
                ("max_width", C.c_int32), ("flags", C.c_int32), ("reserved", C.c_int32 * 3)]


class OpdKernelStat(C.Structure):   # opd_kernel_stat (include/opd_detr.h)
    _fields_ = [("name", C.c_char * 96), ("launches", C.c_int32), ("ms", C.c_float), ("flops", C.c_double)]


class OpdDet(C.Structure):
    _fields_ = [("x1", C.c_float), ("y1", C.c_float), ("x2", C.c_float), ("y2", C.c_float), ("score", C.c_float),
                ("label", C.c_int32), ("query_index", C.c_int32), ("frame", C.c_int32)]


class OpdModelInfo(C.Structure):
    _fields_ = [("depths", C.c_int32 * 4), ("d_model", C.c_int32), ("heads", C.c_int32), ("ffn_dim", C.c_int32),
                ("encoder_layers", C.c_int32), ("decoder_layers", C.c_int32), ("num_queries", C.c_int32),
                ("num_classes_plus1", C.c_int32), ("max_batch", C.c_int32), ("max_height", C.c_int32),
                ("max_width", C.c_int32), ("device_ordinal", C.c_int32), ("weight_bytes_device", C.c_int64),
                ("workspace_bytes_device", C.c_int64)]


# name -> (restype, argtypes): every symbol include/opd_detr.h declares
API = {
    "opd_detr_create": (C.c_int, [C.POINTER(OpdConfig), C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "opd_detr_destroy": (None, [C.c_void_p]),
    "opd_detr_clone": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "opd_detr_info": (C.c_int, [C.c_void_p, C.POINTER(OpdModelInfo)]),
    "opd_detr_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "opd_detr_forward_ragged": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "opd_detr_forward_resized": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p] * 3),
    "opd_detr_resize_u8": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]),
    "opd_detr_detect_resized": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_float, C.POINTER(OpdDet), C.POINTER(C.c_int32)]),
    "opd_detr_detect_frames": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)] + [C.c_int] * 6 + [C.c_float, C.POINTER(OpdDet), C.POINTER(C.c_int32)]),
    "opd_detr_detect_frames_features": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)] + [C.c_int] * 5 + [C.c_float, C.c_int, C.POINTER(OpdDet), C.POINTER(C.c_int32),
                                                   C.POINTER(C.c_float)]),
    "opd_detr_postprocess": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.POINTER(OpdDet), C.POINTER(C.c_int32)]),
    "opd_detr_detect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                  C.c_void_p, C.POINTER(OpdDet), C.POINTER(C.c_int32)]),
    "opd_detr_detect_ragged": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_float, C.c_void_p, C.POINTER(OpdDet), C.POINTER(C.c_int32)]),
    "opd_detr_detect_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                        C.POINTER(OpdDet), C.POINTER(C.c_int32), C.POINTER(C.c_int)]),
    "opd_detr_wait": (C.c_int, [C.c_void_p, C.c_int]),
    "opd_person_nms": (C.c_int, [C.POINTER(OpdDet), C.c_int, C.c_int, C.c_float]),
    "opd_person_nms_batch": (C.c_int, [C.POINTER(OpdDet), C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_float]),
    "opd_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "opd_host_free": (None, [C.c_void_p]),
    "opd_similarity_matrix": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p]),
    "opd_detr_roi_features": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "opd_detr_attention_map": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "opd_detr_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "opd_detr_stage_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "opd_detr_kernel_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "opd_detr_kernel_table": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "opd_comm_available": (C.c_int, []),
    "opd_comm_unique_id": (C.c_int, [C.c_void_p]),
    "opd_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "opd_comm_attach": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "opd_comm_destroy": (None, [C.c_void_p]),
    "opd_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "opd_comm_begin": (C.c_int, [C.c_void_p, C.c_int]),
    "opd_comm_detect": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "opd_comm_buffers": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "opd_comm_exchange": (C.c_int, [C.c_void_p]),
    "opd_comm_wait": (C.c_int, [C.c_void_p, C.POINTER(OpdDet), C.POINTER(C.c_int32)]),
    "opd_last_error": (C.c_char_p, []),
    "opd_version": (C.c_char_p, []),
}

# kernel-level test hooks (csrc/opd_test_api.cpp); not part of the boundary
TEST_API = {
    "opd_test_conv_gemm": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 15),
    "opd_test_gemm_splitk_ln": (C.c_int, [C.c_void_p] * 8 + [C.c_int] * 3),
    "opd_test_gemm_k256": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 5),
    "opd_test_gemm_ln": (C.c_int, [C.c_void_p] * 8 + [C.c_int] * 2),
    "opd_test_bench_gemm_ln": (C.c_int, [C.c_int] * 4 + [C.POINTER(C.c_float)]),
    "opd_test_gemm_ln_deep": (C.c_int, [C.c_void_p] * 7 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] * 3),
    "opd_test_set_heads2": (C.c_int, [C.c_int]),
    "opd_test_enc_ffn": (C.c_int, [C.c_void_p] * 9 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p] * 2 + [C.c_int] * 2 + [C.c_void_p] * 5 + [C.c_int]),
    "opd_test_bench_enc_ffn": (C.c_int, [C.c_int] * 6 + [C.POINTER(C.c_float)]),
    "opd_test_bench_conv": (C.c_int, [C.c_int] * 11 + [C.POINTER(C.c_float)]),
    "opd_test_trace_btail": (C.c_int, [C.c_int] * 6 + [C.POINTER(C.c_ulonglong), C.c_int, C.POINTER(C.c_int)]),
    "opd_test_trace_conv": (C.c_int, [C.c_int] * 10 + [C.POINTER(C.c_ulonglong), C.c_int, C.POINTER(C.c_int)]),
    "opd_test_btail": (C.c_int, [C.c_void_p] * 10 + [C.c_int] * 6),
    "opd_test_bench_btail": (C.c_int, [C.c_int] * 8 + [C.POINTER(C.c_float)]),
    "opd_test_btail_repeat": (C.c_int, [C.c_void_p] * 8 + [C.c_int] * 6 + [C.POINTER(C.c_int)]),
    "opd_test_attention": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_float]),
    "opd_test_layernorm": (C.c_int, [C.c_void_p] * 5 + [C.c_int]),
    "opd_test_maxpool": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 6),
    "opd_test_preprocess_u8": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 5 + [C.c_void_p]),
    "opd_test_attention_masked": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_float, C.c_void_p, C.c_int]),
    "opd_test_stem2": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 5),
    "opd_test_stem_pool": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 7),
    "opd_test_set_fuse_stem_pool": (C.c_int, [C.c_void_p, C.c_int]),
    "opd_test_set_pos_shadow": (C.c_int, [C.c_void_p, C.c_int]),
    "opd_test_stem_pool_u8": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 3),
    "opd_test_set_fuse_btail": (C.c_int, [C.c_void_p, C.c_int]),
    "opd_test_set_fuse_gemm_ln": (C.c_int, [C.c_void_p, C.c_int]),
    "opd_test_f32_to_f16": (C.c_uint16, [C.c_float]),
    "opd_test_f16_to_f32": (C.c_float, [C.c_uint16]),
    "opd_test_normalise_key": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int]),
    "opd_test_inspect_checkpoint": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32)]),
    "opd_test_resize_coeffs": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "opd_test_valid_prefix": (C.c_int, [C.c_int] * 3),
    "opd_test_sine_pos_embed": (C.c_int, [C.c_int] * 5 + [C.c_void_p]),
    "opd_test_conv_dual": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 13),
    "opd_test_btail_sc": (C.c_int, [C.c_void_p] * 11 + [C.c_int] * 3),
    "opd_test_btail_chain": (C.c_int, [C.c_void_p] * 13 + [C.c_int] * 4),
    "opd_test_bench_attention": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_float, C.c_int, C.POINTER(C.c_float)]),
    "opd_test_trace_attention": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_float, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "opd_test_heads": (C.c_int, [C.c_void_p] * 13 + [C.c_int] * 2),
    "opd_test_postprocess": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_float, C.c_void_p, C.c_void_p]),
    "opd_test_roi_features": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 3 + [C.c_void_p]),
    "opd_test_set_conv_flags": (C.c_int, [C.c_int]),
    "opd_test_set_gemm_ln_kloop": (C.c_int, [C.c_int]),
    "opd_test_set_graph_guard": (C.c_int, [C.c_int]),
    "opd_test_set_alloc_poison": (C.c_int, [C.c_int]),
    "opd_test_check_redzones": (C.c_int, [C.c_void_p]),
    "opd_test_set_taps": (C.c_int, [C.c_void_p, C.c_int]),
    "opd_test_read_taps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_int]),
    "opd_test_round_f16_diffused": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "opd_test_dec_qkv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int] * 2 + [C.c_void_p] * 4),
    "opd_test_dec_self": (C.c_int, [C.c_void_p] * 10 + [C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "opd_test_attention_split": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "opd_test_dec_cross_out": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p]),
    "opd_test_dec_ffn": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p]),
    "opd_test_bench_dec": (C.c_int, [C.c_int] * 6 + [C.POINTER(C.c_float)]),
    "opd_test_heads_fused": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 13 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "opd_test_set_fused_dec": (C.c_int, [C.c_void_p, C.c_int]),
    "opd_test_set_elem_bf16": (C.c_int, [C.c_int]),
    "opd_test_trace_dec_self": (C.c_int, [C.c_int, C.c_int, C.c_void_p]),
}

_lib: Optional[C.CDLL] = None
_lib_has_hooks = False


def load_library(test_hooks: bool = False) -> C.CDLL:
    """dlopen libopd_hip.so (built in-tree by ``csrc/build.py``) and attach prototypes.  Raises if it is missing.

    ``test_hooks=True`` (tests/ and tools/ only; also ``OPD_TEST_HOOKS=1`` in the environment) loads ``libopd_hip_test.so`` instead:
    the same objects plus the ``opd_test_*`` hooks of ``csrc/opd_test_api.cpp``.  One process uses ONE of the two (the hooks flip
    process-wide switches of the library they live in), so asking for the hooks after the product library has been loaded raises."""
    global _lib, _lib_has_hooks
    test_hooks = test_hooks or os.environ.get("OPD_TEST_HOOKS") == "1"
    if _lib is not None:
        if test_hooks and not _lib_has_hooks:
            raise RuntimeError("the product library is already loaded in this process; load the test build first "
                               "(_capi.load_library(test_hooks=True) or OPD_TEST_HOOKS=1)")
        return _lib
    path = TEST_LIB_PATH if test_hooks else LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            f"HIP extension not built: {path} is missing (run `python -c 'import __graft_entry__ as g; g.build()'`). "
            "This package has no CPU fallback.")
    lib = C.CDLL(path)
    for table in (API, TEST_API) if test_hooks else (API,):
        for name, (res, args) in table.items():
            fn = getattr(lib, name)  # AttributeError here = the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
    _lib, _lib_has_hooks = lib, test_hooks
    return lib


def last_error() -> str:
    return load_library().opd_last_error().decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")
