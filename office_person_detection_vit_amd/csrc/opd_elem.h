// opd_elem.h — the 16-bit operand type of a kernel translation unit.
//
// Every kernels_*.hip that stores or multiplies 16-bit operands is ONE source compiled TWICE (csrc/build.py): with `elem_t` = fp16 (the
// default mode: fp16 operands, fp32 accumulate) and, under -DOPD_ELEM_BF16, with `elem_t` = bf16 (OPD_FLAG_BF16: the operand type BASELINE.json
// configs[1] names; same MFMA rate on gfx950, 8 mantissa bits instead of 11).  Kernels live in anonymous namespaces, the launchers are
// exported as <name>_f16 / <name>_bf16 (OPD_SYM) and opd_dispatch.cpp picks by the `dtype` of the launch (Params::dtype / last argument).
// Everything else of a kernel — tiling, staging, swizzles, epilogues, waits — is the same text for both types.
#pragma once
#ifdef OPD_ELEM_BF16
typedef __bf16 elem_t;
#define OPD_ELEM_SUFFIX _bf16
#define OPD_MFMA_16x16x32(a, b, ...) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, __VA_ARGS__, 0, 0, 0)
#else
typedef _Float16 elem_t;
#define OPD_ELEM_SUFFIX _f16
#define OPD_MFMA_16x16x32(a, b, ...) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, __VA_ARGS__, 0, 0, 0)
#endif
#define OPD_CAT2(a, b) a##b
#define OPD_CAT(a, b) OPD_CAT2(a, b)
#define OPD_SYM(name) OPD_CAT(name, OPD_ELEM_SUFFIX)
