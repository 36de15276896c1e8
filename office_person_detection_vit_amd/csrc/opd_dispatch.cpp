// opd_dispatch.cpp — the launchers of opd_kernels.h whose kernels have 16-bit operands: one call = the fp16 or the bf16 instantiation of the
// same kernel source (opd_elem.h; csrc/build.py compiles every such kernels_*.hip twice), chosen by the launch's dtype.
#include "opd_kernels.h"

#define OPD_PICK(name, dtype, ...) ((dtype) == OPD_DT_BF16 ? name##_bf16(__VA_ARGS__) : name##_f16(__VA_ARGS__))

hipError_t opd_launch_conv_gemm(const ConvGemmParams& p, hipStream_t s) { return OPD_PICK(opd_launch_conv_gemm, p.dtype, p, s); }
hipError_t opd_launch_conv_w8(const ConvGemmParams& p, hipStream_t s) { return OPD_PICK(opd_launch_conv_w8, p.dtype, p, s); }
hipError_t opd_launch_stem_pool_u8(const uint8_t* frames, const int32_t* valid_hw, const f16_t* w, const float* bias, f16_t* out, int B, int H, int W, int OH,
                                   int OW, int PH, int PW, hipStream_t s, int dtype) {
    return OPD_PICK(opd_launch_stem_pool_u8, dtype, frames, valid_hw, w, bias, out, B, H, W, OH, OW, PH, PW, s);
}
hipError_t opd_launch_stem_pool(const f16_t* x4p, const f16_t* w, const float* bias, f16_t* out, int B, int Hp, int Wp, int OH, int OW, int PH, int PW,
                                hipStream_t s, int dtype) {
    return OPD_PICK(opd_launch_stem_pool, dtype, x4p, w, bias, out, B, Hp, Wp, OH, OW, PH, PW, s);
}
hipError_t opd_launch_btail(const BtailParams& p, hipStream_t s) { return OPD_PICK(opd_launch_btail, p.dtype, p, s); }
hipError_t opd_launch_btail256(const BtailParams& p, hipStream_t s) { return OPD_PICK(opd_launch_btail256, p.dtype, p, s); }
hipError_t opd_launch_gemm_ln(const GemmLnParams& p, hipStream_t s) { return OPD_PICK(opd_launch_gemm_ln, p.dtype, p, s); }
hipError_t opd_launch_enc_ffn(const EncFfnParams& p, hipStream_t s) { return OPD_PICK(opd_launch_enc_ffn, p.dtype, p, s); }
hipError_t opd_launch_gemm_k256(const GemmK256Params& p, hipStream_t s) { return OPD_PICK(opd_launch_gemm_k256, p.dtype, p, s); }
hipError_t opd_launch_attention(const AttnParams& p, hipStream_t s) { return OPD_PICK(opd_launch_attention, p.dtype, p, s); }
hipError_t opd_launch_attention_map(const f16_t* q, int ldq, const f16_t* k, int ldk, const int32_t* sel, int nsel, int heads, int Lk, float scale,
                                    const int32_t* key_valid2, int key_row, void* stat, float* out, hipStream_t s, int dtype) {
    return OPD_PICK(opd_launch_attention_map, dtype, q, ldq, k, ldk, sel, nsel, heads, Lk, scale, key_valid2, key_row, stat, out, s);
}
hipError_t opd_launch_preprocess_u8(const uint8_t* frames, f16_t* out, int B, int H, int W, int Hp, int Wp, const int32_t* valid_hw, hipStream_t s, int dtype) {
    return OPD_PICK(opd_launch_preprocess_u8, dtype, frames, out, B, H, W, Hp, Wp, valid_hw, s);
}
hipError_t opd_launch_preprocess_f32(const float* pv, f16_t* out, int B, int H, int W, int Hp, int Wp, const int32_t* valid_hw, hipStream_t s, int dtype) {
    return OPD_PICK(opd_launch_preprocess_f32, dtype, pv, out, B, H, W, Hp, Wp, valid_hw, s);
}
hipError_t opd_launch_maxpool(const f16_t* x, f16_t* out, int B, int H, int W, int C, int OH, int OW, hipStream_t s, int dtype) {
    return OPD_PICK(opd_launch_maxpool, dtype, x, out, B, H, W, C, OH, OW, s);
}
hipError_t opd_launch_layernorm(const float* x, const float* gamma, const float* beta, float* y, f16_t* y16, int rows, hipStream_t s, int dtype) {
    return OPD_PICK(opd_launch_layernorm, dtype, x, gamma, beta, y, y16, rows, s);
}
hipError_t opd_launch_broadcast_rows(const float* c, float* y, f16_t* y16, int rows, hipStream_t s, int dtype) {
    return OPD_PICK(opd_launch_broadcast_rows, dtype, c, y, y16, rows, s);
}
hipError_t opd_launch_reduce_ln(const float* partials, int nsplit, size_t slab_stride, const float* residual, const float* gamma, const float* beta, float* y,
                                f16_t* y16, int rows, hipStream_t s, int dtype) {
    return OPD_PICK(opd_launch_reduce_ln, dtype, partials, nsplit, slab_stride, residual, gamma, beta, y, y16, rows, s);
}
hipError_t opd_launch_reduce_ln_pos(const float* partials, int nsplit, size_t slab_stride, const float* residual, const float* gamma, const float* beta,
                                    float* y, f16_t* y16, int rows, const float* pos, const float* const* pos_ptrs, int period, f16_t* yp16, hipStream_t s,
                                    int dtype) {
    return OPD_PICK(opd_launch_reduce_ln_pos, dtype, partials, nsplit, slab_stride, residual, gamma, beta, y, y16, rows, pos, pos_ptrs, period, yp16, s);
}
hipError_t opd_launch_cast_f16(const float* x, f16_t* y, size_t n, hipStream_t s, int dtype) { return OPD_PICK(opd_launch_cast_f16, dtype, x, y, n, s); }
hipError_t opd_launch_reduce_act16(const float* partials, int nsplit, size_t slab_stride, f16_t* out, size_t n, int relu, hipStream_t s, int dtype) {
    return OPD_PICK(opd_launch_reduce_act16, dtype, partials, nsplit, slab_stride, out, n, relu, s);
}
