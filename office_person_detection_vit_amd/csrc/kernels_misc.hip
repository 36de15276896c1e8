// kernels_misc.hip — HBM-bound element-wise / small kernels of the DETR detect path (gfx950).
//
//   preprocess   SURVEY.md §8 a1: BGR->RGB, x*(1/255), (x-mean)/std   (HF:models/detr/image_processing_detr.py:639-668)
//   maxpool      a3: MaxPool2d(k3,s2,p1)                              (HF:models/resnet/modeling_resnet.py:84)
//   layernorm    a9/a12: nn.LayerNorm(256), eps 1e-5                  (HF:models/detr/modeling_detr.py:606,629-639)
//   heads        a13: class_labels_classifier + DetrMLPPredictionHead + sigmoid (HF:...modeling_detr.py:1284-1300,1410-1411)
//   postprocess  a14: HF post_process_object_detection               (HF:models/detr/image_processing_detr.py:805-856)
//   roi_features a17: FeatureExtractor.extract_roi_features           (src/tracking/feature_extractor.py:39-88)
#include <hip/hip_runtime.h>
#include <math.h>
#include <vector>
#include "opd_kernels.h"
#include "opd_elem.h"

typedef elem_t half8 __attribute__((ext_vector_type(8)));
typedef elem_t half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

namespace {

// ImageNet statistics, RGB order (HF:utils/constants.py IMAGENET_DEFAULT_MEAN/STD)
__constant__ float c_mean[3] = {0.485f, 0.456f, 0.406f};
__constant__ float c_std[3] = {0.229f, 0.224f, 0.225f};

// One thread per OUTPUT pixel of the zero-bordered NHWC4 image [B][Hp][Wp][4] (image at offset (3,3)): 3 bytes in (BGR),
// 8 bytes out (RGB0 fp16).  Same op order as the oracle: (float(u8) * (1/255) - mean) / std.  The border is the stem
// convolution's zero padding, materialised so that the stem's LDS-DMA needs no per-tap bounds logic.
// valid_hw (nullable): [B][2] = (h, w) of each frame inside the H x W canvas; pixels outside are written as zeros, which
// is what HF's pad-after-normalise produces for a ragged batch (HF:models/detr/image_processing_detr.py:639-668).
__global__ void preprocess_u8_kernel(const uint8_t* __restrict__ in, f16_t* __restrict__ out, int B, int H, int W, int Hp, int Wp,
                                     const int32_t* __restrict__ valid_hw) {
#pragma clang fp contract(off)  // keep mul / sub / div separately rounded, like the reference's tensor ops
    // one thread = 4 consecutive pixels of a padded row: 12 source bytes, two 16-byte stores (one pixel per thread moved
    // 1.7 TB/s: 35 us at batch 8; the arithmetic per pixel is unchanged)
    const int groups = (Wp + 3) >> 2;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * Hp * groups) return;
    const int xg = (int)(i % groups);
    const size_t t = i / groups;
    const int yp = (int)(t % Hp);
    const int b = (int)(t / Hp);
    const int y = yp - 3;
    const int vh = valid_hw ? valid_hw[2 * b] : H, vw = valid_hw ? valid_hw[2 * b + 1] : W;
    const bool row_ok = (unsigned)y < (unsigned)vh;
    const uint8_t* srow = in + ((size_t)b * H + (row_ok ? y : 0)) * W * 3;
    f16_t* orow = out + (((size_t)b * Hp + yp) * Wp) * 4;
    const float k = 1.0f / 255.0f;
    half4 o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int xp = xg * 4 + j, x = xp - 3;
        o[j][0] = o[j][1] = o[j][2] = o[j][3] = (elem_t)0.f;
        if (row_ok && (unsigned)x < (unsigned)vw) {
            const uint8_t* s = srow + (size_t)x * 3;
            const float bl = (float)s[0], g = (float)s[1], r = (float)s[2];
            o[j][0] = (elem_t)((r * k - c_mean[0]) / c_std[0]);
            o[j][1] = (elem_t)((g * k - c_mean[1]) / c_std[1]);
            o[j][2] = (elem_t)((bl * k - c_mean[2]) / c_std[2]);
        }
    }
    if (xg * 4 + 3 < Wp && (Wp & 1) == 0) {   // rows are 8 Wp bytes long: the 32-byte groups are 16-byte aligned when Wp is even ...
        typedef elem_t half8v __attribute__((ext_vector_type(8)));
        half8v a, c;
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[q] = o[0][q]; a[4 + q] = o[1][q]; c[q] = o[2][q]; c[4 + q] = o[3][q]; }
        *reinterpret_cast<half8v*>(orow + (size_t)xg * 16) = a;
        *reinterpret_cast<half8v*>(orow + (size_t)xg * 16 + 8) = c;
    } else {                                    // ... otherwise (and for the last partial group) pixel by pixel
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (xg * 4 + j < Wp) *reinterpret_cast<half4*>(orow + (size_t)(xg * 4 + j) * 4) = o[j];
    }
}

__global__ void preprocess_f32_kernel(const float* __restrict__ pv, f16_t* __restrict__ out, int B, int H, int W, int Hp, int Wp,
                                      const int32_t* __restrict__ valid_hw) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * Hp * Wp) return;
    const int xp = (int)(i % Wp);
    const size_t t = i / Wp;
    const int yp = (int)(t % Hp);
    const int b = (int)(t / Hp);
    const int y = yp - 3, x = xp - 3;
    half4 o;
    o[0] = o[1] = o[2] = o[3] = (elem_t)0.f;
    const int vh = valid_hw ? valid_hw[2 * b] : H, vw = valid_hw ? valid_hw[2 * b + 1] : W;
    if ((unsigned)y < (unsigned)vh && (unsigned)x < (unsigned)vw) {
        const size_t HW = (size_t)H * W;
        const float* s = pv + (size_t)b * 3 * HW + (size_t)y * W + x;
        o[0] = (elem_t)s[0];
        o[1] = (elem_t)s[HW];
        o[2] = (elem_t)s[2 * HW];
    }
    *reinterpret_cast<half4*>(out + i * 4) = o;
}

// Bilinear resize of uint8 HxWx3 frames, bit-exact with Pillow's two-pass 8-bit resampler, which is what HF's image
// processor runs on the host (HF:models/detr/image_processing_detr.py:424-436 -> PIL Image.resize(BILINEAR);
// Pillow src/libImaging/Resample.c: ImagingResampleHorizontal_8bpc / ImagingResampleVertical_8bpc).  Both passes use
// 22-bit fixed-point coefficients (tables built on the host by opd_resize_coeffs with Pillow's formulas) and each pass
// rounds to uint8: out = clip8((2^21 + sum_k pixel_k * coeff_k) >> 22).  One thread per output pixel computes the few
// horizontally resampled source rows it needs (rounded, like the intermediate image) and combines them vertically.
__global__ void resize_bilinear_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int B, int h, int w, int oh,
                                          int ow, const int32_t* __restrict__ bh, const int32_t* __restrict__ kh, int ksh,
                                          const int32_t* __restrict__ bv, const int32_t* __restrict__ kv, int ksv) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * oh * ow) return;
    const int xo = (int)(i % ow);
    const size_t t = i / ow;
    const int yo = (int)(t % oh);
    const int b = (int)(t / oh);
    const int xmin = bh[2 * xo], xcnt = bh[2 * xo + 1];
    const int ymin = bv[2 * yo], ycnt = bv[2 * yo + 1];
    const int32_t* ckh = kh + (size_t)xo * ksh;
    const int32_t* ckv = kv + (size_t)yo * ksv;
    const int half = 1 << 21;
    int a0 = half, a1 = half, a2 = half;
    for (int j = 0; j < ycnt; ++j) {
        const uint8_t* row = in + (((size_t)b * h + ymin + j) * w + xmin) * 3;
        int s0 = half, s1 = half, s2 = half;
        for (int k = 0; k < xcnt; ++k) {
            const int c = ckh[k];
            s0 += (int)row[3 * k] * c;
            s1 += (int)row[3 * k + 1] * c;
            s2 += (int)row[3 * k + 2] * c;
        }
        s0 >>= 22; s1 >>= 22; s2 >>= 22;   // arithmetic shift, then clip8
        s0 = s0 < 0 ? 0 : (s0 > 255 ? 255 : s0);
        s1 = s1 < 0 ? 0 : (s1 > 255 ? 255 : s1);
        s2 = s2 < 0 ? 0 : (s2 > 255 ? 255 : s2);
        const int c = ckv[j];
        a0 += s0 * c; a1 += s1 * c; a2 += s2 * c;
    }
    a0 >>= 22; a1 >>= 22; a2 >>= 22;
    uint8_t* o = out + i * 3;
    o[0] = (uint8_t)(a0 < 0 ? 0 : (a0 > 255 ? 255 : a0));
    o[1] = (uint8_t)(a1 < 0 ? 0 : (a1 > 255 ? 255 : a1));
    o[2] = (uint8_t)(a2 < 0 ? 0 : (a2 > 255 ? 255 : a2));
}

// One thread per (output pixel, 8-channel group); 9 x 16-byte loads, 1 x 16-byte store.
__global__ void maxpool_kernel(const f16_t* __restrict__ x, f16_t* __restrict__ out, int B, int H, int W, int C, int OH,
                               int OW) {
    const int cg = C >> 3;
    const size_t total = (size_t)B * OH * OW * cg;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c8 = (int)(i % cg);
    size_t r = i / cg;
    const int ow = (int)(r % OW);
    r /= OW;
    const int oh = (int)(r % OH);
    const int b = (int)(r / OH);
    half8 m;
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = (elem_t)(-65504.f);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh * 2 - 1 + kh;
        if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int iw = ow * 2 - 1 + kw;
            if ((unsigned)iw >= (unsigned)W) continue;
            const half8 v = *reinterpret_cast<const half8*>(x + (((size_t)b * H + ih) * W + iw) * C + c8 * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] = v[j] > m[j] ? v[j] : m[j];
        }
    }
    *reinterpret_cast<half8*>(out + (((size_t)b * OH + oh) * OW + ow) * C + c8 * 8) = m;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One wave per row of 256: lane holds 4 consecutive elements; two-pass (mean, then centred variance) in fp32.
__global__ __launch_bounds__(256) void layernorm256_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ y,
                                                           f16_t* __restrict__ y16, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float4v v = *reinterpret_cast<const float4v*>(x + (size_t)row * 256 + lane * 4);
    const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
    const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
    const float var = wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const float4v g = *reinterpret_cast<const float4v*>(gamma + lane * 4);
    const float4v bb = *reinterpret_cast<const float4v*>(beta + lane * 4);
    float4v o;
    o[0] = d0 * rstd * g[0] + bb[0];
    o[1] = d1 * rstd * g[1] + bb[1];
    o[2] = d2 * rstd * g[2] + bb[2];
    o[3] = d3 * rstd * g[3] + bb[3];
    if (y) *reinterpret_cast<float4v*>(y + (size_t)row * 256 + lane * 4) = o;
    if (y16) {
        half4 h;
        h[0] = (elem_t)o[0]; h[1] = (elem_t)o[1]; h[2] = (elem_t)o[2]; h[3] = (elem_t)o[3];
        *reinterpret_cast<half4*>(y16 + (size_t)row * 256 + lane * 4) = h;
    }
}

// y[row][:] = c[:] for every row (fp32 and an fp16 copy): the decoder's state after the self-attention block of layer 0, which does
// not depend on the input (opd_model.cpp::build_weights, "dec0").
__global__ __launch_bounds__(256) void broadcast_rows256_kernel(const float* __restrict__ c, float* __restrict__ y, f16_t* __restrict__ y16, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float4v v = *reinterpret_cast<const float4v*>(c + lane * 4);
    *reinterpret_cast<float4v*>(y + (size_t)row * 256 + lane * 4) = v;
    half4 h;
    h[0] = (elem_t)v[0]; h[1] = (elem_t)v[1]; h[2] = (elem_t)v[2]; h[3] = (elem_t)v[3];
    *reinterpret_cast<half4*>(y16 + (size_t)row * 256 + lane * 4) = h;
}

// Split-K reduction + residual + LayerNorm: one wave per row of 256, slabs summed in slice order (deterministic).
__global__ __launch_bounds__(256) void reduce_ln256_kernel(const float* __restrict__ partials, int nsplit, size_t slab_stride,
                                                           const float* __restrict__ residual, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ y,
                                                           f16_t* __restrict__ y16, int rows, const float* __restrict__ pos,
                                                           const float* const* __restrict__ pos_ptrs, int period, f16_t* __restrict__ yp16) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const size_t o = (size_t)row * 256 + lane * 4;
    float4v v = *reinterpret_cast<const float4v*>(partials + o);
    for (int z = 1; z < nsplit; ++z) v += *reinterpret_cast<const float4v*>(partials + z * slab_stride + o);
    if (residual) v += *reinterpret_cast<const float4v*>(residual + o);
    float4v out = v;
    if (gamma) {
        const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
        const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
        const float var = wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        const float4v g = *reinterpret_cast<const float4v*>(gamma + lane * 4);
        const float4v bb = *reinterpret_cast<const float4v*>(beta + lane * 4);
        out[0] = d0 * rstd * g[0] + bb[0];
        out[1] = d1 * rstd * g[1] + bb[1];
        out[2] = d2 * rstd * g[2] + bb[2];
        out[3] = d3 * rstd * g[3] + bb[3];
    }
    if (y) *reinterpret_cast<float4v*>(y + o) = out;
    if (y16) {
        half4 h;
        h[0] = (elem_t)out[0]; h[1] = (elem_t)out[1]; h[2] = (elem_t)out[2]; h[3] = (elem_t)out[3];
        *reinterpret_cast<half4*>(y16 + o) = h;
    }
    if (yp16) {   // second fp16 shadow: the row PLUS its position embedding (what the q / k projections read); pos [period][256] per frame
        const int fr = row / period, pr = row - fr * period;
        const float* prow = (pos_ptrs ? pos_ptrs[fr] : pos) + (size_t)pr * 256 + lane * 4;
        const float4v pe = *reinterpret_cast<const float4v*>(prow);
        half4 h;
        h[0] = (elem_t)(out[0] + pe[0]); h[1] = (elem_t)(out[1] + pe[1]); h[2] = (elem_t)(out[2] + pe[2]); h[3] = (elem_t)(out[3] + pe[3]);
        *reinterpret_cast<half4*>(yp16 + o) = h;
    }
}

__global__ void cast_f16_kernel(const float* __restrict__ x, f16_t* __restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) reinterpret_cast<elem_t*>(y)[i] = (elem_t)x[i];
}

// Plan-build-time fp32 GEMM (pos-embedding folds): one thread per output, K-loop in order; not on the hot path.
__global__ void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ Wt, const float* __restrict__ bias,
                                float* __restrict__ C, int M, int N, int K, int ldc) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y;
    if (n >= N || m >= M) return;
    const float* a = A + (size_t)m * K;
    const float* w = Wt + (size_t)n * K;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(a[k], w[k], acc);
    C[(size_t)m * ldc + n] = acc + (bias ? bias[n] : 0.f);
}

// Heads (HF:models/detr/modeling_detr.py:1284-1300, 1317-1322, 1410-1411): class logits and the box MLP + sigmoid for the decoder
// states, fp32 throughout — on the matrix pipe: v_mfma_f32_16x16x4_f32 is an exact fp32 fma chain (k-ordered, one rounding per
// product) at the fp32 vector rate, without the vector pipe's loads-per-FMA problem.  One 512-thread workgroup per 16 decoder rows
// (the MFMA's 16 columns); a wave owns 16-channel output tiles (weights = A operand straight from L2: [in][out] storage gives
// lane (k, n) a coalesced read; rows = B operand from LDS) and chains the 64 MFMAs of a tile's K = 256 on one accumulator, two tiles
// interleaved per wave.  The decoder's final LayerNorm (two-pass fp32, as layernorm256_kernel) runs on the rows first when
// ln_gamma is given.  Batch 8 (800 rows, 50 workgroups): 30 us, against 35 us for the VALU form of rounds 1-2 (4 rows per
// 1024-thread block, every weight loaded once per 4 rows); what remains is six L2 round trips for the weights.
constexpr int HEAD_ROWS = 16;
constexpr int HEAD_LD = 260;   // fp32 words per LDS row (256 + 4: rows 8 apart do not share a bank)
typedef float float4m __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void heads_kernel(HeadParams p) {
    __shared__ float h[HEAD_ROWS][HEAD_LD];
    __shared__ float t1[HEAD_ROWS][HEAD_LD];
    __shared__ float part[8][64];
    const int row0 = blockIdx.x * HEAD_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {   // rows -> LDS: 32 threads per row, 8 consecutive columns each; optional LayerNorm by the same 32 threads
        const int r = tid >> 5, c0 = (tid & 31) * 8;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = row0 + r < p.rows ? p.hs[(size_t)(row0 + r) * 256 + c0 + j] : 0.f;
        auto ln_rows = [&](const float* __restrict__ gamma, const float* __restrict__ beta) {   // two-pass fp32 LayerNorm by the row's 32 threads
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
            const float mean = s * (1.0f / 256.0f);
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[j] -= mean; q += v[j] * v[j]; }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o);
            const float rstd = 1.0f / sqrtf(q * (1.0f / 256.0f) + 1e-5f);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] * rstd * gamma[c0 + j] + beta[c0 + j];
        };
        if (p.partials) {   // fused decoder: the last layer's FFN arrives as partial sums (kernels_dec.hip::dec_ffn_kernel): + b2, summed in order, LN3
            const size_t rc = (size_t)(row0 + r < p.rows ? row0 + r : p.rows - 1) * 256 + c0;
            float4m pa[16], pb[16];
#pragma unroll
            for (int sp = 0; sp < 16; ++sp) {   // every slab requested before the first one is needed (a dependent load costs ~1 us)
                const float* ps = p.partials + (size_t)(sp < p.nsplit ? sp : p.nsplit - 1) * p.rows * 256 + rc;
                pa[sp] = *reinterpret_cast<const float4m*>(ps);
                pb[sp] = *reinterpret_cast<const float4m*>(ps + 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += p.ffn_b2[c0 + j];
#pragma unroll
            for (int sp = 0; sp < 16; ++sp)
                if (sp < p.nsplit) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] += pa[sp][j]; v[4 + j] += pb[sp][j]; }
                }
            if (row0 + r >= p.rows) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = 0.f;
            }
            ln_rows(p.ln3_gamma, p.ln3_beta);
        }
        if (p.ln_gamma) ln_rows(p.ln_gamma, p.ln_beta);
#pragma unroll
        for (int j = 0; j < 8; ++j) h[r][c0 + j] = v[j];
    }
    __syncthreads();
    // out[m][n] = act(sum_k in[m][k] * wt[k][n] + bias[n]) for the workgroup's 16 rows m: D[row = channel][col = m] per 16-channel
    // tile; lane (kq = lane >> 4, i = lane & 15) feeds A[i][kq] = wt[k0 + kq][n0 + i] and B[kq][i] = in[i][k0 + kq], and receives
    // channels n0 + 4 kq + r of row i.
    const int kq = lane >> 4, li = lane & 15;
    auto layer = [&](const float (*in)[HEAD_LD], const float* __restrict__ wt, const int N, const int ntiles, auto&& emit) {
        for (int nt = wave; nt < ntiles; nt += 16) {   // this wave's tiles nt and nt + 8, interleaved (two accumulator chains)
            const int nA = nt * 16 + li, nB = (nt + 8) * 16 + li;
            const bool okA = nA < N, okB = nt + 8 < ntiles && nB < N;
            float4m accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
            // 2 x 32 weight operands (half the reduction of the tile pair) are requested before their first MFMA: the kernel is a
            // latency chain (every weight comes from L2, ~1 us away), so two round trips per tile pair instead of sixteen (46 us)
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                float a[32], b2[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    const int k = (half * 32 + u) * 4 + kq;
                    a[u] = okA ? wt[(size_t)k * N + nA] : 0.f;
                    b2[u] = okB ? wt[(size_t)k * N + nB] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    const float x = in[li][(half * 32 + u) * 4 + kq];
                    accA = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], x, accA, 0, 0, 0);
                    accB = __builtin_amdgcn_mfma_f32_16x16x4f32(b2[u], x, accB, 0, 0, 0);
                }
            }
            emit(nt * 16 + kq * 4, accA);
            if (nt + 8 < ntiles) emit((nt + 8) * 16 + kq * 4, accB);
        }
    };
    // class logits: lane holds channels n .. n + 3 of row li
    layer(h, p.wc, p.ncls, (p.ncls + 15) / 16, [&](const int n, const float4m acc) {
        if (row0 + li < p.rows)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < p.ncls) p.logits[(size_t)(row0 + li) * p.ncls + n + r] = acc[r] + p.bc[n + r];
    });
    // box MLP: 256 -> 256 -> 256 -> 4, ReLU between, sigmoid at the end
    layer(h, p.w1, 256, 16, [&](const int n, const float4m acc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float v = acc[r] + p.b1[n + r]; t1[li][n + r] = v > 0.f ? v : 0.f; }
    });
    __syncthreads();   // t1 complete; every wave is done reading h
    layer(t1, p.w2, 256, 16, [&](const int n, const float4m acc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float v = acc[r] + p.b2[n + r]; h[li][n + r] = v > 0.f ? v : 0.f; }   // h is free: second hidden layer
    });
    __syncthreads();
    {   // last layer: 64 outputs (row = lane >> 2, coordinate = lane & 3), the reduction split over the 8 waves (32 k each), summed in order
        const int r = lane >> 2, c = lane & 3;
        float a = 0.f;
#pragma unroll 8
        for (int kk = 0; kk < 32; ++kk) a = fmaf(h[r][wave * 32 + kk], p.w3[(wave * 32 + kk) * 4 + c], a);
        part[wave][lane] = a;
    }
    __syncthreads();
    if (tid < 64) {
        const int r = tid >> 2, c = tid & 3;
        float a = part[0][tid];
#pragma unroll
        for (int w = 1; w < 8; ++w) a += part[w][tid];
        if (row0 + r < p.rows) p.boxes[(size_t)(row0 + r) * 4 + c] = 1.0f / (1.0f + expf(-(a + p.b3[c])));
    }
}

struct DetRec {
    float x1, y1, x2, y2, score;
    int32_t label, query_index, frame;
};

// One block (1024 threads = 16 waves) per frame.  Phase 1: wave w takes queries w, w+16, ... (at most 8); it issues the
// loads of ALL its queries first (one L2 round trip instead of one per query), then its lanes stride the classes with
// wave-level max / sum / first-argmax reductions.  Phase 2: one thread per query converts the box and compacts the kept
// queries in query order via a block prefix sum.  (One thread per query with serial class loops: 24 us.)
__global__ __launch_bounds__(1024) void postprocess_kernel(PostParams p) {
    __shared__ int flags[128];
    __shared__ float s_score[128];
    __shared__ int s_label[128];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        float v0[8], v1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int qq = wave + i * 16;
            const float* lg = p.logits + ((size_t)b * p.Q + (qq < p.Q ? qq : 0)) * p.ncls;
            v0[i] = lane < p.ncls ? lg[lane] : -INFINITY;
            v1[i] = lane + 64 < p.ncls ? lg[lane + 64] : -INFINITY;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int qq = wave + i * 16;
            float mx = fmaxf(v0[i], v1[i]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            const float e0 = lane < p.ncls ? expf(v0[i] - mx) : 0.f;
            const float e1 = lane + 64 < p.ncls ? expf(v1[i] - mx) : 0.f;
            float sum = e0 + e1, best = -1.f;
            int bl = 0x7fffffff;
            if (lane < p.ncls - 1) { best = e0; bl = lane; }
            if (lane + 64 < p.ncls - 1 && e1 > best) { best = e1; bl = lane + 64; }   // ascending class order: first maximum
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                sum += __shfl_xor(sum, o);
                const float ob = __shfl_xor(best, o);
                const int ol = __shfl_xor(bl, o);
                if (ob > best || (ob == best && ol < bl)) { best = ob; bl = ol; }   // ties -> lowest class index
            }
            if (lane == 0 && qq < p.Q) { s_score[qq] = best / sum; s_label[qq] = bl; }
        }
    }
    __syncthreads();
    const int q = threadIdx.x;
    float score = 0.f, x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f;
    int label = 0, keep = 0;
    if (q < p.Q) {
        score = s_score[q];
        label = s_label[q];
        const float* bx = p.boxes + ((size_t)b * p.Q + q) * 4;
        const float h = (float)p.orig_hw[b * 2], w = (float)p.orig_hw[b * 2 + 1];
        x1 = (bx[0] - 0.5f * bx[2]) * w;
        y1 = (bx[1] - 0.5f * bx[3]) * h;
        x2 = (bx[0] + 0.5f * bx[2]) * w;
        y2 = (bx[1] + 0.5f * bx[3]) * h;
        keep = score > p.threshold ? 1 : 0;
    }
    if (q < 128) flags[q] = keep;
    __syncthreads();
    if (q >= 128) return;
    int pos = 0;
    for (int i = 0; i < q; ++i) pos += flags[i];
    if (keep) {
        DetRec* rec = reinterpret_cast<DetRec*>(p.records) + (size_t)b * p.Q + pos;
        rec->x1 = x1; rec->y1 = y1; rec->x2 = x2; rec->y2 = y2; rec->score = score;
        rec->label = label; rec->query_index = q; rec->frame = b;
    }
    if (q == 127) p.counts[b] = pos + keep;
}

// One block per ROI: mean over the ROI window of the [h][w][256] map, then L2 normalise.  1024 threads = 16 groups of 64 lanes; a lane owns 4
// channels (one 16-byte load per position), group g takes the window positions g, g + 16, ... in row-major order, and the 16 partial sums of
// a channel are added in group order: a fixed summation order whatever the window.  (The first form -- 256 threads, one channel each, a serial
// loop over the whole window -- took ~90 us for a large box: up to 1050 dependent-latency iterations on one CU.)
__device__ __forceinline__ void roi_pool_l2(const float* __restrict__ enc, float* __restrict__ out_row, const int x0, const int y0, const int x1,
                                            const int y1, const int w, float (&part)[16][256], float (&red)[4]) {
    const int t = threadIdx.x, g = t >> 6, l = t & 63;
    const int bw = x1 - x0, n = (y1 - y0) * bw;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = g; p < n; p += 16) {
        const int dy = p / bw, dx = p - dy * bw;
        const float4 v = *reinterpret_cast<const float4*>(enc + ((size_t)(y0 + dy) * w + (x0 + dx)) * 256 + 4 * l);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(&part[g][4 * l]) = acc;
    __syncthreads();
    if (t < 256) {
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum += part[k][t];
        const float mean = sum / (float)n;
        const float ss = wave_sum(mean * mean);
        if ((t & 63) == 0) red[t >> 6] = ss;
        part[0][t] = mean;   // (row 0 has been read by this thread only)
    }
    __syncthreads();
    if (t < 256) {
        const float norm = sqrtf(red[0] + red[1] + red[2] + red[3]);
        out_row[t] = part[0][t] / (norm + 1e-8f);
    }
}
__global__ __launch_bounds__(1024) void roi_features_kernel(const float* __restrict__ enc, const int32_t* __restrict__ rois,
                                                            float* __restrict__ out, int h, int w) {
    __shared__ float part[16][256];
    __shared__ float red[4];
    const int r = blockIdx.x;
    roi_pool_l2(enc, out + (size_t)r * 256, rois[r * 4], rois[r * 4 + 1], rois[r * 4 + 2], rois[r * 4 + 3], w, part, red);
}
// The same for the RECORDS the post-process kernel left on the device (opd_detr_detect_frames_features): block (i, b) takes record i of frame
// b; a record of class `label` gets its feature in row `query_index` of out[b] (other rows are not written).  Box -> map window as
// opd_detr_roi_features computes it on the host from the (x, y, w, h) float32 box of a `Detection` (feature_extractor.py:68-78): the width is
// the float32 difference, the scaling runs in double, the casts truncate.
__global__ __launch_bounds__(1024) void roi_features_rec_kernel(const float* __restrict__ enc, const DetRec* __restrict__ recs,
                                                               const int32_t* __restrict__ counts, const int32_t* __restrict__ orig_hw, int label,
                                                               float* __restrict__ out, int Q, int h, int w) {
    __shared__ float part[16][256];
    __shared__ float red[4];
    const int i = blockIdx.x, b = blockIdx.y;
    if (i >= counts[b]) return;
    const DetRec r = recs[(size_t)b * Q + i];
    if (r.label != label) return;
    const double oh = (double)orig_hw[2 * b], ow = (double)orig_hw[2 * b + 1];
    const double x = (double)r.x1, y = (double)r.y1, bw = (double)(r.x2 - r.x1), bh = (double)(r.y2 - r.y1);
    int x0 = (int)((x / ow) * w), y0 = (int)((y / oh) * h);
    int x1 = (int)(((x + bw) / ow) * w), y1 = (int)(((y + bh) / oh) * h);
    x0 = max(0, min(x0, w - 1)); y0 = max(0, min(y0, h - 1));
    x1 = max(x0 + 1, min(x1, w)); y1 = max(y0 + 1, min(y1, h));
    roi_pool_l2(enc + (size_t)b * h * w * 256, out + ((size_t)b * Q + r.query_index) * 256, x0, y0, x1, y1, w, part, red);
}

// Cross-attention map (get_attention_map): mean over heads and over the selected queries of softmax(q . k^T * scale) for ONE frame and
// ONE decoder layer, fp32 arithmetic on the fp16 q / k the forward left on the device.  Pass 1: one wave per (selected query, head) row
// -> row maximum and sum of exponentials.  Pass 2: one thread per key sums the normalised weights over the rows in a fixed order.
__global__ __launch_bounds__(256) void attn_map_rowstat_kernel(const f16_t* __restrict__ q, int ldq, const f16_t* __restrict__ k, int ldk,
                                                               const int32_t* __restrict__ sel, int nsel, int heads, int Lk, float scale,
                                                               const int32_t* __restrict__ key_valid2, int key_row, float2* __restrict__ stat) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);   // row = selected query x head
    if (row >= nsel * heads) return;
    const int lane = threadIdx.x & 63, h = row % heads, qi = sel[row / heads];
    float qv[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) qv[d] = (float)reinterpret_cast<const elem_t*>(q)[(size_t)qi * ldq + h * 32 + d];
    const int vr = key_valid2 ? key_valid2[0] : 0x7fffffff, vc = key_valid2 ? key_valid2[1] : 0x7fffffff;
    float mx = -INFINITY;
    for (int key = lane; key < Lk; key += 64) {
        const int kr = key / key_row, kc = key - kr * key_row;
        if (kr >= vr || kc >= vc) continue;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) s = fmaf(qv[d], (float)reinterpret_cast<const elem_t*>(k)[(size_t)key * ldk + h * 32 + d], s);
        mx = fmaxf(mx, s * scale);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int key = lane; key < Lk; key += 64) {
        const int kr = key / key_row, kc = key - kr * key_row;
        if (kr >= vr || kc >= vc) continue;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) s = fmaf(qv[d], (float)reinterpret_cast<const elem_t*>(k)[(size_t)key * ldk + h * 32 + d], s);
        sum += expf(s * scale - mx);
    }
    sum = wave_sum(sum);
    if (lane == 0) stat[row] = make_float2(mx, sum);
}
__global__ __launch_bounds__(256) void attn_map_mean_kernel(const f16_t* __restrict__ q, int ldq, const f16_t* __restrict__ k, int ldk,
                                                            const int32_t* __restrict__ sel, int nsel, int heads, int Lk, float scale,
                                                            const int32_t* __restrict__ key_valid2, int key_row, const float2* __restrict__ stat,
                                                            float* __restrict__ out) {
    const int key = blockIdx.x * 256 + threadIdx.x;
    if (key >= Lk) return;
    const int vr = key_valid2 ? key_valid2[0] : 0x7fffffff, vc = key_valid2 ? key_valid2[1] : 0x7fffffff;
    const int kr = key / key_row, kc = key - kr * key_row;
    float acc = 0.f;
    if (kr < vr && kc < vc) {
        for (int row = 0; row < nsel * heads; ++row) {
            const int h = row % heads, qi = sel[row / heads];
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d)
                s = fmaf((float)reinterpret_cast<const elem_t*>(q)[(size_t)qi * ldq + h * 32 + d],
                         (float)reinterpret_cast<const elem_t*>(k)[(size_t)key * ldk + h * 32 + d], s);
            const float2 st = stat[row];
            acc += expf(s * scale - st.x) / st.y;
        }
    }
    out[key] = acc / (float)(nsel * heads);
}

inline unsigned blocks_for(size_t n, unsigned threads) { return (unsigned)((n + threads - 1) / threads); }

}  // namespace

hipError_t OPD_SYM(opd_launch_attention_map)(const f16_t* q, int ldq, const f16_t* k, int ldk, const int32_t* sel, int nsel, int heads, int Lk, float scale,
                                    const int32_t* key_valid2, int key_row, void* stat, float* out, hipStream_t stream) {
    if (nsel <= 0 || heads <= 0 || Lk <= 0 || key_row <= 0) return hipErrorInvalidValue;
    OPD_LAUNCH(attn_map_rowstat_kernel, dim3((nsel * heads + 3) / 4), dim3(256), 0, stream, q, ldq, k, ldk, sel, nsel, heads, Lk, scale,
                       key_valid2, key_row, reinterpret_cast<float2*>(stat));
    OPD_LAUNCH(attn_map_mean_kernel, dim3((Lk + 255) / 256), dim3(256), 0, stream, q, ldq, k, ldk, sel, nsel, heads, Lk, scale, key_valid2,
                       key_row, reinterpret_cast<const float2*>(stat), out);
    return hipGetLastError();
}

hipError_t OPD_SYM(opd_launch_preprocess_u8)(const uint8_t* frames, f16_t* out, int B, int H, int W, int Hp, int Wp, const int32_t* valid_hw,
                                    hipStream_t stream) {
    if (Hp < H + 6 || Wp < W + 6) return hipErrorInvalidValue;
    const size_t ngroups = (size_t)B * Hp * ((Wp + 3) / 4);
    OPD_LAUNCH(preprocess_u8_kernel, dim3(blocks_for(ngroups, 256)), dim3(256), 0, stream, frames, out, B, H, W, Hp, Wp, valid_hw);
    return hipGetLastError();
}

hipError_t OPD_SYM(opd_launch_preprocess_f32)(const float* pv, f16_t* out, int B, int H, int W, int Hp, int Wp, const int32_t* valid_hw,
                                     hipStream_t stream) {
    if (Hp < H + 6 || Wp < W + 6) return hipErrorInvalidValue;
    const size_t npix = (size_t)B * Hp * Wp;
    OPD_LAUNCH(preprocess_f32_kernel, dim3(blocks_for(npix, 256)), dim3(256), 0, stream, pv, out, B, H, W, Hp, Wp, valid_hw);
    return hipGetLastError();
}

#ifndef OPD_ELEM_BF16   // (no 16-bit operands: defined once)
hipError_t opd_launch_resize_u8(const uint8_t* in, uint8_t* out, int B, int h, int w, int oh, int ow, const int32_t* bounds_h,
                                const int32_t* coeff_h, int ksize_h, const int32_t* bounds_v, const int32_t* coeff_v, int ksize_v,
                                hipStream_t stream) {
    if (B <= 0 || h <= 0 || w <= 0 || oh <= 0 || ow <= 0) return hipErrorInvalidValue;
    const size_t npix = (size_t)B * oh * ow;
    OPD_LAUNCH(resize_bilinear_u8_kernel, dim3(blocks_for(npix, 256)), dim3(256), 0, stream, in, out, B, h, w, oh, ow, bounds_h,
                       coeff_h, ksize_h, bounds_v, coeff_v, ksize_v);
    return hipGetLastError();
}
#endif

hipError_t OPD_SYM(opd_launch_maxpool)(const f16_t* x, f16_t* out, int B, int H, int W, int C, int OH, int OW,
                              hipStream_t stream) {
    if (C % 8 != 0) return hipErrorInvalidValue;
    const size_t total = (size_t)B * OH * OW * (C / 8);
    OPD_LAUNCH(maxpool_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, stream, x, out, B, H, W, C, OH, OW);
    return hipGetLastError();
}

hipError_t OPD_SYM(opd_launch_layernorm)(const float* x, const float* gamma, const float* beta, float* y, f16_t* y16, int rows,
                                hipStream_t stream) {
    if (rows <= 0) return hipErrorInvalidValue;
    OPD_LAUNCH(layernorm256_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, x, gamma, beta, y, y16, rows);
    return hipGetLastError();
}

hipError_t OPD_SYM(opd_launch_broadcast_rows)(const float* c, float* y, f16_t* y16, int rows, hipStream_t stream) {
    if (rows <= 0 || !c || !y || !y16) return hipErrorInvalidValue;
    OPD_LAUNCH(broadcast_rows256_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, c, y, y16, rows);
    return hipGetLastError();
}

hipError_t OPD_SYM(opd_launch_reduce_ln)(const float* partials, int nsplit, size_t slab_stride, const float* residual,
                                const float* gamma, const float* beta, float* y, f16_t* y16, int rows, hipStream_t stream) {
    return OPD_SYM(opd_launch_reduce_ln_pos)(partials, nsplit, slab_stride, residual, gamma, beta, y, y16, rows, nullptr, nullptr, 0, nullptr, stream);
}

// ... plus a second fp16 output yp16 = fp16(y + pos[frame][row % period]) (pos: one table, or pos_ptrs: one table per frame)
hipError_t OPD_SYM(opd_launch_reduce_ln_pos)(const float* partials, int nsplit, size_t slab_stride, const float* residual, const float* gamma,
                                    const float* beta, float* y, f16_t* y16, int rows, const float* pos, const float* const* pos_ptrs,
                                    int period, f16_t* yp16, hipStream_t stream) {
    if (rows <= 0 || nsplit < 1 || (yp16 && (period <= 0 || (!pos && !pos_ptrs)))) return hipErrorInvalidValue;
    OPD_LAUNCH(reduce_ln256_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, partials, nsplit, slab_stride, residual,
                       gamma, beta, y, y16, rows, pos, pos_ptrs, period, yp16);
    return hipGetLastError();
}

// Split-K reduction of a convolution: fp32 slabs summed in slice order (deterministic; slab 0 carries the bias), ReLU, one rounding to the
// 16-bit operand type.  8 elements per thread (two 16-byte loads per slab, one 16-byte store).
static __global__ __launch_bounds__(256) void reduce_act16_kernel(const float* __restrict__ partials, int nsplit, size_t slab_stride, f16_t* __restrict__ out,
                                                           size_t n8, int relu) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const size_t o = i * 8;
    float4v a = *reinterpret_cast<const float4v*>(partials + o), b = *reinterpret_cast<const float4v*>(partials + o + 4);
    for (int z = 1; z < nsplit; ++z) {
        a += *reinterpret_cast<const float4v*>(partials + z * slab_stride + o);
        b += *reinterpret_cast<const float4v*>(partials + z * slab_stride + o + 4);
    }
    typedef elem_t half8v __attribute__((ext_vector_type(8)));
    half8v h;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        h[r] = (elem_t)(relu && a[r] < 0.f ? 0.f : a[r]);
        h[4 + r] = (elem_t)(relu && b[r] < 0.f ? 0.f : b[r]);
    }
    *reinterpret_cast<half8v*>(reinterpret_cast<elem_t*>(out) + o) = h;
}

// Diagnostic tap (opd_test_set_taps): position-weighted 64-bit sum of a buffer's 32-bit words, one partial per block.
static __global__ void checksum_kernel(const uint32_t* __restrict__ buf, size_t nwords, unsigned long long* __restrict__ slots) {
    __shared__ unsigned long long part[256];
    unsigned long long acc = 0ull;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x)
        acc += (unsigned long long)buf[i] * (unsigned long long)((i * 2654435761ull + 1ull) | 1ull);
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) slots[blockIdx.x] = part[0];
}

#ifndef OPD_ELEM_BF16   // (no 16-bit operands: defined once)
hipError_t opd_launch_checksum(const void* buf, size_t bytes, unsigned long long* slots, hipStream_t stream) {
    OPD_LAUNCH(checksum_kernel, dim3(OPD_TAP_BLOCKS), dim3(256), 0, stream, reinterpret_cast<const uint32_t*>(buf), bytes / 4, slots);
    return hipGetLastError();
}
#endif

hipError_t OPD_SYM(opd_launch_reduce_act16)(const float* partials, int nsplit, size_t slab_stride, f16_t* out, size_t n, int relu, hipStream_t stream) {
    if (nsplit < 1 || n == 0 || (n % 8) != 0 || (slab_stride % 4) != 0) return hipErrorInvalidValue;
    OPD_LAUNCH(reduce_act16_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, stream, partials, nsplit, slab_stride, out, n / 8, relu);
    return hipGetLastError();
}

hipError_t OPD_SYM(opd_launch_cast_f16)(const float* x, f16_t* y, size_t n, hipStream_t stream) {
    OPD_LAUNCH(cast_f16_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream, x, y, n);
    return hipGetLastError();
}

#ifndef OPD_ELEM_BF16   // (no 16-bit operands: defined once)
hipError_t opd_launch_gemm_f32(const float* A, const float* Wt, const float* bias, float* C, int M, int N, int K, int ldc,
                               hipStream_t stream) {
    OPD_LAUNCH(gemm_f32_kernel, dim3((N + 63) / 64, M), dim3(64), 0, stream, A, Wt, bias, C, M, N, K, ldc);
    return hipGetLastError();
}
#endif

#ifndef OPD_ELEM_BF16   // (no 16-bit operands: defined once)
hipError_t opd_launch_heads(const HeadParams& p, hipStream_t stream) {
    if (p.wc_f && p.w1_f && p.w2_f && p.ncls <= 128) return opd_launch_heads2(p, stream);   // split fp16 operands through the decoder's rings
    if (p.ncls > 256 || p.rows <= 0) return hipErrorInvalidValue;
    if (p.partials && (p.nsplit < 1 || p.nsplit > 16 || !p.ffn_b2 || !p.ln3_gamma || !p.ln3_beta)) return hipErrorInvalidValue;
    OPD_LAUNCH(heads_kernel, dim3((p.rows + HEAD_ROWS - 1) / HEAD_ROWS), dim3(512), 0, stream, p);
    return hipGetLastError();
}
#endif

#ifndef OPD_ELEM_BF16   // (no 16-bit operands: defined once)
hipError_t opd_launch_postprocess(const PostParams& p, hipStream_t stream) {
    if (p.Q > 128 || p.ncls > 128 || p.B <= 0) return hipErrorInvalidValue;
    OPD_LAUNCH(postprocess_kernel, dim3(p.B), dim3(1024), 0, stream, p);
    return hipGetLastError();
}
#endif

// Tracker cost matrix (src/tracking/similarity.py:42-220): one thread per (track i, detection j).
//   cos = clip(dot_fp32(f1[i], f2[j]), -1, 1)  if both carry features        (appearance term, weight aw)
//   iou of the xywh boxes, 0 when the intersection is empty or the union is not positive, clipped to [0, 1]   (weight mw)
//   similarity = clip((aw*cos + mw*iou) / (weights used), 0, 1) in double like the reference's Python floats, stored as fp32;
//   as_distance: 1 - similarity (computed on the fp32 value, like `1.0 - similarity_matrix`).
#ifndef OPD_ELEM_BF16
__global__ void similarity_matrix_kernel(const float* __restrict__ f1, const float* __restrict__ b1, const uint8_t* __restrict__ has1,
                                         int n1, const float* __restrict__ f2, const float* __restrict__ b2,
                                         const uint8_t* __restrict__ has2, int n2, int D, double aw, double mw, int as_distance,
                                         float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n1 * n2) return;
    const int i = idx / n2, j = idx - i * n2;
    double score = 0.0, total = 0.0;
    if (f1 && f2 && (!has1 || has1[i]) && (!has2 || has2[j])) {
        const float* a = f1 + (size_t)i * D;
        const float* c = f2 + (size_t)j * D;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 4 <= D; k += 4) {
            acc[0] = fmaf(a[k], c[k], acc[0]); acc[1] = fmaf(a[k + 1], c[k + 1], acc[1]);
            acc[2] = fmaf(a[k + 2], c[k + 2], acc[2]); acc[3] = fmaf(a[k + 3], c[k + 3], acc[3]);
        }
        for (; k < D; ++k) acc[0] = fmaf(a[k], c[k], acc[0]);
        float dot = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        dot = dot < -1.f ? -1.f : (dot > 1.f ? 1.f : dot);
        score += aw * (double)dot;
        total += aw;
    }
    {
        const double x1 = b1[4 * i], y1 = b1[4 * i + 1], w1 = b1[4 * i + 2], h1 = b1[4 * i + 3];
        const double x2 = b2[4 * j], y2 = b2[4 * j + 1], w2 = b2[4 * j + 2], h2 = b2[4 * j + 3];
        const double ix0 = x1 > x2 ? x1 : x2, iy0 = y1 > y2 ? y1 : y2;
        const double ix1 = (x1 + w1) < (x2 + w2) ? (x1 + w1) : (x2 + w2), iy1 = (y1 + h1) < (y2 + h2) ? (y1 + h1) : (y2 + h2);
        double iou = 0.0;
        if (ix1 > ix0 && iy1 > iy0) {
            const double inter = (ix1 - ix0) * (iy1 - iy0);
            const double uni = w1 * h1 + w2 * h2 - inter;
            if (uni > 0.0) {
                iou = inter / uni;
                iou = iou < 0.0 ? 0.0 : (iou > 1.0 ? 1.0 : iou);
            }
        }
        score += mw * iou;
        total += mw;
    }
    double sim = total > 0.0 ? score / total : 0.0;
    sim = sim < 0.0 ? 0.0 : (sim > 1.0 ? 1.0 : sim);
    const float simf = (float)sim;
    out[idx] = as_distance ? 1.0f - simf : simf;
}

#endif
#ifndef OPD_ELEM_BF16   // (no 16-bit operands: defined once)
hipError_t opd_launch_similarity_matrix(const float* f1, const float* b1, const uint8_t* has1, int n1, const float* f2, const float* b2,
                                        const uint8_t* has2, int n2, int D, double aw, double mw, int as_distance, float* out,
                                        hipStream_t stream) {
    if (n1 <= 0 || n2 <= 0 || D <= 0 || !b1 || !b2 || !out) return hipErrorInvalidValue;
    OPD_LAUNCH(similarity_matrix_kernel, dim3((n1 * n2 + 127) / 128), dim3(128), 0, stream, f1, b1, has1, n1, f2, b2, has2, n2, D, aw,
                       mw, as_distance, out);
    return hipGetLastError();
}
#endif

#ifndef OPD_ELEM_BF16   // (no 16-bit operands: defined once)
hipError_t opd_launch_roi_features(const float* enc, const int32_t* rois, float* out, int n, int h, int w,
                                   hipStream_t stream) {
    if (n <= 0) return hipErrorInvalidValue;
    OPD_LAUNCH(roi_features_kernel, dim3(n), dim3(1024), 0, stream, enc, rois, out, h, w);
    return hipGetLastError();
}
hipError_t opd_launch_roi_features_records(const float* enc, const void* records, const int32_t* counts, const int32_t* orig_hw, int label,
                                           float* out, int B, int Q, int h, int w, hipStream_t stream) {
    if (B <= 0 || Q <= 0 || h <= 0 || w <= 0 || !enc || !records || !counts || !orig_hw || !out) return hipErrorInvalidValue;
    OPD_LAUNCH(roi_features_rec_kernel, dim3(Q, B), dim3(1024), 0, stream, enc, reinterpret_cast<const DetRec*>(records), counts, orig_hw, label, out, Q, h, w);
    return hipGetLastError();
}
#endif
