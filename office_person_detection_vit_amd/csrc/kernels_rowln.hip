// kernels_rowln.hip — Linear(K -> 256) + bias + residual + LayerNorm in ONE kernel (gfx950, fp16 MFMA, fp32 statistics).
//
// the attention output projections of the transformer layers,
//     y = LayerNorm(x . W^T + b + residual)            (HF:models/detr/modeling_detr.py:646-660, 734-775: post-LN layers)
// The output width IS the model width (256), so a workgroup that owns whole rows can finish the LayerNorm itself: no fp32
// pre-norm tensor, no split-K slabs, no second launch.  At batch 8 the unfused pair cost 11 + 8 us per encoder layer
// (M = 8400) and 6 + 6 us per decoder projection (M = 800, split-K 4 + reduce); this kernel is one launch of ~6 us.
//
// Tile: 32 rows x 256 columns x 64 (k) per workgroup of 4 waves; wave w owns columns 64w..64w+63 for all 32 rows, as
// 4 x 2 accumulator tiles of the 16x16x32 MFMA (weights = A operand, rows of x = B operand, as in kernels_gemm.hip).
// Staging is the LDS-DMA scheme of conv_gemm_dma_kernel (1-KiB pieces, XOR swizzle on the source side, two stage
// buffers); every wave stages the 8 weight pieces of its own columns plus one activation piece.  LayerNorm statistics are
// two-pass (mean, then centred variance) in fp32 like layernorm256_kernel: lane partials -> 4 lane groups (shuffles) ->
// 4 waves (LDS).
//
// Also in this file, same row-owner idea: gemm_ln256_os_kernel (K = 256 in one staging batch), gemm_ln256_ring_kernel (deep K through a
// three-stage ring: the input projection), enc_ffn_kernel (round 4: everything behind an encoder layer's attention -- output projection +
// LayerNorm + fc1 + ReLU + fc2 + LayerNorm -- in one launch, weights as per-wave fragment streams through wave-private rings) and
// gemm_k256_kernel (small-M linears of the unfused decoder chain).
#include <hip/hip_runtime.h>
#include "opd_kernels.h"
#include "opd_elem.h"

typedef elem_t half8 __attribute__((ext_vector_type(8)));
typedef elem_t half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

namespace {

constexpr int ROW_BYTES = 128;
constexpr int TM = 32;                       // rows per workgroup
constexpr int A_BYTES = TM * ROW_BYTES;      // 4 KiB
constexpr int W_BYTES = 256 * ROW_BYTES;     // 32 KiB
constexpr int STAGE_BYTES = A_BYTES + W_BYTES;

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROW_BYTES + ((chunk ^ (row & 7)) << 4); }

__global__ __launch_bounds__(256, 2) void gemm_ln256_kernel(GemmLnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float red[2][4][TM];          // [pass][wave][row]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int m_base = blockIdx.x * TM;
    const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;

    // rows >= M fall outside the activation descriptor: zero fill
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.x), 0, (unsigned)((size_t)p.M * p.K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.w), 0, (unsigned)((size_t)256 * p.K * 2), 0x00020000);
    const unsigned xoff = (unsigned)((m_base + wave * 8 + lrow) * p.K) * 2u + (unsigned)lchunk * 16u;   // activation piece `wave`
    const unsigned woff = (unsigned)((wave * 64 + lrow) * p.K) * 2u + (unsigned)lchunk * 16u;           // weight rows 64w + 8i + lrow
    const unsigned wstep = (unsigned)(8 * p.K) * 2u;
    auto issue = [&](int ks, int buf) {
        unsigned char* As = smem + buf * STAGE_BYTES;
        unsigned char* Ws = As + A_BYTES;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(As + wave * 1024), 16, xoff, ks * 128, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(Ws + (wave * 8 + i) * 1024), 16,
                                                     woff + (unsigned)i * wstep, ks * 128, 0, 0);
    };

    issue(0, 0);
    float4v acc[4][2];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const float4v b = *reinterpret_cast<const float4v*>(p.bias + wave * 64 + nt * 16 + g * 4);
        acc[nt][0] = b;
        acc[nt][1] = b;
    }
    OPD_DMA_BARRIER();   // (the bias loads above are younger than the first stage's requests)
    const int nk = p.K / 64;
    for (int ks = 0; ks < nk; ++ks) {
        if (ks + 1 < nk) issue(ks + 1, (ks + 1) & 1);
        const unsigned char* As = smem + (ks & 1) * STAGE_BYTES;
        const unsigned char* Ws = As + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 xf[2], wf[4];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) xf[mt] = *reinterpret_cast<const half8*>(As + swz(mt * 16 + li, kk * 4 + g));
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wf[nt] = *reinterpret_cast<const half8*>(Ws + swz(wave * 64 + nt * 16 + li, kk * 4 + g));
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = OPD_MFMA_16x16x32(wf[nt], xf[mt], acc[nt][mt]);
        }
        __syncthreads();
    }

    // ---- + residual, LayerNorm over the 256 columns of each row -----------------------------------------------------------
    // lane (g, li) holds, for row mt*16 + li, the 16 columns 64*wave + nt*16 + 4g + r
    float sum[2] = {0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = m_base + mt * 16 + li;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if (p.res32 && m < p.M) acc[nt][mt] += *reinterpret_cast<const float4v*>(p.res32 + (size_t)m * 256 + wave * 64 + nt * 16 + g * 4);
            sum[mt] += acc[nt][mt][0] + acc[nt][mt][1] + acc[nt][mt][2] + acc[nt][mt][3];
        }
        sum[mt] += __shfl_xor(sum[mt], 16);
        sum[mt] += __shfl_xor(sum[mt], 32);
        if (g == 0) red[0][wave][mt * 16 + li] = sum[mt];
    }
    __syncthreads();
    float mean[2], sq[2] = {0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int r = mt * 16 + li;
        mean[mt] = (red[0][0][r] + red[0][1][r] + red[0][2][r] + red[0][3][r]) * (1.0f / 256.0f);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            acc[nt][mt] -= mean[mt];
            sq[mt] += acc[nt][mt][0] * acc[nt][mt][0] + acc[nt][mt][1] * acc[nt][mt][1] + acc[nt][mt][2] * acc[nt][mt][2] +
                      acc[nt][mt][3] * acc[nt][mt][3];
        }
        sq[mt] += __shfl_xor(sq[mt], 16);
        sq[mt] += __shfl_xor(sq[mt], 32);
        if (g == 0) red[1][wave][r] = sq[mt];
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int r = mt * 16 + li;
        const int m = m_base + r;
        const float var = (red[1][0][r] + red[1][1][r] + red[1][2][r] + red[1][3][r]) * (1.0f / 256.0f);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int c = wave * 64 + nt * 16 + g * 4;
            const float4v gm = *reinterpret_cast<const float4v*>(p.gamma + c);
            const float4v bt = *reinterpret_cast<const float4v*>(p.beta + c);
            float4v o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = acc[nt][mt][q] * rstd * gm[q] + bt[q];
            if (m < p.M) {
                if (p.y32) *reinterpret_cast<float4v*>(p.y32 + (size_t)m * 256 + c) = o;
                if (p.y16) {
                    half4 h;
                    h[0] = (elem_t)o[0]; h[1] = (elem_t)o[1]; h[2] = (elem_t)o[2]; h[3] = (elem_t)o[3];
                    *reinterpret_cast<half4*>(p.y16 + (size_t)m * 256 + c) = h;
                }
            }
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// One-shot form for K = 256 (every attention output projection of the model).  The k-loop above is four dependent staging
// round trips per workgroup (14.6 us at M = 8400, 10 us at M = 800 for 0.1-1 GFLOP); here a workgroup of 48 rows stages the
// WHOLE weight matrix (256 x 256 fp16 = 128 KiB) and its 48 activation rows (24 KiB) in one batch of LDS-DMA, waits once and
// runs its 96 MFMAs per wave back to back, then the same two-pass LayerNorm.  152 KiB of LDS, one workgroup per CU;
// M = 8400 gives 175 workgroups (one round), M = 800 gives 17.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int OS_TM = 48;
constexpr int OS_XSUB = OS_TM * ROW_BYTES;   // one [48 rows][64 k] activation sub-tile: 6 KiB
constexpr int OS_WSUB = 256 * ROW_BYTES;     // one [256 rows][64 k] weight sub-tile: 32 KiB
constexpr int OS_LDS = 4 * (OS_XSUB + OS_WSUB);

constexpr int OS_NW = 8;   // waves per workgroup: 32 columns each; twice the LDS-DMA instructions in flight of a 4-wave workgroup

__global__ __launch_bounds__(64 * OS_NW, 1) void gemm_ln256_os_kernel(GemmLnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float red[2][OS_NW][OS_TM];   // [pass][wave][row]
    unsigned char* const Xs = smem;                   // 4 sub-tiles
    unsigned char* const Ws = smem + 4 * OS_XSUB;     // 4 sub-tiles
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int m_base = blockIdx.x * OS_TM;
    const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;
    constexpr int NT = 256 / 16 / OS_NW;     // column tiles per wave (2)
    constexpr int WP = 256 / 8 / OS_NW;      // weight pieces per wave and sub-tile (4)

    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.x), 0, (unsigned)((size_t)p.M * 256 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.w), 0, (unsigned)(256 * 256 * 2), 0x00020000);
    // weights: wave w stages its own 32 rows (4 pieces) of each of the 4 sub-tiles; activations: 6 pieces per sub-tile, 24 in all,
    // 3 per wave (piece q = 3 wave + i -> sub-tile q / 6, rows 8 (q % 6) ..); rows >= M are outside the descriptor: zero fill
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int i = 0; i < WP; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(Ws + t * OS_WSUB + (wave * WP + i) * 1024), 16,
                                                     (unsigned)((wave * (8 * WP) + i * 8 + lrow) * 256) * 2u + (unsigned)lchunk * 16u, t * 128, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int q = wave * 3 + i, t = q / 6, r = (q % 6) * 8 + lrow;
        const int m = m_base + r;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(Xs + t * OS_XSUB + (q % 6) * 1024), 16,
                                                 m < p.M ? (unsigned)(m * 256) * 2u + (unsigned)lchunk * 16u : 0x80000000u, t * 128, 0, 0);
    }
    float4v acc[NT][3];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float4v b = *reinterpret_cast<const float4v*>(p.bias + (wave * NT + nt) * 16 + g * 4);
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) acc[nt][mt] = b;
    }
    // residual rows in flight during the MFMAs: lane (g, li) holds, for row mt*16 + li, columns 32 wave + nt*16 + 4g + r
    float4v res[NT][3];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
        const int m = m_base + mt * 16 + li;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            res[nt][mt] = float4v{0.f, 0.f, 0.f, 0.f};
            if (p.res32 && m < p.M) res[nt][mt] = *reinterpret_cast<const float4v*>(p.res32 + (size_t)m * 256 + (wave * NT + nt) * 16 + g * 4);
        }
    }
    OPD_DMA_BARRIER();   // drain + barrier: everything is in LDS (the bias / residual loads above are younger than the requests)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 xf[3], wf[NT];
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) xf[mt] = *reinterpret_cast<const half8*>(Xs + t * OS_XSUB + swz(mt * 16 + li, kk * 4 + g));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[nt] = *reinterpret_cast<const half8*>(Ws + t * OS_WSUB + swz((wave * NT + nt) * 16 + li, kk * 4 + g));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) acc[nt][mt] = OPD_MFMA_16x16x32(wf[nt], xf[mt], acc[nt][mt]);
        }
    // ---- + residual, LayerNorm over the 256 columns of each row (as in gemm_ln256_kernel) -----------------------------------
    float sum[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            acc[nt][mt] += res[nt][mt];
            sum[mt] += acc[nt][mt][0] + acc[nt][mt][1] + acc[nt][mt][2] + acc[nt][mt][3];
        }
        sum[mt] += __shfl_xor(sum[mt], 16);
        sum[mt] += __shfl_xor(sum[mt], 32);
        if (g == 0) red[0][wave][mt * 16 + li] = sum[mt];
    }
    __syncthreads();
    float sq[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
        const int r = mt * 16 + li;
        float mean = 0.f;
#pragma unroll
        for (int w = 0; w < OS_NW; ++w) mean += red[0][w][r];
        mean *= (1.0f / 256.0f);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            acc[nt][mt] -= mean;
            sq[mt] += acc[nt][mt][0] * acc[nt][mt][0] + acc[nt][mt][1] * acc[nt][mt][1] + acc[nt][mt][2] * acc[nt][mt][2] +
                      acc[nt][mt][3] * acc[nt][mt][3];
        }
        sq[mt] += __shfl_xor(sq[mt], 16);
        sq[mt] += __shfl_xor(sq[mt], 32);
        if (g == 0) red[1][wave][r] = sq[mt];
    }
    __syncthreads();
    float4v gm[NT], bt[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        gm[nt] = *reinterpret_cast<const float4v*>(p.gamma + (wave * NT + nt) * 16 + g * 4);
        bt[nt] = *reinterpret_cast<const float4v*>(p.beta + (wave * NT + nt) * 16 + g * 4);
    }
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
        const int r = mt * 16 + li;
        const int m = m_base + r;
        float var = 0.f;
#pragma unroll
        for (int w = 0; w < OS_NW; ++w) var += red[1][w][r];
        var *= (1.0f / 256.0f);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c = (wave * NT + nt) * 16 + g * 4;
            float4v o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = acc[nt][mt][q] * rstd * gm[nt][q] + bt[nt][q];
            if (m < p.M) {
                if (p.y32) *reinterpret_cast<float4v*>(p.y32 + (size_t)m * 256 + c) = o;
                if (p.y16) {
                    half4 h;
                    h[0] = (elem_t)o[0]; h[1] = (elem_t)o[1]; h[2] = (elem_t)o[2]; h[3] = (elem_t)o[3];
                    *reinterpret_cast<half4*>(p.y16 + (size_t)m * 256 + c) = h;
                }
            }
        }
    }
#endif
}

// Rows of 256 columns held by EIGHT waves in accumulator layout (lane (g, li) of wave w: rows mt * 16 + li, columns (2 w + nt) * 16 + 4 g + r):
// optional LayerNorm (two-pass fp32 statistics: lanes -> 4 lane groups by shuffles -> 8 waves through `red`), then y32 / y16 / the position
// shadow yp16 = fp16(y + pos).  Shared by the deep-K ring kernel and the fused encoder FFN.
struct RowLnOut {
    const float* gamma; const float* beta;
    float* y32; f16_t* y16; f16_t* yp16;
    const float* pos; const float* const* pos_ptrs; int pos_period;
    int M;
    // optional (fused encoder FFN with a tail projection): fp16(y) and fp16(y + pos) also as [rows][256] B-operand images in LDS, 16-byte chunks
    // XOR-swizzled by (row & 15)
    unsigned char* img_x = nullptr;
    unsigned char* img_xp = nullptr;
};
template <bool LN, int MT = 4>
__device__ __forceinline__ void row_ln_store(const RowLnOut& p, float4v (&acc)[2][MT], float (&red)[2][8][16 * MT], const int m_base, const int wave, const int g,
                                             const int li) {
    constexpr int NT = 2, RG_NW = 8;
    float sum[MT] = {};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) sum[mt] += acc[nt][mt][0] + acc[nt][mt][1] + acc[nt][mt][2] + acc[nt][mt][3];
        if constexpr (LN) {
            sum[mt] += __shfl_xor(sum[mt], 16);
            sum[mt] += __shfl_xor(sum[mt], 32);
            if (g == 0) red[0][wave][mt * 16 + li] = sum[mt];
        }
    }
    if constexpr (LN) __syncthreads();
    float sq[MT] = {};
#pragma unroll
    for (int mt = 0; LN && mt < MT; ++mt) {
        const int r = mt * 16 + li;
        float mean = 0.f;
#pragma unroll
        for (int w = 0; w < RG_NW; ++w) mean += red[0][w][r];
        mean *= (1.0f / 256.0f);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            acc[nt][mt] -= mean;
            sq[mt] += acc[nt][mt][0] * acc[nt][mt][0] + acc[nt][mt][1] * acc[nt][mt][1] + acc[nt][mt][2] * acc[nt][mt][2] +
                      acc[nt][mt][3] * acc[nt][mt][3];
        }
        sq[mt] += __shfl_xor(sq[mt], 16);
        sq[mt] += __shfl_xor(sq[mt], 32);
        if (g == 0) red[1][wave][r] = sq[mt];
    }
    if constexpr (LN) __syncthreads();
    float4v gm[NT], bt[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        gm[nt] = LN ? *reinterpret_cast<const float4v*>(p.gamma + (wave * NT + nt) * 16 + g * 4) : float4v{1.f, 1.f, 1.f, 1.f};
        bt[nt] = LN ? *reinterpret_cast<const float4v*>(p.beta + (wave * NT + nt) * 16 + g * 4) : float4v{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int r = mt * 16 + li;
        const int m = m_base + r;
        float var = 0.f;
#pragma unroll
        for (int w = 0; LN && w < RG_NW; ++w) var += red[1][w][r];
        var *= (1.0f / 256.0f);
        const float rstd = LN ? 1.0f / sqrtf(var + 1e-5f) : 1.0f;
        const float* pos = nullptr;   // second fp16 shadow y + position embedding (encoder: the next layer's q / k projection input)
        if ((p.yp16 || p.img_xp) && p.pos_period > 0 && m < p.M) {
            const int fr = m / p.pos_period, t = m - fr * p.pos_period;
            pos = (p.pos_ptrs ? p.pos_ptrs[fr] : p.pos) + (size_t)t * 256;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c = (wave * NT + nt) * 16 + g * 4;
            float4v o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = LN ? acc[nt][mt][q] * rstd * gm[nt][q] + bt[nt][q] : acc[nt][mt][q];
            if (m < p.M) {
                if (p.y32) *reinterpret_cast<float4v*>(p.y32 + (size_t)m * 256 + c) = o;
                if (p.y16) {
                    half4 h;
                    h[0] = (elem_t)o[0]; h[1] = (elem_t)o[1]; h[2] = (elem_t)o[2]; h[3] = (elem_t)o[3];
                    *reinterpret_cast<half4*>(p.y16 + (size_t)m * 256 + c) = h;
                }
                if (pos && p.yp16) {
                    const float4v pe = *reinterpret_cast<const float4v*>(pos + c);
                    half4 h;
#pragma unroll
                    for (int q = 0; q < 4; ++q) h[q] = (elem_t)(o[q] + pe[q]);
                    *reinterpret_cast<half4*>(p.yp16 + (size_t)m * 256 + c) = h;
                }
            }
            if (p.img_x) {   // (rows past M: finite values nobody stores)
                const int ioff = r * 512 + ((((c >> 3) ^ li) << 4) | ((c & 4) << 1));
                half4 h, hp;
                float4v pe = float4v{0.f, 0.f, 0.f, 0.f};
                if (pos) pe = *reinterpret_cast<const float4v*>(pos + c);
#pragma unroll
                for (int q = 0; q < 4; ++q) { h[q] = (elem_t)o[q]; hp[q] = (elem_t)(o[q] + pe[q]); }
                *reinterpret_cast<half4*>(p.img_x + ioff) = h;
                *reinterpret_cast<half4*>(p.img_xp + ioff) = hp;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Deep-K form (encoder FFN-2: K = 2048): row owners again — a workgroup of 64 rows walks the WHOLE reduction and finishes the
// LayerNorm itself, so the split-K fp32 slabs (34 MB written and re-read per encoder layer at batch 8) and the reduce launch
// disappear.  What such a workgroup is bound by is the weight stream: all 256 x K weights (1 MiB) pass through its CU's LDS-DMA
// path, which moves 21 / 33 / 42 bytes per clock with 32 / 64 / 128 KiB in flight (tools/microbench/ldpath.hip).  Hence a
// THREE-stage ring of [64 x 64] activation + [256 x 64] weight sub-tiles (40 KiB per stage, two k-steps = 80 KiB in flight
// while the third is being multiplied), counted vmcnt waits, one barrier per k-step.  The per-workgroup time does not depend on
// the number of rows (24-32 MFMAs per wave and k-step against ~1100 clocks of staging), so the tile is as tall as the registers
// allow: M = 8400 gives 132 workgroups, which leaves half the CUs to whatever else is in flight.  Eight waves (two per SIMD, 32
// columns each) rather than four: more LDS-DMA instructions in flight per CU (31.0 -> 26.7 us at M = 8400; the split-K launch
// plus its reduce launch take 19.4 + 12.2 us, and move 514 MB of fp32 slabs per forward that this kernel does not).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int RG_TM = 64;
constexpr int RG_NS = 3;
constexpr int RG_NW = 8;                        // waves per workgroup: two per SIMD, each owning 32 of the 256 columns
constexpr int RG_XSUB = RG_TM * ROW_BYTES;      // 8 KiB
constexpr int RG_STAGE = RG_XSUB + W_BYTES;     // 40 KiB
constexpr int RG_LDS = RG_NS * RG_STAGE;

// LN = false (gamma == null: the input projection): no normalisation, y = x . W^T + b (+ res).
template <bool LN>
__global__ __launch_bounds__(64 * RG_NW, 1) void gemm_ln256_ring_kernel(GemmLnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float red[2][8][64];              // [pass][wave][row]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int m_base = blockIdx.x * RG_TM;
    const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;
    constexpr int NT = 256 / 16 / RG_NW;         // column tiles per wave (2)
    constexpr int WP = 256 / 8 / RG_NW;          // weight pieces per wave and k-step (4)
    constexpr int PIECES = WP + 1;               // + one piece of the activation tile

    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.x), 0, (unsigned)((size_t)p.M * p.K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.w), 0, (unsigned)((size_t)256 * p.K * 2), 0x00020000);
    // per k-step every wave issues exactly PIECES pieces: activation rows 8 wave + lrow (rows >= M fall outside the descriptor: zero
    // fill) and the WP pieces of its own 32 weight rows
    const unsigned xoff = (unsigned)((m_base + wave * 8 + lrow) * p.K) * 2u + (unsigned)lchunk * 16u;
    const unsigned woff = (unsigned)((wave * (8 * WP) + lrow) * p.K) * 2u + (unsigned)lchunk * 16u;
    const unsigned wstep = (unsigned)(8 * p.K) * 2u;
    auto issue = [&](int ks) {
        unsigned char* Xs = smem + (ks % RG_NS) * RG_STAGE;
        unsigned char* Ws = Xs + RG_XSUB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(Xs + wave * 1024), 16, xoff, ks * 128, 0, 0);
#pragma unroll
        for (int i = 0; i < WP; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(Ws + (wave * WP + i) * 1024), 16,
                                                     woff + (unsigned)i * wstep, ks * 128, 0, 0);
    };
    const int nk = p.K / 64;
    issue(0);
    if (nk > 1) issue(1);
    float4v acc[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float4v b = *reinterpret_cast<const float4v*>(p.bias + (wave * NT + nt) * 16 + g * 4);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = b;
    }
    float4v res[NT][4];
#pragma unroll 1
    for (int ks = 0; ks < nk; ++ks) {
        // k-step ks has landed once at most the pieces of k-step ks + 1 are outstanding (LDS-DMA requests retire in issue order among themselves;
        // nothing else is in flight here: the residual rows are requested behind the last wait)
        if (ks + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();       // everybody's pieces of ks are in LDS; everybody has finished multiplying stage (ks - 1) % 3
        if (ks + 2 < nk) issue(ks + 2);     // ... which is the stage this goes to
        // (a second barrier per k-step, so that k-step ks + 2 is issued BEFORE the wait for ks and two whole stages stay in flight, measured
        //  slower: 30.0 against 26.7 us.  Round 4 tried more bytes in flight outright: FOUR 64-wide stages (all 160 KiB of LDS, LayerNorm
        //  scratch over stage 0): 27.2 us, no change; SEVEN 32-wide stages (120 KiB in flight): 40.0 us.  A one-shot LDS-DMA of 256 KiB per
        //  workgroup lands 0.7 us after the launch floor (tools/microbench/oneshot.hip), so neither the path's rate nor the bytes in flight
        //  bound this loop: a k-step costs ~0.8 us of wait + barrier + fragment-read latency + 96 KiB of LDS fragment reads per workgroup
        //  (every wave reads the whole 64 x 64 activation tile) whatever its width, and 64 narrow steps cost more than 32 wide ones.)
        if (ks + 1 == nk) {                 // last k-step: the residual rows travel during its MFMAs
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int m = m_base + mt * 16 + li;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    res[nt][mt] = float4v{0.f, 0.f, 0.f, 0.f};
                    if (p.res32 && m < p.M) res[nt][mt] = *reinterpret_cast<const float4v*>(p.res32 + (size_t)m * 256 + (wave * NT + nt) * 16 + g * 4);
                }
            }
        }
        const unsigned char* Xs = smem + (ks % RG_NS) * RG_STAGE;
        const unsigned char* Ws = Xs + RG_XSUB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 xf[4], wf[NT];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) xf[mt] = *reinterpret_cast<const half8*>(Xs + swz(mt * 16 + li, kk * 4 + g));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[nt] = *reinterpret_cast<const half8*>(Ws + swz((wave * NT + nt) * 16 + li, kk * 4 + g));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = OPD_MFMA_16x16x32(wf[nt], xf[mt], acc[nt][mt]);
        }
    }
    // ---- + residual, LayerNorm over the 256 columns of each row -------------------------------------------------------------
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt][mt] += res[nt][mt];
    const RowLnOut o{p.gamma, p.beta, p.y32, p.y16, p.yp16, p.pos, p.pos_ptrs, p.pos_period, p.M};
    row_ln_store<LN>(o, acc, red, m_base, wave, g, li);
#endif
}


// ---------------------------------------------------------------------------------------------------------------------
// The encoder's feed-forward block in ONE launch:  y = LayerNorm(x32 + relu(x . W1^T + b1) . W2^T + b2)  (+ the fp16 copies)
// HF:models/detr/modeling_detr.py:646-660 (DetrEncoderLayer: fc1 -> activation -> fc2 -> residual -> final_layer_norm).
//
// The two-launch form (fc1 through conv_gemm_dma_kernel, fc2 through the ring kernel above) writes the [M][2048] hidden tensor (34 MB at
// batch 8) and reads it back, and each launch pulls its whole weight matrix through every workgroup's k-loop with a workgroup barrier per
// k-step (the ring kernel: ~0.8 us per k-step whatever its width).  Here a workgroup owns 48 rows for BOTH layers (M = 8400: 175 workgroups,
// one round): the hidden activations of one 128-wide chunk live in LDS between the two GEMMs and never reach HBM, the slab's own 24 x-fragments
// per lane stay in REGISTERS for all 16 chunks, and the weights travel as in the decoder kernels (kernels_dec.hip): pre-arranged at load in
// MFMA-fragment order, one contiguous stream per wave (`opd_encffn_pack`), through a WAVE-PRIVATE LDS-DMA ring with counted vmcnt waits -- no
// barrier in either k-loop, ONE barrier per chunk.
//   fc1: wave w owns hidden columns 16 w .. 16 w + 15 of the chunk for the slab's three 16-row tiles (accumulators start from the bias, which
//        travels in the stream as a piece in accumulator layout); ReLU, fp16, into H[chunk & 1];
//   fc2: wave w owns output columns 32 w .. 32 w + 31 (two tiles) for the three row tiles; 4 k-steps per chunk.
// A wave's work is cut into GROUPS of 4-5 pieces (G0: bias + W1 k-steps 0-3, G1: W1 k-steps 4-7, G2 / G3: W2 k-steps 0-1 / 2-3 of both tiles).
// The first form ran wait -> fragment reads -> MFMAs -> re-request per group and took the SUM of the three resources' times (41 us: LDS reads,
// MFMAs and LDS-DMA issue are ~14 us each); this form is skewed and software pipelined:
//   * fc1 runs one chunk AHEAD of fc2 (H is double buffered), so the barrier that publishes a hidden chunk is a whole iteration away from
//     the reads that need it;
//   * the fragment reads of group k + 1 (into a second register set) are issued BEFORE the MFMAs of group k and its slots are re-requested
//     right after, so LDS latency and LDS-DMA issue of one wave overlap its own MFMAs and those of the SIMD's other wave.
// Stream order = consumption order: G0(0) G1(0) G0(1) { G1(c+1) G2(c) G3(c) G0(c+2) } for c = 0 .. F/128 - 1, with zero blocks for the fc1
// chunks past the end (one hidden chunk is computed in vain; every wait count is the same in every iteration).
// LDS: x slab [48][256] (24 KiB, read once) and H [2][48][128] (24 KiB), rows XOR-swizzled by (row & 15) in 16-byte chunks so that the four
// lane groups of a ds_read_b128 fragment read cover all 64 banks; 8 x 13 KiB of rings: 152 KiB + 3 KiB of LayerNorm scratch.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int EF_MT = 3;
constexpr int EF_TM = 16 * EF_MT;
constexpr int EF_R = 13;                          // ring slots (1 KiB) per wave
constexpr int EF_X = EF_TM * 512;
constexpr int EF_H = EF_TM * 256;
constexpr int EF_RING0 = EF_X + 2 * EF_H;
constexpr int EF_LDS = EF_RING0 + 8 * EF_R * 1024;
constexpr int EF_PIECES(const int nch, const int tail, const int front) { return (front ? 16 : 0) + 14 + 17 * nch + (tail ? 17 * tail + 5 : 0); }   // per wave

template <int N>
__device__ __forceinline__ void ef_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#define EF_FENCE() __builtin_amdgcn_sched_barrier(0)
#define EF_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

struct EfFrags {
    float4v b, b2;
    half8 w[4];
    half8 h[2][EF_MT];
};

template <int DBG>   // DBG: timing ablations of tools/bench_enc_ffn.py (compile-time: 0 in the model): 1 no MFMAs, 2 no re-requests, 4 no hidden-chunk traffic, 8 no barriers
__global__ __launch_bounds__(512, 1) void enc_ffn_kernel(EncFfnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    __shared__ float red[2][8][EF_TM];
    constexpr int MT = EF_MT;
    unsigned char* const X = smem;
    unsigned char* const H = smem + EF_X;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15, lane16 = lane * 16;
    const int m_base = blockIdx.x * EF_TM;
    unsigned char* const ring = smem + EF_RING0 + wave * (EF_R * 1024);
    const int nch = p.F / 128;
    const unsigned stride = (unsigned)EF_PIECES(nch, p.pack_tail, p.pack_front) * 1024u;   // bytes of a wave's stream
    const unsigned skip = (p.pack_front && !p.attn) ? 16u * 1024u : 0u;                    // (a stream packed with the front projection, run without it)
    const unsigned total = stride - skip;
    const unsigned char* const wsrc = p.wpack + (size_t)wave * stride + skip + lane16;
    auto dma = [&](const unsigned char* src_lane, unsigned char* slot) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_lane, (__attribute__((address_space(3))) void*)slot, 16, 0, 0);
    };
    // ---- the slab: rows 2 MT w .. + 2 MT - 1 by this wave (1 KiB = two rows; lane -> row, position; the position holds chunk position ^ (row & 15))
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int row = 2 * MT * wave + 2 * i + (lane >> 5);
        const int c16 = (lane & 31) ^ (row & 15);
        const int grow = m_base + row < p.M ? m_base + row : p.M - 1;     // (rows past the end: a valid address, never stored)
        dma(reinterpret_cast<const unsigned char*>(p.attn ? p.attn : p.x) + (size_t)grow * 512 + c16 * 16, X + (2 * MT * wave + 2 * i) * 512);
    }
#pragma unroll
    for (int n = 0; n < EF_R; ++n) dma(wsrc + n * 1024, ring + n * 1024);
    unsigned noff = EF_R * 1024u;      // stream offset of the next piece to request
    // Cooperative L2 warm-up of the weight stream (round 5).  Inside the forward the layer's 2.4 MB of weights are in nobody's L2 when the launch
    // starts, and every workgroup of an XCD walks the SAME stream from the same end: the leader takes every miss (13 KiB in flight per wave
    // against a ~2-us round trip), the others follow in its wake -- 53 us in the graph against 35-39 us back to back on warm caches.  So each
    // wave first TOUCHES a 1 / n-th share of its stream, n = the workgroups that share its XCD (blockIdx mod 8: placement is speed only), one
    // dword per 128-byte line through an LDS-DMA request whose 256 bytes land in the hidden-chunk buffer long before that is first written
    // (requests return in order, and these are older than every piece re-requested below).  The ring's counted waits stay conservative: they
    // assume fewer requests in flight than there are.
    if (p.wprefetch) {
        const unsigned nx = (gridDim.x + 7u) >> 3, xi = blockIdx.x >> 3;
        const unsigned seg = ((total + nx - 1u) / nx + 127u) & ~127u;
        const unsigned lo = xi * seg, hi = lo + seg < total ? lo + seg : total;
        const unsigned char* const wbase = p.wpack + (size_t)wave * stride + skip;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const unsigned off = lo + (unsigned)(q * 64 + lane) * 128u;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase + (off < hi ? off : total - 128u)),
                                             (__attribute__((address_space(3))) void*)(H + wave * 512 + q * 256), 4, 0, 0);
        }
    }
    unsigned slot = 0u;                // ring offset of the next piece to consume
    auto take = [&]() { const unsigned s_ = slot; slot = slot + 1024u == EF_R * 1024u ? 0u : slot + 1024u; return s_; };
    // (past the end of the stream the last piece is requested again: the wait counts stay the same in every iteration, nobody reads the slot)
    constexpr int dbg = DBG;
    auto reissue = [&](const unsigned s_) { if constexpr (!(dbg & 2)) dma(wsrc + (noff < total ? noff : total - 1024u), ring + s_); noff += 1024u; };
    auto frag = [&](const unsigned s_) { return *reinterpret_cast<const half8*>(ring + s_ + lane16); };
    float4v acc1[MT], acc2[2][MT];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc2[j][mt] = float4v{0.f, 0.f, 0.f, 0.f};
    ef_wait_vm<EF_R>();                                       // the slab pieces are older than the ring's
    __builtin_amdgcn_s_barrier();
    half8 xf[8][MT];                                          // this lane's B fragments of the slab: rows mt * 16 + li, k = 32 ks + 8 g .. + 7
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xf[ks][mt] = *reinterpret_cast<const half8*>(X + (mt * 16 + li) * 512 + (((ks * 4 + g) ^ li) << 4));
    // hidden chunk in registers -> H buffer: rows mt * 16 + li, hidden columns 16 w + 4 g .. + 3 of the chunk
    auto write_h = [&](unsigned char* Hw) {
        if constexpr (dbg & 4) return;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            half4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = (elem_t)(acc1[mt][r] > 0.f ? acc1[mt][r] : 0.f);
            *reinterpret_cast<half4*>(Hw + (mt * 16 + li) * 256 + (((2 * wave + (g >> 1)) ^ li) << 4) + (g & 1) * 8) = h;
        }
    };
    // one k-step of fc1 / one (k-step, tile) pair of fc2: MT MFMAs
    auto mma_fc1 = [&](const EfFrags& f, const int ks0, const int ks, const bool first) {
        if constexpr (dbg & 1) return;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc1[mt] = OPD_MFMA_16x16x32(f.w[ks], xf[ks0 + ks][mt], (first && ks == 0) ? f.b : acc1[mt]);
    };
    auto mma_fc2 = [&](const EfFrags& f, const int q) {   // q = 2 i + j
        if constexpr (dbg & 1) return;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc2[q & 1][mt] = OPD_MFMA_16x16x32(f.w[q], f.h[q >> 1][mt], acc2[q & 1][mt]);
    };
    auto read_h = [&](EfFrags& f, const unsigned char* Hr, const int ks0) {
        if constexpr (dbg & 4) return;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) f.h[i][mt] = *reinterpret_cast<const half8*>(Hr + (mt * 16 + li) * 256 + ((((ks0 + i) * 4 + g) ^ li) << 4));
    };
    // A STEP: the fragments of the next group are requested from LDS into the idle register set, then the current group's MFMAs run with the
    // re-requests of ITS ring slots (read one step ago: no wait) placed between them -- LDS latency and LDS-DMA issue hide behind the MFMAs.
    // Wait count of a step: R - (pieces of the group in registers) - (pieces of the group being read).
    // (the next group's reads come AFTER the first MFMAs of the current one: the compiler guards the current operands with lgkmcnt(0), which
    //  would otherwise also wait for the reads just issued)
#define EF_FC1_STEP(F_, KS0, FIRST, SL, NSL, NEXT)                                \
    do {                                                                           \
        EF_FENCE();                                                                \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                         \
            mma_fc1(F_, KS0, q_, FIRST);                                           \
            EF_FENCE();                                                            \
            if (q_ == 0) { NEXT; EF_FENCE(); }                                     \
            reissue(SL[q_]);                                                       \
            if (q_ == 3 && NSL == 5) reissue(SL[4]);                               \
            EF_FENCE();                                                            \
        }                                                                          \
    } while (0)
#define EF_FC2_STEP(F_, SL, NEXT)                                                 \
    do {                                                                           \
        EF_FENCE();                                                                \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                         \
            mma_fc2(F_, q_);                                                       \
            EF_FENCE();                                                            \
            if (q_ == 0) { NEXT; EF_FENCE(); }                                     \
            reissue(SL[q_]);                                                       \
            EF_FENCE();                                                            \
        }                                                                          \
    } while (0)
    EfFrags A, B;
    unsigned sa[5], sb[4];
    auto take_g0 = [&](EfFrags& f) {   // bias piece + W1 k-steps 0 .. 3 -> sa
#pragma unroll
        for (int i = 0; i < 5; ++i) sa[i] = take();
        f.b = *reinterpret_cast<const float4v*>(ring + sa[0] + lane16);
#pragma unroll
        for (int i = 0; i < 4; ++i) f.w[i] = frag(sa[1 + i]);
    };
    auto take_w4 = [&](EfFrags& f, unsigned (&sl)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { sl[i] = take(); f.w[i] = frag(sl[i]); }
    };
    auto load_img = [&](const unsigned char* img) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xf[ks][mt] = *reinterpret_cast<const half8*>(img + (mt * 16 + li) * 512 + (((ks * 4 + g) ^ li) << 4));
    };
    auto mma_tail = [&](const EfFrags& f, const int ks0, const int q) {   // q = 2 i + j: k-step ks0 + i, tile j
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc2[q & 1][mt] = OPD_MFMA_16x16x32(f.w[q], xf[ks0 + (q >> 1)][mt], acc2[q & 1][mt]);
    };
#define EF_TAIL_STEP(F_, KS0, SL, NEXT)                                            \
    do {                                                                           \
        EF_FENCE();                                                                \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                         \
            mma_tail(F_, KS0, q_);                                                 \
            EF_FENCE();                                                            \
            if (q_ == 0) { NEXT; EF_FENCE(); }                                     \
            reissue(SL[q_]);                                                       \
            EF_FENCE();                                                            \
        }                                                                          \
    } while (0)
    // ---- fc1 of chunk 0 (the pipeline fills), then the first group of chunk 1 into set A ------------------------------------------------
    unsigned s0[5], sa4[4];
    if (p.attn == nullptr) {
        ef_wait_vm<EF_R - 5>();
        take_g0(A);
    } else {
        // ---- FRONT phase (the slab holds the ATTENTION output): x = LayerNorm(res + attn . Wo^T + bo), HF:models/detr/modeling_detr.py:640-645.
        //      The stream starts with Wo's 16 pieces for this wave's two tiles (fc2's group format); same pipeline; x goes to y32 (the residual the
        //      epilogue reads back) and, as fp16, into the slab's place in LDS -- it never exists as a tensor of its own.
        ef_wait_vm<EF_R - 4>();
        take_w4(A, sa4);
        EF_TAIL_STEP(A, 0, sa4, (ef_wait_vm<EF_R - 8>(), take_w4(B, sb)));
        EF_TAIL_STEP(B, 2, sb, (ef_wait_vm<EF_R - 8>(), take_w4(A, sa4)));
        EF_TAIL_STEP(A, 4, sa4, (ef_wait_vm<EF_R - 8>(), take_w4(B, sb)));
        EF_TAIL_STEP(B, 6, sb, (ef_wait_vm<EF_R - 9>(), take_g0(A)));   // (the FFN's first group)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cidx = (2 * wave + j) * 16 + 4 * g;
            const float4v bo = *reinterpret_cast<const float4v*>(p.bo + cidx);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = m_base + mt * 16 + li;
                float4v r = float4v{0.f, 0.f, 0.f, 0.f};
                if (p.res32 && m < p.M) r = *reinterpret_cast<const float4v*>(p.res32 + (size_t)m * 256 + cidx);
                acc2[j][mt] += bo + r;
            }
        }
        RowLnOut o1{p.gamma1, p.beta1, p.y32, nullptr, nullptr, nullptr, nullptr, 0, p.M};
        o1.img_x = X; o1.img_xp = H;      // (no position table here: both images are fp16(x); H is not in use yet)
        row_ln_store<true, MT>(o1, acc2, red, m_base, wave, g, li);   // (its barriers separate every wave's slab reads from the image writes)
        EF_LGKM0();
        __builtin_amdgcn_s_barrier();
        load_img(X);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc2[j][mt] = float4v{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) s0[i] = sa[i];
    EF_FC1_STEP(A, 0, true, s0, 5, (ef_wait_vm<EF_R - 9>(), take_w4(B, sb)));
    EF_FC1_STEP(B, 4, false, sb, 4, (ef_wait_vm<EF_R - 9>(), take_g0(A)));
    write_h(H);
    EF_LGKM0();
    __builtin_amdgcn_s_barrier();
#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
        const unsigned char* const Hr = H + (c & 1) * EF_H;        // hidden chunk c (fc2 reads it)
        unsigned char* const Hw = H + ((c + 1) & 1) * EF_H;        // hidden chunk c + 1 (fc1 writes it)
        // step G0(c + 1): multiply A, request G1(c + 1) into B
        EF_FC1_STEP(A, 0, true, sa, 5, (ef_wait_vm<EF_R - 9>(), take_w4(B, sb)));
        // step G1(c + 1): multiply B, request G2(c) into A (two k-steps of W2's tiles and of hidden chunk c), publish hidden chunk c + 1
        EF_FC1_STEP(B, 4, false, sb, 4, (ef_wait_vm<EF_R - 8>(), take_w4(A, sa4), read_h(A, Hr, 0)));
        write_h(Hw);
        // step G2(c): multiply A, request G3(c) into B
        EF_FC2_STEP(A, sa4, (ef_wait_vm<EF_R - 8>(), take_w4(B, sb), read_h(B, Hr, 2)));
        // step G3(c): multiply B, request G0(c + 2) into A
        EF_FC2_STEP(B, sb, (ef_wait_vm<EF_R - 9>(), take_g0(A)));
        EF_LGKM0();
        if constexpr (!(dbg & 8)) __builtin_amdgcn_s_barrier();   // hidden chunk c + 1 is complete, every read of chunk c has returned
    }
    const int T = p.tail;
    if (T == 0) ef_wait_vm<0>();   // (the pieces requested past the end of the stream: no LDS-DMA may be in flight when the workgroup ends)
    else {
#pragma unroll
        for (int i = 0; i < 5; ++i) reissue(sa[i]);   // (the slots of the last, empty fc1 group: the stream continues with the tail's weights)
    }
    // ---- + b2 + residual, LayerNorm, outputs -----------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int cidx = (2 * wave + j) * 16 + 4 * g;
        const float4v b2 = *reinterpret_cast<const float4v*>(p.b2 + cidx);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m_base + mt * 16 + li;
            float4v r = float4v{0.f, 0.f, 0.f, 0.f};
            if (p.res32 && m < p.M) r = *reinterpret_cast<const float4v*>(p.res32 + (size_t)m * 256 + cidx);
            acc2[j][mt] += b2 + r;
        }
    }
    RowLnOut o{p.gamma, p.beta, p.y32, p.y16, p.yp16, p.pos, p.pos_ptrs, p.pos_period, p.M};
    if (T) { o.img_x = X; o.img_xp = H; }   // (the slab and the hidden chunks are dead: their 48 KiB take the two images)
    row_ln_store<true, MT>(o, acc2, red, m_base, wave, g, li);
    if (T == 0) return;
    // ---- tail projection: out[:, col_t .. + 255] = (y or y + pos) . Wt_t^T + bias_t for T passes of 256 columns (the next layer's q / k / v, or
    //      the decoder's memory keys / values): the wave's tiles 2 w, 2 w + 1 of every pass, 16 pieces per pass (4 groups of two k-steps x two
    //      tiles, as fc2's), the y fragments in registers; the first `tail_pos` passes multiply y + pos.  Same pipeline as above.
    EF_LGKM0();
    __builtin_amdgcn_s_barrier();      // both images are complete
    // Every pass = 17 pieces: the biases of the wave's two tiles (accumulator layout: lane (g, li < 8) tile 0, (g, li >= 8) tile 1) + 16 weight
    // pieces; nothing but LDS-DMA requests and the output stores touch vector memory inside the loop (a global load here makes the compiler
    // drain vmcnt(0) -- the whole ring -- at every use).
    // The wait counts are those of the FFN loop: they count LDS-DMA requests only.  The output stores of a pass are in flight meanwhile; they
    // retire out of order with respect to the requests (tools/microbench/vmorder.hip) and so can neither be relied on nor be in the way.
    const bool full = m_base + EF_TM <= p.M;
    auto take_t0 = [&](EfFrags& f) {   // bias piece + the first group (k-steps 0, 1 of both tiles) -> sa
#pragma unroll
        for (int i = 0; i < 5; ++i) sa[i] = take();
        f.b = *reinterpret_cast<const float4v*>(ring + sa[0] + g * 256);
        f.b2 = *reinterpret_cast<const float4v*>(ring + sa[0] + g * 256 + 128);
#pragma unroll
        for (int i = 0; i < 4; ++i) f.w[i] = frag(sa[1 + i]);
    };
    load_img(p.tail_pos > 0 ? H : X);
    ef_wait_vm<0>();                  // (once: the epilogue's stores; the ring is full)
    take_t0(A);
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        if (t == p.tail_pos && t > 0) load_img(X);
        const float4v bias[2] = {A.b, A.b2};
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc2[j][mt] = float4v{0.f, 0.f, 0.f, 0.f};
        unsigned s1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) s1[i] = sa[1 + i];
        const unsigned sbias = sa[0];
        // (slots are re-requested in ring order: the stream's next piece belongs into the slot that `take` will visit next)
        EF_TAIL_STEP(A, 0, s1, (ef_wait_vm<EF_R - 9>(), take_w4(B, sb), reissue(sbias)));
        EF_TAIL_STEP(B, 2, sb, (ef_wait_vm<EF_R - 8>(), take_w4(A, sa4)));
        EF_TAIL_STEP(A, 4, sa4, (ef_wait_vm<EF_R - 8>(), take_w4(B, sb)));
        EF_TAIL_STEP(B, 6, sb, (ef_wait_vm<EF_R - 9>(), take_t0(A)));   // (the next pass's bias + first group; after the last pass: five zero pieces)
        f16_t* const orow = p.tail_out + p.tail_col[t] + (2 * wave) * 16 + 4 * g;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m_base + mt * 16 + li;
            if (full || m < p.M) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    half4 h;
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[r] = (elem_t)(acc2[j][mt][r] + bias[j][r]);
                    *reinterpret_cast<half4*>(orow + (size_t)m * p.tail_ld + j * 16) = h;
                }
            }
        }
        EF_FENCE();
    }
    ef_wait_vm<0>();
#endif
}
#undef EF_FC1_STEP
#undef EF_FC2_STEP
#undef EF_TAIL_STEP

}  // namespace

hipError_t OPD_SYM(opd_launch_gemm_ln)(const GemmLnParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.K <= 0 || p.K % 64 != 0 || !p.bias || (!p.deep_k && (!p.gamma || !p.beta)) || (p.gamma && !p.beta)) return hipErrorInvalidValue;
    if ((size_t)p.M * p.K * 2 >= 0x7fffff00ull) return hipErrorInvalidValue;  // 31-bit buffer offsets
    if (p.deep_k) {   // row-owner ring (the encoder's FFN-2): no split-K slabs, no reduce launch
        if (p.yp16 && (p.pos_period <= 0 || (!p.pos && !p.pos_ptrs))) return hipErrorInvalidValue;
        OPD_SET_MAX_LDS_ONCE(gemm_ln256_ring_kernel<true>, RG_LDS);
        OPD_SET_MAX_LDS_ONCE(gemm_ln256_ring_kernel<false>, RG_LDS);
        if (p.gamma) OPD_LAUNCH(gemm_ln256_ring_kernel<true>, dim3((p.M + RG_TM - 1) / RG_TM), dim3(64 * RG_NW), RG_LDS, stream, p);
        else OPD_LAUNCH(gemm_ln256_ring_kernel<false>, dim3((p.M + RG_TM - 1) / RG_TM), dim3(64 * RG_NW), RG_LDS, stream, p);
        return hipGetLastError();
    }
    if (p.yp16) return hipErrorInvalidValue;   // (the position shadow is written by the deep-K form only)
    if (p.K == 256 && !p.kloop) {
        OPD_SET_MAX_LDS_ONCE(gemm_ln256_os_kernel, OS_LDS);
        OPD_LAUNCH(gemm_ln256_os_kernel, dim3((p.M + OS_TM - 1) / OS_TM), dim3(64 * OS_NW), OS_LDS, stream, p);
        return hipGetLastError();
    }
    constexpr int LDS = 2 * STAGE_BYTES;
    OPD_SET_MAX_LDS_ONCE(gemm_ln256_kernel, LDS);
    OPD_LAUNCH(gemm_ln256_kernel, dim3((p.M + TM - 1) / TM), dim3(256), LDS, stream, p);
    return hipGetLastError();
}

hipError_t OPD_SYM(opd_launch_enc_ffn)(const EncFfnParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.F <= 0 || p.F % 128 != 0 || (!p.x && !p.attn) || !p.wpack || !p.b2 || !p.gamma || !p.beta) return hipErrorInvalidValue;
    // front phase (attention output projection + LayerNorm in front of the FFN): its weights lead the stream; x is then made inside, and the
    // residual the epilogue adds is what the front phase wrote (y32 == res32)
    if ((p.attn && !p.pack_front) || (p.attn && (!p.bo || !p.gamma1 || !p.beta1 || !p.y32 || p.y32 != p.res32))) return hipErrorInvalidValue;
    if (p.yp16 && (p.pos_period <= 0 || (!p.pos && !p.pos_ptrs))) return hipErrorInvalidValue;
    if ((size_t)EF_PIECES(p.F / 128, p.pack_tail, p.pack_front) * 1024 >= 0x7fffff00ull) return hipErrorInvalidValue;   // 32-bit stream offsets
    if (p.pack_tail < 0 || p.pack_tail > 16 || (p.tail != 0 && p.tail != p.pack_tail) || p.tail < 0 || p.tail > 16 || p.tail_pos < 0 || p.tail_pos > p.tail || (p.tail && (!p.tail_out || p.tail_ld < 256))) return hipErrorInvalidValue;
    if (p.tail_pos > 0 && (p.pos_period <= 0 || (!p.pos && !p.pos_ptrs))) return hipErrorInvalidValue;
    const dim3 grid((p.M + EF_TM - 1) / EF_TM);
#define EF_LAUNCH(D)                                                                   \
    do {                                                                               \
        OPD_SET_MAX_LDS_ONCE(enc_ffn_kernel<D>, EF_LDS);                               \
        OPD_LAUNCH(enc_ffn_kernel<D>, grid, dim3(512), EF_LDS, stream, p);     \
        opd_last_kernel_name = "enc_ffn_kernel<" #D ">";                               \
        return hipGetLastError();                                                      \
    } while (0)
    switch (p.dbg) {
        case 0: EF_LAUNCH(0);
        case 1: EF_LAUNCH(1);
        case 2: EF_LAUNCH(2);
        case 3: EF_LAUNCH(3);
        case 4: EF_LAUNCH(4);
        case 8: EF_LAUNCH(8);
        case 15: EF_LAUNCH(15);
        default: return hipErrorInvalidValue;
    }
}

#ifndef OPD_ELEM_BF16   // (a permutation of 16-bit words and fp32 biases: the same for both element types)
size_t opd_encffn_pack_bytes(int F, int tail, int front) { return (size_t)8 * EF_PIECES(F / 128, tail, front) * 1024; }
// host: w1 [F][256], w2 [256][F] as 16-bit elements (fp16 or bf16), b1 [F] fp32 -> the eight per-wave streams of enc_ffn_kernel, in the order the
// kernel consumes them: G0(0) G1(0) G0(1) { G1(c+1) G2(c) G3(c) G0(c+2) } for c = 0 .. F/128 - 1 (zeros for fc1 chunks past the end).
//   G0(c): the bias piece (lane L = 16 g + li: b1[128 c + 16 w + 4 g .. + 3] as fp32), then k-steps 0 .. 3 of W1's tile (lane L:
//          w1[128 c + 16 w + li][32 ks + 8 g .. + 7]);  G1(c): k-steps 4 .. 7;
//   G2(c) / G3(c): W2's k-steps i = 0, 1 / 2, 3 of the chunk for the wave's tiles j = 0, 1 at 2 (i & 1) + j (lane L:
//          w2[(2 w + j) 16 + li][128 c + 32 i + 8 g .. + 7]).
// Tail (optional): wt [tail * 256][256] = the weight rows of the tail projection in PASS order, bt [tail * 256] its biases; per pass the bias piece
// (lane (g, li): bt[256 t + (2 w + (li >> 3)) 16 + 4 g .. + 3]) and 16 pieces in fc2's group format (group q: k-steps 2 q, 2 q + 1 of the wave's
// tiles 2 w, 2 w + 1 of that pass); five zero pieces at the end.
// Front (optional): wo [256][256] = the attention output projection; its 16 pieces (fc2's group format) lead every wave's stream.
void opd_encffn_pack(const uint16_t* w1, const float* b1, const uint16_t* w2, int F, const uint16_t* wt, const float* bt, int tail, const uint16_t* wo, unsigned char* out) {
    const int nch = F / 128;
    const int front = wo != nullptr;
    __builtin_memset(out, 0, opd_encffn_pack_bytes(F, tail, front));
    for (int w = 0; w < 8; ++w) {
        unsigned char* o = out + (size_t)w * EF_PIECES(nch, tail, front) * 1024;
        if (front)
            for (int q = 0; q < 4; ++q) {
                for (int L = 0; L < 64; ++L) {
                    const int g = L >> 4, li = L & 15;
                    for (int i = 0; i < 2; ++i)
                        for (int j = 0; j < 2; ++j)
                            __builtin_memcpy(o + (2 * i + j) * 1024 + L * 16, wo + (size_t)((2 * w + j) * 16 + li) * 256 + 32 * (2 * q + i) + 8 * g, 16);
                }
                o += 4 * 1024;
            }
        auto g0 = [&](const int c) {   // 5 pieces
            if (c < nch)
                for (int L = 0; L < 64; ++L) {
                    const int g = L >> 4, li = L & 15;
                    __builtin_memcpy(o + L * 16, b1 + 128 * c + 16 * w + 4 * g, 16);
                    for (int ks = 0; ks < 4; ++ks) __builtin_memcpy(o + (1 + ks) * 1024 + L * 16, w1 + (size_t)(128 * c + 16 * w + li) * 256 + 32 * ks + 8 * g, 16);
                }
            o += 5 * 1024;
        };
        auto g1 = [&](const int c) {   // 4 pieces
            if (c < nch)
                for (int L = 0; L < 64; ++L) {
                    const int g = L >> 4, li = L & 15;
                    for (int ks = 4; ks < 8; ++ks) __builtin_memcpy(o + (ks - 4) * 1024 + L * 16, w1 + (size_t)(128 * c + 16 * w + li) * 256 + 32 * ks + 8 * g, 16);
                }
            o += 4 * 1024;
        };
        auto g23 = [&](const int c, const int i0) {   // 4 pieces: k-steps i0, i0 + 1
            for (int L = 0; L < 64; ++L) {
                const int g = L >> 4, li = L & 15;
                for (int i = 0; i < 2; ++i)
                    for (int j = 0; j < 2; ++j)
                        __builtin_memcpy(o + (2 * i + j) * 1024 + L * 16, w2 + (size_t)((2 * w + j) * 16 + li) * F + 128 * c + 32 * (i0 + i) + 8 * g, 16);
            }
            o += 4 * 1024;
        };
        g0(0); g1(0); g0(1);
        for (int c = 0; c < nch; ++c) { g1(c + 1); g23(c, 0); g23(c, 2); g0(c + 2); }
        for (int t = 0; t < tail; ++t) {
            for (int L = 0; L < 64; ++L) __builtin_memcpy(o + L * 16, bt + 256 * t + (2 * w + ((L & 15) >> 3)) * 16 + 4 * (L >> 4), 16);
            o += 1024;
            for (int q = 0; q < 4; ++q) {
                for (int L = 0; L < 64; ++L) {
                    const int g = L >> 4, li = L & 15;
                    for (int i = 0; i < 2; ++i)
                        for (int j = 0; j < 2; ++j)
                            __builtin_memcpy(o + (2 * i + j) * 1024 + L * 16, wt + (size_t)(t * 256 + (2 * w + j) * 16 + li) * 256 + 32 * (2 * q + i) + 8 * g, 16);
                }
                o += 4 * 1024;
            }
        }
    }
}
#endif

// ---------------------------------------------------------------------------------------------------------------------
// Small-M linear layers of the decoder (M = batch x 100 queries): out = act(x[:, k0:k0+256] . W[:, k0:k0+256]^T + bias).
// At M = 800 every launch is a latency chain, not a throughput problem: the general kernel walks K = 256 as four
// double-buffered k-steps, i.e. four dependent L2 round trips (~7 us per launch at batch 8).  Here a workgroup stages its
// whole 64 x 256 activation tile and 64 x 256 weight tile in ONE batch of LDS-DMA (64 KiB), waits once, and runs the 32
// MFMAs per wave back to back.  Deeper reductions (FFN-2, K = 2048) are cut into 256-wide slices along gridDim.z that write
// fp32 slabs for reduce_ln256_kernel (bias in slice 0), like the split-K form of conv_gemm_dma_kernel.
// Tile: 64 rows x 64 columns, 4 waves as 2 x 2, each 32 x 32 (2 x 2 accumulator tiles).  LDS: four [64][64-k] sub-tiles
// per operand with the usual 128-byte-row XOR swizzle.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256, 2) void gemm_k256_kernel(GemmK256Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int SUB = 64 * ROW_BYTES;       // one [64 rows][64 k] sub-tile: 8 KiB
    unsigned char* const As = smem;           // 4 sub-tiles
    unsigned char* const Ws = smem + 4 * SUB; // 4 sub-tiles
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    const int m_base = blockIdx.y * 64, n_base = blockIdx.x * 64, z = blockIdx.z;
    const int k0 = z * 256;
    const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;

    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.x), 0, (unsigned)((size_t)p.M * p.ldx * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.w), 0, (unsigned)((size_t)p.N * p.ldw * 2), 0x00020000);
    // wave w stages rows 16w..16w+15 of both operands: 2 row-pieces x 4 sub-tiles each; rows >= M are out of range -> zeros
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wave * 16 + i * 8 + lrow;
        const unsigned xo = (unsigned)((m_base + r) * p.ldx + k0) * 2u + (unsigned)lchunk * 16u;
        const unsigned wo = (unsigned)((n_base + r) * p.ldw + k0) * 2u + (unsigned)lchunk * 16u;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(As + t * SUB + (wave * 2 + i) * 1024), 16,
                                                     m_base + r < p.M ? xo : 0x80000000u, t * 128, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(Ws + t * SUB + (wave * 2 + i) * 1024), 16, wo,
                                                     t * 128, 0, 0);
        }
    }
    // accumulators start from the bias (vector or row-periodic; slices z > 0 start from zero)
    float4v acc[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int n = n_base + wn * 32 + nt * 16 + g * 4;
            const int m = m_base + wm * 32 + mt * 16 + li;
            acc[nt][mt] = float4v{0.f, 0.f, 0.f, 0.f};
            if (z == 0 && m < p.M)
                acc[nt][mt] = *reinterpret_cast<const float4v*>(p.bias + (p.bias_period ? (size_t)(m % p.bias_period) * p.N : 0) + n);
        }
    OPD_DMA_BARRIER();   // drain + barrier: both tiles are in LDS
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 xf[2], wf[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) xf[mt] = *reinterpret_cast<const half8*>(As + t * SUB + swz(wm * 32 + mt * 16 + li, kk * 4 + g));
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) wf[nt] = *reinterpret_cast<const half8*>(Ws + t * SUB + swz(wn * 32 + nt * 16 + li, kk * 4 + g));
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = OPD_MFMA_16x16x32(wf[nt], xf[mt], acc[nt][mt]);
        }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int n = n_base + wn * 32 + nt * 16 + g * 4;
            const int m = m_base + wm * 32 + mt * 16 + li;
            if (m >= p.M) continue;
            float4v v = acc[nt][mt];
            if (p.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
            }
            if (p.out32) {
                *reinterpret_cast<float4v*>(p.out32 + ((size_t)z * p.M + m) * p.N + n) = v;
            } else {
                half4 h;
                h[0] = (elem_t)v[0]; h[1] = (elem_t)v[1]; h[2] = (elem_t)v[2]; h[3] = (elem_t)v[3];
                *reinterpret_cast<half4*>(p.out16 + (size_t)m * p.N + n) = h;
            }
        }
#endif
}

}  // namespace

hipError_t OPD_SYM(opd_launch_gemm_k256)(const GemmK256Params& p, hipStream_t stream) {
    if (p.M <= 0 || p.N <= 0 || p.N % 64 != 0 || p.slices < 1 || !p.bias || (!p.out16 && !p.out32)) return hipErrorInvalidValue;
    if (p.slices > 1 && (!p.out32 || p.relu)) return hipErrorInvalidValue;   // partial sums: fp32 slabs, no activation
    if (p.ldx < 256 * p.slices || p.ldw < 256 * p.slices) return hipErrorInvalidValue;
    if ((size_t)p.M * p.ldx * 2 >= 0x7fffff00ull || (size_t)p.N * p.ldw * 2 >= 0x7fffff00ull) return hipErrorInvalidValue;
    constexpr int LDS = 8 * 64 * ROW_BYTES;
    OPD_SET_MAX_LDS_ONCE(gemm_k256_kernel, LDS);
    OPD_LAUNCH(gemm_k256_kernel, dim3(p.N / 64, (p.M + 63) / 64, p.slices), dim3(256), LDS, stream, p);
    return hipGetLastError();
}
