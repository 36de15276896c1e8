// kernels_gemm.hip — implicit-GEMM convolution / linear layer for gfx950 (CDNA4), fp16 MFMA, fp32 accumulate.
//
// Covers SURVEY.md §8(a) rows a3 (stem 7x7 s2), a4 (bottleneck 1x1 / 3x3 / strided shortcut, FrozenBN folded,
// residual add + ReLU fused in the epilogue), a6 (input_projection) and every Linear of a8-a12 (q/k/v/o, fc1, fc2).
// Reference arithmetic: HF:models/resnet/modeling_resnet.py:72-93,139-178 ; HF:models/detr/modeling_detr.py:576-590.
//
// Formulation: out[m][n] = sum_k A[m][k] * Wt[n][k], m = (b, oh, ow) flattened (NHWC), k = (kh, kw, cin) with cin
// fastest, n = cout.  Both operands are K-contiguous, so both tiles are staged the same way (16-byte chunks) and both
// MFMA fragments are one ds_read_b128 per lane.
//
// Tile: 128 (m) x BN (n) x 64 (k) per workgroup of 256 threads = 4 waves (64-wide), each wave 64(m) x BN/2(n) as
// 16x16x32 MFMA tiles.  LDS: double-buffered A/B tiles with a 16-byte-chunk XOR swizzle (chunk ^= row & 7) so the
// ds_read_b128 fragment reads of a 128-byte-row tile are bank-conflict free; global->register->LDS staging with the
// next tile's global loads in flight during the MFMAs (one barrier per k-step).  The epilogue stages the fp32 tile
// through LDS so that global stores are 16-byte, row-contiguous (NHWC rows), with bias / residual / ReLU fused.
//
// MFMA operand orientation: the WEIGHT fragment is the A operand and the ACTIVATION fragment the B operand, so the
// accumulator holds D[row = n][col = m]: lane l owns 4 consecutive n for one m (col = l & 15, row = 4*(l >> 4) + reg).
#include <hip/hip_runtime.h>
#include "opd_kernels.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128;
constexpr int BK = 64;           // halfs per k-step = 128 bytes per tile row
constexpr int ROW_BYTES = BK * 2;

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROW_BYTES + ((chunk ^ (row & 7)) << 4); }

template <int BN>
struct Smem {
    static constexpr int A_BYTES = BM * ROW_BYTES;
    static constexpr int B_BYTES = BN * ROW_BYTES;
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static constexpr int LDC = BN + 4;  // fp32 words per row of the epilogue tile
    static constexpr int C_BYTES = BM * LDC * 4;
    static constexpr int TOTAL = (2 * STAGE > C_BYTES) ? 2 * STAGE : C_BYTES;
};

__device__ __forceinline__ uint4 ldg16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ uint2 ldg8(const void* p) { return *reinterpret_cast<const uint2*>(p); }

template <int BN, bool STEM>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(ConvGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using S = Smem<BN>;
    constexpr int NT = BN / 32;        // 16-wide n tiles per wave
    constexpr int B_LOADS = BN / 32;   // 16-byte chunks of the B tile per thread

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_n = p.N / BN;
    const int tile_n = blockIdx.x % tiles_n;
    const int tile_m = blockIdx.x / tiles_n;
    const int m_base = tile_m * BM;
    const int n_base = tile_n * BN;

    // ---- per-thread staging coordinates -------------------------------------------------------------------------
    const int chunk = tid & 7;     // 16-byte chunk inside the 128-byte tile row
    const int row0 = tid >> 3;     // rows row0 + 32*i
    // A rows: decompose m -> (b, oh, ow) once
    long long a_base[4];           // element offset of pixel (b, 0, 0)
    int a_ih0[4], a_iw0[4];
    bool a_ok[4];
    {
        const int ohw = p.OH * p.OW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m_base + row0 + 32 * i;
            a_ok[i] = m < p.M;
            const int mm = a_ok[i] ? m : 0;
            const int b = mm / ohw;
            const int r = mm - b * ohw;
            const int oh = r / p.OW;
            const int ow = r - oh * p.OW;
            a_base[i] = (long long)b * p.H * p.W;
            a_ih0[i] = oh * p.stride - p.pad;
            a_iw0[i] = ow * p.stride - p.pad;
        }
    }
    const f16_t* wrow[B_LOADS];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) wrow[i] = p.w + (size_t)(n_base + row0 + 32 * i) * p.K + chunk * 8;

    const int nk = p.K / BK;
    const int kpc = STEM ? 1 : p.Cin / BK;  // k-steps per filter tap
    int tap_kh = 0, tap_kw = 0, tap_c = 0;  // state of the NEXT k-step to be loaded

    uint4 ra[4], rb[B_LOADS];

    auto load_regs = [&](int ks) {
        if constexpr (STEM) {
            // k-step = 2 filter rows x (8 pixels x 4 channels); chunk = 2 pixels of NHWC4 input (8-byte aligned)
            const int kh = ks * 2 + (chunk >> 2);
            const int px = (chunk & 3) * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ih = a_ih0[i] + kh;
                const int iw = a_iw0[i] + px;
                const bool rowok = a_ok[i] && kh < 7 && (unsigned)ih < (unsigned)p.H;
                const f16_t* src = p.x + ((a_base[i] + (long long)ih * p.W + iw) << 2);
                uint2 lo = make_uint2(0u, 0u), hi = make_uint2(0u, 0u);
                if (rowok && (unsigned)iw < (unsigned)p.W) lo = ldg8(src);
                if (rowok && (unsigned)(iw + 1) < (unsigned)p.W) hi = ldg8(src + 4);
                ra[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ih = a_ih0[i] + tap_kh;
                const int iw = a_iw0[i] + tap_kw;
                const bool ok = a_ok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (ok) v = ldg16(p.x + (a_base[i] + (long long)ih * p.W + iw) * p.Cin + tap_c * BK + chunk * 8);
                ra[i] = v;
            }
            if (++tap_c == kpc) {
                tap_c = 0;
                if (++tap_kw == p.KW) { tap_kw = 0; ++tap_kh; }
            }
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) rb[i] = ldg16(wrow[i] + (size_t)ks * BK);
    };
    auto store_lds = [&](int buf) {
        unsigned char* As = smem + buf * S::STAGE;
        unsigned char* Bs = As + S::A_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(As + swz(row0 + 32 * i, chunk)) = ra[i];
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) *reinterpret_cast<uint4*>(Bs + swz(row0 + 32 * i, chunk)) = rb[i];
    };

    float4v acc[NT][4];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = float4v{0.f, 0.f, 0.f, 0.f};

    load_regs(0);
    store_lds(0);
    __syncthreads();

    const int frow = lane & 15;   // fragment row (m for activations, n for weights)
    const int fchk = lane >> 4;   // 16-byte k chunk within a 32-wide k sub-step

    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) load_regs(ks + 1);
        const unsigned char* As = smem + buf * S::STAGE;
        const unsigned char* Bs = As + S::A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 xf[4], wf[NT];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                xf[mt] = *reinterpret_cast<const half8*>(As + swz(wm * 64 + mt * 16 + frow, kk * 4 + fchk));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                wf[nt] = *reinterpret_cast<const half8*>(Bs + swz(wn * (BN / 2) + nt * 16 + frow, kk * 4 + fchk));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
        }
        if (ks + 1 < nk) store_lds(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: accumulators -> LDS (fp32 [m][n]) -> fused bias/residual/ReLU -> 16-byte row-contiguous stores --
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = wm * 64 + mt * 16 + (lane & 15);
            const int n = wn * (BN / 2) + nt * 16 + (lane >> 4) * 4;
            *reinterpret_cast<float4v*>(Cs + m * S::LDC + n) = acc[nt][mt];
        }
    __syncthreads();

    constexpr int CPR = BN / 8;  // 8-wide column groups per tile row
    for (int idx = tid; idx < BM * CPR; idx += 256) {
        const int r = idx / CPR;
        const int c8 = idx - r * CPR;
        const int m = m_base + r;
        if (m >= p.M) continue;
        const int n = n_base + c8 * 8;
        float v[8];
        {
            const float4v lo = *reinterpret_cast<const float4v*>(Cs + r * S::LDC + c8 * 8);
            const float4v hi = *reinterpret_cast<const float4v*>(Cs + r * S::LDC + c8 * 8 + 4);
            v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
            v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        }
        {
            const float* bp = p.bias + (p.bias_period > 0 ? (size_t)(m % p.bias_period) * p.N : 0) + n;
            const float4v b0 = *reinterpret_cast<const float4v*>(bp);
            const float4v b1 = *reinterpret_cast<const float4v*>(bp + 4);
            v[0] += b0[0]; v[1] += b0[1]; v[2] += b0[2]; v[3] += b0[3];
            v[4] += b1[0]; v[5] += b1[1]; v[6] += b1[2]; v[7] += b1[3];
        }
        const size_t o = (size_t)m * p.N + n;
        if (p.res16) {
            const half8 rr = *reinterpret_cast<const half8*>(p.res16 + o);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)rr[j];
        }
        if (p.res32) {
            const float4v r0 = *reinterpret_cast<const float4v*>(p.res32 + o);
            const float4v r1 = *reinterpret_cast<const float4v*>(p.res32 + o + 4);
            v[0] += r0[0]; v[1] += r0[1]; v[2] += r0[2]; v[3] += r0[3];
            v[4] += r1[0]; v[5] += r1[1]; v[6] += r1[2]; v[7] += r1[3];
        }
        if (p.relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
        }
        if (p.out_f32) {
            float* op = reinterpret_cast<float*>(p.out) + o;
            *reinterpret_cast<float4v*>(op) = float4v{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<float4v*>(op + 4) = float4v{v[4], v[5], v[6], v[7]};
            if (p.out16_aux) {
                half8 h;
#pragma unroll
                for (int j = 0; j < 8; ++j) h[j] = (_Float16)v[j];
                *reinterpret_cast<half8*>(p.out16_aux + o) = h;
            }
        } else {
            half8 h;
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = (_Float16)v[j];
            *reinterpret_cast<half8*>(reinterpret_cast<f16_t*>(p.out) + o) = h;
        }
    }
}

template <int BN, bool STEM>
hipError_t launch(const ConvGemmParams& p, hipStream_t stream) {
    using S = Smem<BN>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_kernel<BN, STEM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, S::TOTAL);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tiles_n = p.N / BN;
    hipLaunchKernelGGL((conv_gemm_kernel<BN, STEM>), dim3(tiles_m * tiles_n), dim3(256), S::TOTAL, stream, p);
    return hipGetLastError();
}

}  // namespace

hipError_t opd_launch_conv_gemm(const ConvGemmParams& p, hipStream_t stream) {
    // host-side shape contract of the kernel (checked before every launch: a violated assumption would fault the GPU)
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.N % 64) != 0 || (p.K % BK) != 0) return hipErrorInvalidValue;
    if (p.stem) {
        if (p.K != 256 || p.KH != 7 || p.KW != 7 || p.stride != 2 || p.pad != 3) return hipErrorInvalidValue;
    } else {
        if ((p.Cin % BK) != 0 || p.K != p.KH * p.KW * p.Cin) return hipErrorInvalidValue;
    }
    if ((long long)p.B * p.OH * p.OW != p.M) return hipErrorInvalidValue;
    const int tiles_m = (p.M + BM - 1) / BM;
    const bool wide = (p.N % 128 == 0) && ((long long)tiles_m * (p.N / 128) >= 384);
    if (p.stem) return launch<64, true>(p, stream);
    return wide ? launch<128, false>(p, stream) : launch<64, false>(p, stream);
}
