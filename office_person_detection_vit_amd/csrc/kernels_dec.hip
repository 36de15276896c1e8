// kernels_dec.hip — the fused DETR decoder on split (hi + lo) fp16 operands, gfx950.
//
// SURVEY.md §8(a) rows a11 / a12 (HF:models/detr/modeling_detr.py:496-573 cross-attention, 650-739 decoder layer, 994-1106 decoder):
//   h = LN1(h + SelfAttn(q = k = h + qpos, v = h)) ; h = LN2(h + CrossAttn(q = h + qpos, k = mem + pos, v = mem)) ; h = LN3(h + FFN(h))
// on M = batch x 100 query rows.  Round 3 ran this as nine launches per layer of 4-11 us each (54 launches, 0.83 ms of a 3.76 ms serial
// step for 2 % of the FLOPs) on fp16 GEMM operands, and tools/drift_split.py shows that on ordinary (not fp16-representable) weights the
// decoder's operand rounding — weights as much as activations, every linear layer about equally — is the largest transformer share
// of the box drift (4.9e-4 of 8.6e-4).  Both are addressed by one design:
//
//  * Split operands.  Every GEMM input x and weight W of the decoder's linear layers is carried as TWO fp16 numbers,
//        hi = fp16(x),  lo = fp16((x - hi) * 2048)          (x - hi is exact in fp32; the scale keeps lo out of the subnormals)
//    and a product is three MFMAs into two fp32 accumulators,  W.x = [Whi.xhi] + [Whi.xlo + Wlo.xhi] / 2048  (the lo.lo term is
//    2^-22 relative): fp32-grade linear layers at 3/16 of the fp16 matrix rate instead of the 1/16 of v_mfma_f32_16x16x4_f32.
//    Attention scores and P.V stay single fp16 (their rounding sites measure 1e-5 .. 5e-5 of box drift).
//  * Row-slab workgroups with weights as MFMA A operands straight from L2.  M = 800 rows are 50 slabs of 16: a workgroup of eight
//    waves owns a slab, keeps it (hi / lo) in LDS as the B operand and reads its weight fragments (16 rows x 64 bytes per
//    wave-instruction) directly from global memory into registers, all requested up front: one L2 round trip per GEMM instead of the
//    LDS-DMA staging chain of a tile GEMM.  What needs the whole frame (self-attention K / V) or all heads (output projections) fixes
//    the five launches per layer:
//      dec_qkv_kernel        [previous FFN: reduce partial sums + b2 + residual + LN3] -> q | k | v^T                        (slab)
//      dec_self_kernel       self-attention (wave = head) + o-proj + residual + LN1 + cross-attention q projection   (frame x slab)
//      attention_kernel<SPLIT>  cross-attention over a third of the keys, unnormalised partials (kernels_attn.hip)
//      dec_cross_out_kernel  combine key splits + o-proj + residual + LN2                                               (slab)
//      dec_ffn_kernel        fc1 chunk + ReLU + fc2 partial sums over a 128-wide slice of the hidden layer    (64-row slab x 16 chunks)
//    The FFN's hidden tensor never leaves the CU; its 16 fp32 partial sums per row are summed in fixed order by the consumer
//    (dec_qkv_kernel of the next layer / heads_kernel), so results are deterministic.
//
// MFMA operand layout used throughout (as in kernels_gemm.hip): v_mfma_f32_16x16x32_f16(A = weight rows, B = data rows): lane
// (g = lane >> 4, li = lane & 15) feeds A[n0 + li][k0 + 8g .. 8g + 7] and B[row li][k0 + 8g .. 8g + 7] and receives
// out[row li][n0 + 4g + r], r = 0 .. 3.
#include <hip/hip_runtime.h>
#include <math.h>
#include "opd_kernels.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));

namespace {

constexpr int XP = 528;                 // bytes per LDS row of a [rows][256] fp16 operand: 512 + 16, so that the 16 rows a fragment read touches start 4 banks apart
constexpr int HP = 272;                 // the same for a [rows][128] operand
constexpr float LO_SCALE = 2048.0f, LO_INV = 1.0f / 2048.0f;

__device__ __forceinline__ void split1(const float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)((v - (float)hi) * LO_SCALE);
}
// three MFMAs of one split product: ah += Whi.xhi ; al += Whi.xlo ; am += Wlo.xhi  (three accumulators: no MFMA waits for its predecessor)
__device__ __forceinline__ void mma3(const half8& wh, const half8& wl, const half8& xh, const half8& xl, float4v& ah, float4v& al, float4v& am) {
    ah = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, ah, 0, 0, 0);
    al = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, al, 0, 0, 0);
    am = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, am, 0, 0, 0);
}
__device__ __forceinline__ half8 lds_frag(const unsigned char* X, const int pitch, const int row, const int ks, const int g) {
    return *reinterpret_cast<const half8*>(X + row * pitch + (ks * 32 + g * 8) * 2);
}
// weight fragments of the 16 rows n0 .. n0 + 15 of a [N][ldw] matrix, k = k0 + 32 ks + 8g .. + 7, for NKS k-steps: requested together
template <int NKS>
__device__ __forceinline__ void load_w(const f16_t* __restrict__ whi, const f16_t* __restrict__ wlo, const int ldw, const int n, const int k0,
                                       const int g, half8 (&fh)[NKS], half8 (&fl)[NKS]) {
    const size_t off = (size_t)n * ldw + k0 + g * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        fh[ks] = *reinterpret_cast<const half8*>(whi + off + ks * 32);
        fl[ks] = *reinterpret_cast<const half8*>(wlo + off + ks * 32);
    }
}
// 8 consecutive fp32 values -> hi / lo halves at X*[row][c0 .. c0 + 7]
__device__ __forceinline__ void store_split8(unsigned char* Xhi, unsigned char* Xlo, const int pitch, const int row, const int c0, const float (&v)[8]) {
    half8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) { _Float16 a, b; split1(v[j], a, b); hi[j] = a; lo[j] = b; }
    *reinterpret_cast<half8*>(Xhi + row * pitch + c0 * 2) = hi;
    *reinterpret_cast<half8*>(Xlo + row * pitch + c0 * 2) = lo;
}
__device__ __forceinline__ void store_split4(unsigned char* Xhi, unsigned char* Xlo, const int pitch, const int row, const int c0, const float4v v) {
    half4 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) { _Float16 a, b; split1(v[j], a, b); hi[j] = a; lo[j] = b; }
    *reinterpret_cast<half4*>(Xhi + row * pitch + c0 * 2) = hi;
    *reinterpret_cast<half4*>(Xlo + row * pitch + c0 * 2) = lo;
}
// Every kernel below is a LATENCY chain (a dependent global access costs ~1 us on this chip, a launch ~3): all loads that do not depend
// on computed data are requested at the top of a phase, unconditionally (out-of-range rows / keys / splits read a clamped, valid address
// and are zeroed by a select afterwards: a predicated load becomes a branch with its own wait), and the scheduler is fenced so that it
// neither sinks them to their uses nor hoists the consumers' waits.
#define DEC_FENCE() __builtin_amdgcn_sched_barrier(0)

struct Acc3 { float4v h[2], l[2], m[2]; };
__device__ __forceinline__ void acc3_zero(Acc3& a) {
#pragma unroll
    for (int j = 0; j < 2; ++j) { a.h[j] = float4v{0.f, 0.f, 0.f, 0.f}; a.l[j] = a.h[j]; a.m[j] = a.h[j]; }
}
__device__ __forceinline__ float acc3_get(const Acc3& a, const int j, const int r) { return a.h[j][r] + (a.l[j][r] + a.m[j][r]) * LO_INV; }
// [16 rows][32 NKS of K] (hi / lo in LDS, k-steps ks0 ..) x this wave's two 16-channel weight tiles (fragments in registers)
template <int NKS>
__device__ __forceinline__ void gemm16(const unsigned char* Xhi, const unsigned char* Xlo, const int ks0, const half8 (&wh)[2][NKS], const half8 (&wl)[2][NKS],
                                       const int g, const int li, Acc3& a) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const half8 xh = lds_frag(Xhi, XP, li, ks0 + ks, g), xl = lds_frag(Xlo, XP, li, ks0 + ks, g);
#pragma unroll
        for (int j = 0; j < 2; ++j) mma3(wh[j][ks], wl[j][ks], xh, xl, a.h[j], a.l[j], a.m[j]);
    }
}

// LayerNorm over the 256 channels of the slab's 16 rows in accumulator layout: lane (g, li) of wave w holds, for row li, the channels
// (2w + j) * 16 + 4g + r.  Two-pass fp32 statistics (mean, then centred variance), lanes -> 4 lane groups (shuffles) -> 8 waves (LDS).
// Every thread of the workgroup must call it (two barriers).  gm / bt: this lane's gamma / beta (loaded by the caller, early).
__device__ __forceinline__ void layernorm_acc(float4v (&v)[2], float (*red)[8][16], const int wave, const int g, const int li,
                                              const float4v (&gm)[2], const float4v (&bt)[2]) {
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) sum += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    if (g == 0) red[0][wave][li] = sum;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) mean += red[0][w][li];
    mean *= (1.0f / 256.0f);
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        v[j] -= mean;
        sq += (v[j][0] * v[j][0] + v[j][1] * v[j][1]) + (v[j][2] * v[j][2] + v[j][3] * v[j][3]);
    }
    sq += __shfl_xor(sq, 16);
    sq += __shfl_xor(sq, 32);
    if (g == 0) red[1][wave][li] = sq;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) var += red[1][w][li];
    const float rstd = 1.0f / sqrtf(var * (1.0f / 256.0f) + 1e-5f);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[j][r] = v[j][r] * rstd * gm[j][r] + bt[j][r];
}
// two-pass LayerNorm of a row held by 32 consecutive lanes, 8 channels each (the slab prologues' layout)
__device__ __forceinline__ void layernorm_row32(float (&v)[8], const float4v g0, const float4v g1, const float4v b0, const float4v b1) {
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s1 += v[j];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s1 += __shfl_xor(s1, o);
    const float mean = s1 * (1.0f / 256.0f);
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] -= mean; s2 += v[j] * v[j]; }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
    const float rstd = 1.0f / sqrtf(s2 * (1.0f / 256.0f) + 1e-5f);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = v[j] * rstd * g0[j] + b0[j]; v[4 + j] = v[4 + j] * rstd * g1[j] + b1[j]; }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// dec_qkv_kernel: grid (slabs of 16 rows), 512 threads.  One workgroup reduces the previous FFN's partial sums for its rows ONCE (16
// slabs x 16 KiB) and then runs the three projections q, k, v as six half-rounds (2 column tiles per wave x 4 k-steps each) whose weight
// fragments are double-buffered in registers: while one half multiplies, the next is in flight.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int QKV_MAXS = 16;
__global__ __launch_bounds__(512) void dec_qkv_kernel(DecQkvParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char Xhi[16 * XP];
    __shared__ __attribute__((aligned(16))) unsigned char Xlo[16 * XP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int row0 = blockIdx.x * 16;
    half8 wh[2][2][4], wl[2][2][4];   // [buffer][column tile][k-step of the half]
    auto load_half = [&](const int part, const int half, const int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) load_w<4>(p.w_hi, p.w_lo, 256, part * 256 + (2 * wave + j) * 16 + li, half * 128, g, wh[buf][j], wl[buf][j]);
    };
    {   // the slab's rows: 32 threads per row, 8 consecutive channels each; [previous FFN's sum + LN3]
        const int r = tid >> 5, c0 = (tid & 31) * 8, row = row0 + r;
        const size_t rc = (size_t)(row < p.M ? row : p.M - 1) * 256 + c0;   // (rows beyond M: a valid address, results unused)
        const float* src = p.partials ? p.h_in : p.h_out;
        float4v xa = *reinterpret_cast<const float4v*>(src + rc), xb = *reinterpret_cast<const float4v*>(src + rc + 4);
        float v[8];
        if (p.partials) {   // (uniform over the launch)
            float4v pa[QKV_MAXS], pb[QKV_MAXS];
#pragma unroll
            for (int s = 0; s < QKV_MAXS; ++s) {   // every slab requested before the first one is needed
                const float* ps = p.partials + (size_t)(s < p.nsplit ? s : p.nsplit - 1) * p.M * 256 + rc;
                pa[s] = *reinterpret_cast<const float4v*>(ps);
                pb[s] = *reinterpret_cast<const float4v*>(ps + 4);
            }
            const float4v ba = *reinterpret_cast<const float4v*>(p.b2 + c0), bb = *reinterpret_cast<const float4v*>(p.b2 + c0 + 4);
            const float4v g0 = *reinterpret_cast<const float4v*>(p.ln_g + c0), g1 = *reinterpret_cast<const float4v*>(p.ln_g + c0 + 4);
            const float4v e0 = *reinterpret_cast<const float4v*>(p.ln_b + c0), e1 = *reinterpret_cast<const float4v*>(p.ln_b + c0 + 4);
            load_half(0, 0, 0);   // q's first half: in flight behind the slabs
            DEC_FENCE();
            xa += ba; xb += bb;
#pragma unroll
            for (int s = 0; s < QKV_MAXS; ++s)   // fixed order: deterministic
                if (s < p.nsplit) { xa += pa[s]; xb += pb[s]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = xa[j]; v[4 + j] = xb[j]; }
            layernorm_row32(v, g0, g1, e0, e1);
            if (row < p.M) {
                *reinterpret_cast<float4v*>(p.h_out + (size_t)row * 256 + c0) = float4v{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<float4v*>(p.h_out + (size_t)row * 256 + c0 + 4) = float4v{v[4], v[5], v[6], v[7]};
            }
        } else {
            load_half(0, 0, 0);
            DEC_FENCE();
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = xa[j]; v[4 + j] = xb[j]; }
        }
        store_split8(Xhi, Xlo, XP, r, c0, v);
    }
    load_half(0, 1, 1);
    DEC_FENCE();
    __syncthreads();
    const int row = row0 + li;
    const bool ok = row < p.M;
    const int rcl = ok ? row : p.M - 1;
    const int fr = rcl / p.Q, qi = rcl - fr * p.Q;
#pragma unroll
    for (int part = 0; part < 3; ++part) {
        float4v bs[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) bs[j] = *reinterpret_cast<const float4v*>(p.bias + (size_t)qi * 768 + part * 256 + (2 * wave + j) * 16 + 4 * g);
        Acc3 a;
        acc3_zero(a);
        gemm16<4>(Xhi, Xlo, 0, wh[0], wl[0], g, li, a);
        DEC_FENCE();
        if (part < 2) load_half(part + 1, 0, 0);
        DEC_FENCE();
        gemm16<4>(Xhi, Xlo, 4, wh[1], wl[1], g, li, a);
        DEC_FENCE();
        if (part < 2) load_half(part + 1, 1, 1);
        DEC_FENCE();
        if (ok) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = (2 * wave + j) * 16 + 4 * g;   // channel inside this part
                half4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (_Float16)(acc3_get(a, j, r) + bs[j][r]);
                if (part == 0) *reinterpret_cast<half4*>(p.q16 + (size_t)row * 256 + c) = o;
                else if (part == 1) *reinterpret_cast<half4*>(p.k16 + (size_t)row * 256 + c) = o;
                else {   // v^T [frame][head][dim][key]
                    _Float16* vt = reinterpret_cast<_Float16*>(p.vT) + ((size_t)(fr * 8 + (c >> 5)) * 32 + (c & 31)) * 128 + qi;
#pragma unroll
                    for (int r = 0; r < 4; ++r) vt[(size_t)r * 128] = o[r];
                }
            }
        }
    }
}

// o-proj + bias + residual + LayerNorm of a 16-row slab whose attention output sits (hi / lo) in LDS; returns the normalised rows in
// accumulator layout.  rs = bias (+ residual) of this lane's channels, gm / bt = gamma / beta: loaded by the caller, early.
__device__ __forceinline__ void oproj_ln(const unsigned char* Xhi, const unsigned char* Xlo, const half8 (&wh)[2][8], const half8 (&wl)[2][8],
                                         const float4v (&rs)[2], const float4v (&gm)[2], const float4v (&bt)[2], float (*red)[8][16], const int wave,
                                         const int g, const int li, float4v (&out)[2]) {
    Acc3 a;
    acc3_zero(a);
    gemm16<8>(Xhi, Xlo, 0, wh, wl, g, li, a);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[j][r] = acc3_get(a, j, r) + rs[j][r];
    layernorm_acc(out, red, wave, g, li, gm, bt);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// dec_self_kernel: grid (ceil(Q / 16) query slabs, B frames), 512 threads: wave = head during the attention, then column tiles.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void dec_self_kernel(DecSelfParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char Xhi[16 * XP];
    __shared__ __attribute__((aligned(16))) unsigned char Xlo[16 * XP];
    __shared__ float red[2][8][16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int q0 = blockIdx.x * 16, b = blockIdx.y;
    const int Q = p.Q;
    const bool q_ok = q0 + li < Q;
    const size_t row = (size_t)b * Q + (q_ok ? q0 + li : Q - 1);   // this lane's data row (clamped: padding lanes compute on a valid row, store nothing)
    const int h = wave;
    // ---- every load of the attention and of the o-proj, requested at once --------------------------------------------------------
    const half8 qf = *reinterpret_cast<const half8*>(p.q16 + row * 256 + h * 32 + g * 8);
    half8 kf[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
        const int key = kt * 16 + li;
        kf[kt] = *reinterpret_cast<const half8*>(p.k16 + ((size_t)b * Q + (key < Q ? key : Q - 1)) * 256 + h * 32 + g * 8);   // (keys >= Q are masked below)
    }
    // V^T operand: A[dim li][k-slot 8g + j]: j < 4 -> key kb*32 + 4g + j, j >= 4 -> key kb*32 + 16 + 4g + (j - 4) (the P operand's k order)
    const _Float16* vt = reinterpret_cast<const _Float16*>(p.vT) + (size_t)(b * 8 + h) * 32 * 128;
    half4 v4[4][2][2];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int key0 = kb * 32 + hh * 16 + g * 4;   // (Q % 4 == 0: a group of 4 keys is all valid or all padding)
                v4[kb][dt][hh] = *reinterpret_cast<const half4*>(vt + (size_t)(dt * 16 + li) * 128 + (key0 < Q ? key0 : 0));
            }
    float4v rs[2], gm[2], bt[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = (2 * wave + j) * 16 + 4 * g;
        rs[j] = *reinterpret_cast<const float4v*>(p.bo + c) + *reinterpret_cast<const float4v*>(p.h + row * 256 + c);
        gm[j] = *reinterpret_cast<const float4v*>(p.ln_g + c);
        bt[j] = *reinterpret_cast<const float4v*>(p.ln_b + c);
    }
    half8 wh[2][8], wl[2][8];
#pragma unroll
    for (int j = 0; j < 2; ++j) load_w<8>(p.wo_hi, p.wo_lo, 256, (2 * wave + j) * 16 + li, 0, g, wh[j], wl[j]);
    DEC_FENCE();
    // ---- attention of head `wave`: S^T = K Q^T over the frame's Q keys, softmax per query, O^T = V^T P^T -------------------------------
    float4v s[8];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt], qf, float4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[kt][r] = kt * 16 + g * 4 + r < Q ? s[kt][r] : -INFINITY;
            mx = fmaxf(mx, s[kt][r]);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float scale2 = p.scale * 1.44269504088896340736f;
    const float mneg = -mx * scale2;
    float lsum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[kt][r] = __builtin_amdgcn_exp2f(fmaf(s[kt][r], scale2, mneg));   // masked keys: exp2(-inf) = 0
            lsum += s[kt][r];
        }
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
    float4v oacc[2] = {float4v{0.f, 0.f, 0.f, 0.f}, float4v{0.f, 0.f, 0.f, 0.f}};
    const half4 z4 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        half8 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pf[r] = (_Float16)s[2 * kb][r]; pf[4 + r] = (_Float16)s[2 * kb + 1][r]; }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const half4 lo4 = kb * 32 + g * 4 < Q ? v4[kb][dt][0] : z4, hi4 = kb * 32 + 16 + g * 4 < Q ? v4[kb][dt][1] : z4;   // (padding keys: whatever the buffer holds x 0 must stay 0)
            half8 vf;
#pragma unroll
            for (int r = 0; r < 4; ++r) { vf[r] = lo4[r]; vf[4 + r] = hi4[r]; }
            oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, oacc[dt], 0, 0, 0);
        }
    }
    const float inv = 1.0f / lsum;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) store_split4(Xhi, Xlo, XP, li, h * 32 + dt * 16 + g * 4, oacc[dt] * inv);
    __syncthreads();
    // ---- o-proj + residual + LN1 --------------------------------------------------------------------------------------
    float4v hn[2];
    oproj_ln(Xhi, Xlo, wh, wl, rs, gm, bt, red, wave, g, li, hn);
    // cross-attention query projection: weights and bias table requested now, they travel during the exchange below
    DEC_FENCE();
#pragma unroll
    for (int j = 0; j < 2; ++j) load_w<8>(p.wq_hi, p.wq_lo, 256, (2 * wave + j) * 16 + li, 0, g, wh[j], wl[j]);
    float4v bs[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bs[j] = *reinterpret_cast<const float4v*>(p.rbq + (size_t)(q_ok ? q0 + li : 0) * 256 + (2 * wave + j) * 16 + 4 * g);
    DEC_FENCE();
    // (layernorm_acc's barriers lie between the o-proj's last fragment read and these writes)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = (2 * wave + j) * 16 + 4 * g;
        if (q_ok) *reinterpret_cast<float4v*>(p.h + row * 256 + c) = hn[j];
        store_split4(Xhi, Xlo, XP, li, c, hn[j]);
    }
    __syncthreads();
    // ---- q_c = h . Wq_c^T + (qpos . Wq_c^T + bq_c) ----------------------------------------------------------------------
    Acc3 a;
    acc3_zero(a);
    gemm16<8>(Xhi, Xlo, 0, wh, wl, g, li, a);
    if (q_ok) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            half4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (_Float16)(acc3_get(a, j, r) + bs[j][r]);
            *reinterpret_cast<half4*>(p.qc16 + row * 256 + (2 * wave + j) * 16 + 4 * g) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// dec_cross_out_kernel: grid (slabs of 16 rows), 512 threads.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int CROSS_MAXS = 6;
__global__ __launch_bounds__(512) void dec_cross_out_kernel(DecCrossOutParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char Xhi[16 * XP];
    __shared__ __attribute__((aligned(16))) unsigned char Xlo[16 * XP];
    __shared__ float red[2][8][16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int row0 = blockIdx.x * 16;
    const int row = row0 + li;
    const bool ok = row < p.M;
    const int rcl = ok ? row : p.M - 1;
    {   // combine the key splits: o = sum_s 2^(m_s - m) O_s / sum_s 2^(m_s - m) l_s ; 32 threads per row, 8 channels (a quarter head) each
        const int r = tid >> 5, c0 = (tid & 31) * 8, prow = row0 + r < p.M ? row0 + r : p.M - 1, head = c0 >> 5;
        float2v ml[CROSS_MAXS];
        float4v oa[CROSS_MAXS], ob[CROSS_MAXS];
#pragma unroll
        for (int s = 0; s < CROSS_MAXS; ++s) {
            const size_t sr = (size_t)(s < p.splits ? s : p.splits - 1) * p.M + prow;
            ml[s] = *reinterpret_cast<const float2v*>(p.part_ml + (sr * 8 + head) * 2);
            oa[s] = *reinterpret_cast<const float4v*>(p.part_o + sr * 256 + c0);
            ob[s] = *reinterpret_cast<const float4v*>(p.part_o + sr * 256 + c0 + 4);
        }
        DEC_FENCE();
        float mmax = -INFINITY;
#pragma unroll
        for (int s = 0; s < CROSS_MAXS; ++s)
            if (s < p.splits) mmax = fmaxf(mmax, ml[s][0]);
        float L = 0.f;
        float4v va = {0.f, 0.f, 0.f, 0.f}, vb = va;
#pragma unroll
        for (int s = 0; s < CROSS_MAXS; ++s) {   // fixed order
            // a split whose keys are all masked (or that is empty) carries no weight: its reference is -inf, its sums are zero
            const float w = (s < p.splits && ml[s][0] != -INFINITY) ? __builtin_amdgcn_exp2f(ml[s][0] - mmax) : 0.f;
            L = fmaf(w, w != 0.f ? ml[s][1] : 0.f, L);
#pragma unroll
            for (int j = 0; j < 4; ++j) { va[j] = fmaf(w, w != 0.f ? oa[s][j] : 0.f, va[j]); vb[j] = fmaf(w, w != 0.f ? ob[s][j] : 0.f, vb[j]); }
        }
        const float inv = 1.0f / L;
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = va[j] * inv; v[4 + j] = vb[j] * inv; }
        store_split8(Xhi, Xlo, XP, r, c0, v);
    }
    float4v rs[2], gm[2], bt[2];
    const float* res = p.res + (size_t)(p.res_period > 0 ? rcl % p.res_period : rcl) * 256;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = (2 * wave + j) * 16 + 4 * g;
        rs[j] = *reinterpret_cast<const float4v*>(p.bo + c) + *reinterpret_cast<const float4v*>(res + c);
        gm[j] = *reinterpret_cast<const float4v*>(p.ln_g + c);
        bt[j] = *reinterpret_cast<const float4v*>(p.ln_b + c);
    }
    half8 wh[2][8], wl[2][8];
#pragma unroll
    for (int j = 0; j < 2; ++j) load_w<8>(p.wo_hi, p.wo_lo, 256, (2 * wave + j) * 16 + li, 0, g, wh[j], wl[j]);
    DEC_FENCE();
    __syncthreads();
    float4v hn[2];
    oproj_ln(Xhi, Xlo, wh, wl, rs, gm, bt, red, wave, g, li, hn);
    if (ok) {
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<float4v*>(p.h + (size_t)row * 256 + (2 * wave + j) * 16 + 4 * g) = hn[j];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// dec_ffn_kernel: grid (slabs of 64 rows, F / 128 hidden chunks), 512 threads, 100 KiB of LDS.
//   fc1: wave w owns hidden tile w of the chunk (16 channels) for the slab's four 16-row tiles; fc2: wave w owns output tiles 2w, 2w + 1.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int FFN_ROWS = 64;
constexpr int FFN_LDS = 2 * FFN_ROWS * XP + 2 * FFN_ROWS * HP;

__global__ __launch_bounds__(512) void dec_ffn_kernel(DecFfnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Xhi = smem;
    unsigned char* const Xlo = smem + FFN_ROWS * XP;
    unsigned char* const Hhi = smem + 2 * FFN_ROWS * XP;
    unsigned char* const Hlo = Hhi + FFN_ROWS * HP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int row0 = blockIdx.x * FFN_ROWS, chunk = blockIdx.y;
    // every load of the kernel, requested at once: the slab's rows, W1's fragments, W2's fragments, b1
    const int r = tid >> 3, t8 = tid & 7;   // rows -> hi / lo in LDS: 8 threads per row, four 8-channel pieces each
    const size_t xrow = (size_t)(row0 + r < p.M ? row0 + r : p.M - 1) * 256;
    float4v xa[4], xb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        xa[i] = *reinterpret_cast<const float4v*>(p.h + xrow + (i * 8 + t8) * 8);
        xb[i] = *reinterpret_cast<const float4v*>(p.h + xrow + (i * 8 + t8) * 8 + 4);
    }
    half8 w1h[8], w1l[8];
    load_w<8>(p.w1_hi, p.w1_lo, 256, chunk * OPD_DEC_FFN_CHUNK + wave * 16 + li, 0, g, w1h, w1l);
    const float4v b1 = *reinterpret_cast<const float4v*>(p.b1 + chunk * OPD_DEC_FFN_CHUNK + wave * 16 + 4 * g);
    half8 w2h[2][4], w2l[2][4];   // rows = output channels, k = this chunk's 128 hidden channels
#pragma unroll
    for (int j = 0; j < 2; ++j) load_w<4>(p.w2_hi, p.w2_lo, p.F, (2 * wave + j) * 16 + li, chunk * OPD_DEC_FFN_CHUNK, g, w2h[j], w2l[j]);
    DEC_FENCE();
    const bool rok = row0 + r < p.M;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = rok ? xa[i][j] : 0.f; v[4 + j] = rok ? xb[i][j] : 0.f; }
        store_split8(Xhi, Xlo, XP, r, (i * 8 + t8) * 8, v);
    }
    __syncthreads();
    // ---- hidden chunk = relu(x . W1^T + b1) ---------------------------------------------------------------------------------
    float4v ah[4], al[4], am[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) { ah[mt] = float4v{0.f, 0.f, 0.f, 0.f}; al[mt] = ah[mt]; am[mt] = ah[mt]; }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const half8 xh = lds_frag(Xhi, XP, mt * 16 + li, ks, g), xl = lds_frag(Xlo, XP, mt * 16 + li, ks, g);
            mma3(w1h[ks], w1l[ks], xh, xl, ah[mt], al[mt], am[mt]);
        }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float4v hid;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { const float tv = ah[mt][rr] + (al[mt][rr] + am[mt][rr]) * LO_INV + b1[rr]; hid[rr] = tv > 0.f ? tv : 0.f; }
        store_split4(Hhi, Hlo, HP, mt * 16 + li, wave * 16 + 4 * g, hid);
    }
    __syncthreads();
    // ---- partial = hidden chunk . W2[:, chunk]^T ----------------------------------------------------------------------------
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        Acc3 a;
        acc3_zero(a);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const half8 xh = lds_frag(Hhi, HP, mt * 16 + li, ks, g), xl = lds_frag(Hlo, HP, mt * 16 + li, ks, g);
#pragma unroll
            for (int j = 0; j < 2; ++j) mma3(w2h[j][ks], w2l[j][ks], xh, xl, a.h[j], a.l[j], a.m[j]);
        }
        const int row = row0 + mt * 16 + li;
        if (row < p.M) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float4v o;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) o[rr] = acc3_get(a, j, rr);
                *reinterpret_cast<float4v*>(p.partials + ((size_t)chunk * p.M + row) * 256 + (2 * wave + j) * 16 + 4 * g) = o;
            }
        }
    }
}

}  // namespace

// host: x -> (fp16(x), fp16((x - fp16(x)) * 2048)) as raw half bits
void opd_split_f16(const float* w, size_t n, f16_t* hi, f16_t* lo) {
    for (size_t i = 0; i < n; ++i) {
        const _Float16 h = (_Float16)w[i];
        const _Float16 l = (_Float16)((w[i] - (float)h) * 2048.0f);
        __builtin_memcpy(&hi[i], &h, 2);
        __builtin_memcpy(&lo[i], &l, 2);
    }
}

hipError_t opd_launch_dec_qkv(const DecQkvParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.Q <= 0 || !p.h_out || !p.w_hi || !p.w_lo || !p.bias || !p.q16 || !p.k16 || !p.vT) return hipErrorInvalidValue;
    if (p.partials && (!p.h_in || !p.b2 || !p.ln_g || !p.ln_b || p.nsplit < 1)) return hipErrorInvalidValue;
    if (p.Q > 128 || p.M % p.Q != 0) return hipErrorInvalidValue;   // v^T rows hold 128 keys; rows are (frame, query)
    if (p.partials && p.nsplit > QKV_MAXS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(dec_qkv_kernel, dim3((p.M + 15) / 16), dim3(512), 0, stream, p);
    return hipGetLastError();
}
hipError_t opd_launch_dec_self(const DecSelfParams& p, hipStream_t stream) {
    if (p.B <= 0 || p.Q <= 0 || p.Q > 128 || (p.Q & 3) || !p.q16 || !p.k16 || !p.vT || !p.h || !p.wo_hi || !p.wo_lo || !p.bo || !p.ln_g || !p.ln_b ||
        !p.wq_hi || !p.wq_lo || !p.rbq || !p.qc16)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(dec_self_kernel, dim3((p.Q + 15) / 16, p.B), dim3(512), 0, stream, p);
    return hipGetLastError();
}
hipError_t opd_launch_dec_cross_out(const DecCrossOutParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.splits < 1 || p.splits > CROSS_MAXS || !p.part_o || !p.part_ml || !p.res || !p.h || !p.wo_hi || !p.wo_lo || !p.bo || !p.ln_g || !p.ln_b || p.res_period < 0)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(dec_cross_out_kernel, dim3((p.M + 15) / 16), dim3(512), 0, stream, p);
    return hipGetLastError();
}
hipError_t opd_launch_dec_ffn(const DecFfnParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.F <= 0 || p.F % OPD_DEC_FFN_CHUNK != 0 || !p.h || !p.w1_hi || !p.w1_lo || !p.b1 || !p.w2_hi || !p.w2_lo || !p.partials) return hipErrorInvalidValue;
    OPD_SET_MAX_LDS_ONCE(dec_ffn_kernel, FFN_LDS);
    hipLaunchKernelGGL(dec_ffn_kernel, dim3((p.M + FFN_ROWS - 1) / FFN_ROWS, p.F / OPD_DEC_FFN_CHUNK), dim3(512), FFN_LDS, stream, p);
    return hipGetLastError();
}
