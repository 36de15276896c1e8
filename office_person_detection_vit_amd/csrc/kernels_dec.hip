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
//        hi = fp16(x),  lo = fp16((x - hi) * 2048)          (x - hi is exact in fp32; the scale keeps lo out of the subnormals, and
//                                                             |x| < 2^-14 travels in lo alone: the matrix pipe flushes subnormal inputs)
//    and a product is three MFMAs into two fp32 accumulators,  W.x = [Whi.xhi] + [Whi.xlo + Wlo.xhi] / 2048  (the lo.lo term is
//    2^-22 relative): fp32-grade linear layers at 3/16 of the fp16 matrix rate instead of the 1/16 of v_mfma_f32_16x16x4_f32.
//    Attention scores and P.V stay single fp16 (their rounding sites measure 1e-5 .. 5e-5 of box drift).
//  * Row-slab workgroups, weights in MFMA-FRAGMENT ORDER through wave-private LDS-DMA rings.  M = 800 rows are 50 slabs of 16: a
//    workgroup of eight waves owns a slab and keeps it (hi / lo) in LDS as the B operand; wave w owns output columns 32w .. 32w + 31.  The
//    loader stores every weight matrix so that the 1 KiB a wave's MFMA needs as its A operand for one (16-column tile, 32-wide k-step) is
//    contiguous, lane L's 16 bytes at offset 16 L, hi block then lo block (opd_split_f16_frag): a wave streams ITS tiles with
//    global_load_lds (1 KiB per instruction, fully coalesced) into its own ring of LDS slots and reads them back lane-linear
//    (conflict free), with counted vmcnt waits and no barrier.  Measured (tools/microbench/oneshot.hip): 256 KiB per workgroup arrive
//    0.7 us after the launch floor by LDS-DMA, against 5.8 us through coalesced VGPR loads and 12.8 us through 16 x 64-byte fragment
//    loads (the first form of this file: 23 us for the self-attention block).  What needs the whole frame (self-attention K / V) or
//    all heads (output projections) fixes the five launches per layer:
//      dec_qkv_kernel        [previous FFN: reduce partial sums + b2 + residual + LN3] -> q | k | v^T                        (slab)
//      dec_self_kernel       self-attention (wave = head) + o-proj + residual + LN1 + cross-attention q projection   (frame x slab)
//      attention_kernel<SPLIT>  cross-attention over a third of the keys, unnormalised partials (kernels_attn.hip)
//      dec_cross_out_kernel  combine key splits + o-proj + residual + LN2                                               (slab)
//      dec_ffn_kernel        fc1 chunk + ReLU + fc2 partial sums over a 128-wide slice of the hidden layer    (64-row slab x 16 chunks)
//    The FFN's hidden tensor never leaves the CU; its 16 fp32 partial sums per row are summed in fixed order by the consumer
//    (dec_qkv_kernel of the next layer / heads2_kernel), so results are deterministic.  heads2_kernel (end of this file) runs the class
//    head and the box MLP the same way.
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

// (the matrix pipe reads fp16 SUBNORMAL inputs as zero — measured: elements |x| < 2^-14 lost their hi part, 6e-5 absolute — while the
//  conversion produces them: such a value travels in lo alone, x * 2048 < 0.125 is a normal fp16 number)
__device__ __forceinline__ void split1(const float v, _Float16& hi, _Float16& lo) {
    hi = fabsf(v) < 6.103515625e-5f ? (_Float16)0.f : (_Float16)v;
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
#define DEC_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// tools only: wave 0 stamps the shader clock (s_memtime) into dec_ts[i]; DEC_STAMP_FLUSH writes trace[block][0..7] when the kernel ends.
// (Round 5: the stamps used to be stored where they were taken -- a store between the ring's requests, inside the window of its counted
//  waits, which tools/scan_dma_waits.py flags once no kernel is exempted by name: stores retire out of order with respect to LDS-DMA requests.)
#define DEC_STAMP(tr, i)                                                                                   \
    do {                                                                                                   \
        if (tr) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dec_ts[i])::"memory");          \
    } while (0)
#define DEC_STAMP_FLUSH(tr)                                                                                \
    do {                                                                                                   \
        if ((tr) && threadIdx.x == 0) {                                                                    \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                               \
            _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                               \
                (tr)[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + i_] = dec_ts[i_];                 \
        }                                                                                                  \
    } while (0)

struct Acc3 { float4v h[2], l[2], m[2]; };
__device__ __forceinline__ void acc3_zero(Acc3& a) {
#pragma unroll
    for (int j = 0; j < 2; ++j) { a.h[j] = float4v{0.f, 0.f, 0.f, 0.f}; a.l[j] = a.h[j]; a.m[j] = a.h[j]; }
}
__device__ __forceinline__ float acc3_get(const Acc3& a, const int j, const int r) { return a.h[j][r] + (a.l[j][r] + a.m[j][r]) * LO_INV; }

// ---- wave-private LDS-DMA ring ------------------------------------------------------------------------------------------------------
// A wave consumes a fixed SEQUENCE of 1-KiB pieces (weight fragment blocks, rows of fp32 partial sums) in order; piece n lives in slot
// n mod R of the wave's ring.  R pieces are requested up front; when pieces [first, first + count) have been consumed (their LDS reads
// have returned: lgkmcnt(0)) pieces [first + R, ...) are requested into the freed slots.  LDS-DMA requests retire in issue order AMONG
// THEMSELVES, so piece `last` has landed once at most (pieces issued so far) - 1 - last requests are outstanding.  Stores and loads into
// registers retire out of order with respect to an older request (tools/microbench/vmorder.hip): they are never part of a count, and as
// extra outstanding operations they only make a wait longer than necessary, never shorter.  Every index is a compile-time constant after
// unrolling.
__device__ __forceinline__ void dma1k(const unsigned char* src_lane, unsigned char* slot) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_lane, (__attribute__((address_space(3))) void*)slot, 16, 0, 0);
}
__device__ __forceinline__ void wait_vm(const int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // (never taken with R <= 16 and groups >= 4; safe if it were)
    }
}
template <int R, int TOTAL>
struct PieceRing {
    unsigned char* ring;   // this wave's R KiB of LDS
    int lane16;
    template <typename Src>
    __device__ __forceinline__ void prime(Src&& src) {
#pragma unroll
        for (int n = 0; n < (R < TOTAL ? R : TOTAL); ++n) dma1k(src(n) + lane16, ring + n * 1024);
    }
    __device__ __forceinline__ void wait(const int first, const int last) const { wait_vm((TOTAL < R + first ? TOTAL : R + first) - 1 - last); }
    template <typename Src>
    __device__ __forceinline__ void advance(Src&& src, const int first, const int count) {
        DEC_LGKM0();
#pragma unroll
        for (int n = first; n < first + count; ++n)
            if (n + R < TOTAL) dma1k(src(n + R) + lane16, ring + ((n + R) % R) * 1024);
    }
    __device__ __forceinline__ const unsigned char* slot(const int n) const { return ring + (n % R) * 1024; }
};
// one group of 8 pieces = four k-steps (hi block, lo block each) of one 16-column weight tile, against rows li of X (k-steps ks0 ..)
template <typename RingT>
__device__ __forceinline__ void gemm_group(const RingT& rg, const int first, const unsigned char* Xhi, const unsigned char* Xlo, const int xrow, const int ks0,
                                           const int lane16, const int g, float4v& ah, float4v& al, float4v& am) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const half8 wh = *reinterpret_cast<const half8*>(rg.slot(first + 2 * i) + lane16);
        const half8 wl = *reinterpret_cast<const half8*>(rg.slot(first + 2 * i + 1) + lane16);
        const half8 xh = lds_frag(Xhi, XP, xrow, ks0 + i, g), xl = lds_frag(Xlo, XP, xrow, ks0 + i, g);
        mma3(wh, wl, xh, xl, ah, al, am);
    }
}
// [16 rows][256] x this wave's two 16-column tiles of a [256 k] weight matrix = sequence pieces [seq0, seq0 + 32): tile j, half hf at
// seq0 + 16 j + 8 hf
template <typename RingT, typename Src>
__device__ __forceinline__ void gemm256(RingT& rg, Src&& src, const int seq0, const unsigned char* Xhi, const unsigned char* Xlo, const int lane16, const int g,
                                        const int li, Acc3& a) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int first = seq0 + 16 * j + 8 * hf;
            rg.wait(first, first + 7);
            gemm_group(rg, first, Xhi, Xlo, li, 4 * hf, lane16, g, a.h[j], a.l[j], a.m[j]);
            rg.advance(src, first, 8);
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
    DEC_LGKM0();
    __builtin_amdgcn_s_barrier();   // (raw barriers: __syncthreads would also drain the weight pieces in flight)
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
    DEC_LGKM0();
    __builtin_amdgcn_s_barrier();
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

// LDS map of the 16-row slab kernels: X hi / lo, LayerNorm scratch, then the eight waves' rings of 16 KiB
constexpr int SLAB_X = 2 * 16 * XP;              // 16 896
constexpr int SLAB_RED = SLAB_X;                 // float [2][8][16] = 1 KiB
constexpr int SLAB_RING = 18432;                 // 1 KiB aligned
constexpr int SLAB_R = 16;
constexpr int SLAB_LDS = SLAB_RING + 8 * SLAB_R * 1024;   // 149 504

// ---------------------------------------------------------------------------------------------------------------------------------
// dec_qkv_kernel: grid (slabs of 16 rows, 3 = q | k | v), 512 threads.  A workgroup reduces the previous FFN's partial sums for its rows
// and runs ONE of the three projections.  A wave's piece sequence: 32 rows of partial sums (rows 2w, 2w + 1 of the slab: exactly the rows
// its own lanes normalise, so no barrier), then the 32 weight pieces of its two column tiles.  (One workgroup per slab running q, k and v
// in turn was measured: 14.2 us — eight ring rounds in a row; three workgroups re-read the partial sums, 39 MB of L2 / Infinity Cache
// traffic per launch, and finish in four.)
// k and v are written in the ORDER dec_self_kernel's waves stream them, one KiB per MFMA operand:
//   kf [frame][head][key tile kt][lane 16 g + li][8]   = k[key 16 kt + li][32 head + 8 g ..]                (A operand of S^T = K Q^T)
//   vf [frame][head][kb][dt][lane 16 g + li][8]        = v[key kb*32 + {4g .. 4g+3, 16+4g .. 16+4g+3}][32 head + 16 dt + li]   (A operand of O^T = V^T P^T)
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int QKV_MAXS = 16;
template <bool PRO>
__global__ __launch_bounds__(512) void dec_qkv_kernel(DecQkvParams p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const Xhi = smem;
    unsigned char* const Xlo = smem + 16 * XP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15, lane16 = lane * 16;
    const int row0 = blockIdx.x * 16, part = blockIdx.y;
    constexpr int NP = PRO ? 32 : 0, TOTAL = NP + 32;
    PieceRing<SLAB_R, TOTAL> rg{smem + SLAB_RING + wave * SLAB_R * 1024, lane16};
    const unsigned char* const wsrc = reinterpret_cast<const unsigned char*>(p.w) + (size_t)(part * 16 + 2 * wave) * 16384;
    const int ra = row0 + 2 * wave < p.M ? row0 + 2 * wave : p.M - 1, rb = row0 + 2 * wave + 1 < p.M ? row0 + 2 * wave + 1 : p.M - 1;
    auto src = [&](const int n) -> const unsigned char* {
        if (PRO && n < NP) {   // partial slab s = n / 2, row 2w + (n & 1)
            const int s = n >> 1 < p.nsplit ? n >> 1 : p.nsplit - 1;
            return reinterpret_cast<const unsigned char*>(p.partials + ((size_t)s * p.M + ((n & 1) ? rb : ra)) * 256);
        }
        return wsrc + (size_t)(n - NP) * 1024;   // tiles 16 part + 2w, + 1 are consecutive 16-KiB blocks
    };
    const int row = row0 + li;
    const bool ok = row < p.M;
    const int rcl = ok ? row : p.M - 1;
    const int fr = rcl / p.Q, qi = rcl - fr * p.Q;
    float4v bs[2];
    {   // the slab's rows: 32 threads per row, 8 consecutive channels each; [previous FFN's sum + LN3]
        const int r = tid >> 5, c0 = (tid & 31) * 8, prow = row0 + r;
        const size_t rc = (size_t)(prow < p.M ? prow : p.M - 1) * 256 + c0;   // (rows beyond M: a valid address, results unused)
        const float* hsrc = PRO ? p.h_in : p.h_out;
        float4v xa = *reinterpret_cast<const float4v*>(hsrc + rc), xb = *reinterpret_cast<const float4v*>(hsrc + rc + 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) bs[j] = *reinterpret_cast<const float4v*>(p.bias + (size_t)qi * 768 + part * 256 + (2 * wave + j) * 16 + 4 * g);
        float v[8];
        if constexpr (PRO) {
            const float4v ba = *reinterpret_cast<const float4v*>(p.b2 + c0), bb = *reinterpret_cast<const float4v*>(p.b2 + c0 + 4);
            const float4v g0 = *reinterpret_cast<const float4v*>(p.ln_g + c0), g1 = *reinterpret_cast<const float4v*>(p.ln_g + c0 + 4);
            const float4v e0 = *reinterpret_cast<const float4v*>(p.ln_b + c0), e1 = *reinterpret_cast<const float4v*>(p.ln_b + c0 + 4);
            rg.prime(src);
            DEC_FENCE();
#pragma unroll
            for (int bt = 0; bt < 4; ++bt) {   // four batches of 4 slabs x 2 rows; fixed order: deterministic
                rg.wait(8 * bt, 8 * bt + 7);
                if (bt == 0) { xa += ba; xb += bb; }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const unsigned char* sl = rg.slot(8 * bt + 2 * s + (lane >> 5)) + (lane & 31) * 32;
                    const float4v pa = *reinterpret_cast<const float4v*>(sl), pb = *reinterpret_cast<const float4v*>(sl + 16);
                    if (4 * bt + s < p.nsplit) { xa += pa; xb += pb; }
                }
                rg.advance(src, 8 * bt, 8);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = xa[j]; v[4 + j] = xb[j]; }
            layernorm_row32(v, g0, g1, e0, e1);
            if (part == 0 && prow < p.M) {
                *reinterpret_cast<float4v*>(p.h_out + (size_t)prow * 256 + c0) = float4v{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<float4v*>(p.h_out + (size_t)prow * 256 + c0 + 4) = float4v{v[4], v[5], v[6], v[7]};
            }
        } else {
            rg.prime(src);
            DEC_FENCE();
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = xa[j]; v[4 + j] = xb[j]; }
        }
        store_split8(Xhi, Xlo, XP, r, c0, v);
    }
    DEC_LGKM0();
    __builtin_amdgcn_s_barrier();
    Acc3 a;
    acc3_zero(a);
    gemm256(rg, src, NP, Xhi, Xlo, lane16, g, li, a);
    if (!ok) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = (2 * wave + j) * 16 + 4 * g;   // channel inside this part: head c >> 5, dim c & 31 .. + 3
        half4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (_Float16)(acc3_get(a, j, r) + bs[j][r]);
        const int hd = c >> 5, d = c & 31;
        if (part == 0) {
            *reinterpret_cast<half4*>(p.q16 + (size_t)row * 256 + c) = o;
        } else if (part == 1) {   // kf: tile qi >> 4, lane 16 (d >> 3) + (qi & 15), elements d & 7 .. + 3
            _Float16* kf = reinterpret_cast<_Float16*>(p.k16) + ((size_t)((fr * 8 + hd) * 8 + (qi >> 4)) * 64 + (d >> 3) * 16 + (qi & 15)) * 8 + (d & 7);
            *reinterpret_cast<half4*>(kf) = o;
        } else {                  // vf: block (qi >> 5, d >> 4), lane 16 gk + (d & 15), element j = 4 * upper half + (key & 3)
            const int kk = qi & 31, hi16 = kk >> 4, gk = (kk & 15) >> 2, jj = hi16 * 4 + (kk & 3);
            _Float16* vf = reinterpret_cast<_Float16*>(p.vT) + ((size_t)(((fr * 8 + hd) * 4 + (qi >> 5)) * 2 + (d >> 4)) * 64 + gk * 16 + (d & 15)) * 8 + jj;
#pragma unroll
            for (int r = 0; r < 4; ++r) vf[(size_t)r * 8] = o[r];   // dims d .. d + 3 are consecutive lanes
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// dec_self_kernel: grid (ceil(Q / 16) query slabs, B frames), 512 threads: wave = head during the attention, then column tiles.
// Everything but the 16 query rows arrives through the wave's ring: K and V^T of its head (fragment order, written by dec_qkv_kernel),
// then its two tiles of Wo, then of Wq_c.  (K / V through fragment-shaped VGPR loads: 170 KiB per workgroup on the slow path, 3.5 us.)
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void dec_self_kernel(DecSelfParams p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const Xhi = smem;
    unsigned char* const Xlo = smem + 16 * XP;
    float (*red)[8][16] = reinterpret_cast<float (*)[8][16]>(smem + SLAB_RED);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15, lane16 = lane * 16;
    const int q0 = blockIdx.x * 16, b = blockIdx.y;
    const int Q = p.Q;
    const bool q_ok = q0 + li < Q;
    const size_t row = (size_t)b * Q + (q_ok ? q0 + li : Q - 1);   // this lane's data row (clamped: padding lanes compute on a valid row, store nothing)
    const int h = wave;
    unsigned long long* const tr = p.trace;
    unsigned long long dec_ts[8] = {};
    DEC_STAMP(tr, 0);
    PieceRing<SLAB_R, 80> rg{smem + SLAB_RING + wave * SLAB_R * 1024, lane16};
    const unsigned char* const wo = reinterpret_cast<const unsigned char*>(p.wo) + (size_t)(2 * wave) * 16384;
    const unsigned char* const wq = reinterpret_cast<const unsigned char*>(p.wq) + (size_t)(2 * wave) * 16384;
    const unsigned char* const kfs = reinterpret_cast<const unsigned char*>(p.k16) + (size_t)(b * 8 + h) * 8192;
    const unsigned char* const vfs = reinterpret_cast<const unsigned char*>(p.vT) + (size_t)(b * 8 + h) * 8192;
    // piece sequence: this head's 8 key tiles of K, its 8 blocks of V^T, this wave's two tiles of Wo (32), of Wq_c (32)
    auto src = [&](const int n) -> const unsigned char* {
        if (n < 8) return kfs + (size_t)n * 1024;
        if (n < 16) return vfs + (size_t)(n - 8) * 1024;
        return (n < 48 ? wo : wq) + (size_t)((n - 16) & 31) * 1024;
    };
    // ---- every load of the kernel that does not wait for computed data, requested at once ----------------------------------------------
    const half8 qf = *reinterpret_cast<const half8*>(p.q16 + row * 256 + h * 32 + g * 8);
    float4v bo[2], hres[2], gm[2], bt[2], bs[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = (2 * wave + j) * 16 + 4 * g;
        bo[j] = *reinterpret_cast<const float4v*>(p.bo + c);
        hres[j] = *reinterpret_cast<const float4v*>(p.h + row * 256 + c);
        gm[j] = *reinterpret_cast<const float4v*>(p.ln_g + c);
        bt[j] = *reinterpret_cast<const float4v*>(p.ln_b + c);
        bs[j] = *reinterpret_cast<const float4v*>(p.rbq + (size_t)(q_ok ? q0 + li : 0) * 256 + c);
    }
    rg.prime(src);
    DEC_FENCE();
    DEC_STAMP(tr, 1);
    // ---- attention of head `wave`: S^T = K Q^T over the frame's Q keys, softmax per query, O^T = V^T P^T -------------------------------
    float4v s[8];
    float mx = -INFINITY;
    rg.wait(0, 7);
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
        const half8 kf = *reinterpret_cast<const half8*>(rg.slot(kt) + lane16);   // (tiles / rows beyond Q: whatever the buffer holds, masked below)
        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf, float4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[kt][r] = kt * 16 + g * 4 + r < Q ? s[kt][r] : -INFINITY;
            mx = fmaxf(mx, s[kt][r]);
        }
    }
    rg.advance(src, 0, 8);
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
    DEC_STAMP(tr, 2);
    float4v oacc[2] = {float4v{0.f, 0.f, 0.f, 0.f}, float4v{0.f, 0.f, 0.f, 0.f}};
    rg.wait(8, 15);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        half8 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pf[r] = (_Float16)s[2 * kb][r]; pf[4 + r] = (_Float16)s[2 * kb + 1][r]; }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            half8 vf = *reinterpret_cast<const half8*>(rg.slot(8 + 2 * kb + dt) + lane16);
            // k-slots 0..3 = keys kb*32 + 4g .., 4..7 = keys kb*32 + 16 + 4g ..  (Q % 4 == 0: a group of 4 keys is all valid or all padding);
            // padding keys: whatever the buffer holds x 0 must stay 0
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                vf[r] = kb * 32 + g * 4 < Q ? vf[r] : (_Float16)0.f;
                vf[4 + r] = kb * 32 + 16 + g * 4 < Q ? vf[4 + r] : (_Float16)0.f;
            }
            oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, oacc[dt], 0, 0, 0);
        }
    }
    rg.advance(src, 8, 8);
    const float inv = 1.0f / lsum;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) store_split4(Xhi, Xlo, XP, li, h * 32 + dt * 16 + g * 4, oacc[dt] * inv);
    DEC_LGKM0();
    __builtin_amdgcn_s_barrier();
    DEC_STAMP(tr, 3);
    // ---- o-proj + residual + LN1 (Wq_c's first half-tiles are requested as Wo's slots drain: they travel during the LayerNorm) -------------
    Acc3 a;
    acc3_zero(a);
    gemm256(rg, src, 16, Xhi, Xlo, lane16, g, li, a);
    float4v hn[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) hn[j][r] = acc3_get(a, j, r) + (bo[j][r] + hres[j][r]);
    DEC_STAMP(tr, 4);
    layernorm_acc(hn, red, wave, g, li, gm, bt);
    DEC_STAMP(tr, 5);
    // (layernorm_acc's barriers lie between the o-proj's last fragment read and these writes)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = (2 * wave + j) * 16 + 4 * g;
        if (q_ok) *reinterpret_cast<float4v*>(p.h + row * 256 + c) = hn[j];
        store_split4(Xhi, Xlo, XP, li, c, hn[j]);
    }
    DEC_LGKM0();
    __builtin_amdgcn_s_barrier();
    // ---- q_c = h . Wq_c^T + (qpos . Wq_c^T + bq_c) ----------------------------------------------------------------------
    acc3_zero(a);
    DEC_STAMP(tr, 6);
    gemm256(rg, src, 48, Xhi, Xlo, lane16, g, li, a);
    DEC_STAMP(tr, 7);
    if (q_ok) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (p.qc_bf16) {   // (launch-uniform: the handle's operand type is bf16, and so are the K / V this query meets)
                typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
                bf4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (__bf16)(acc3_get(a, j, r) + bs[j][r]);
                *reinterpret_cast<bf4*>(p.qc16 + row * 256 + (2 * wave + j) * 16 + 4 * g) = o;
            } else {
                half4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (_Float16)(acc3_get(a, j, r) + bs[j][r]);
                *reinterpret_cast<half4*>(p.qc16 + row * 256 + (2 * wave + j) * 16 + 4 * g) = o;
            }
        }
    }
    DEC_STAMP_FLUSH(tr);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// dec_cross_out_kernel: grid (slabs of 16 rows), 512 threads.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int CROSS_MAXS = 6;
__global__ __launch_bounds__(512) void dec_cross_out_kernel(DecCrossOutParams p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const Xhi = smem;
    unsigned char* const Xlo = smem + 16 * XP;
    float (*red)[8][16] = reinterpret_cast<float (*)[8][16]>(smem + SLAB_RED);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15, lane16 = lane * 16;
    const int row0 = blockIdx.x * 16;
    const int row = row0 + li;
    const bool ok = row < p.M;
    const int rcl = ok ? row : p.M - 1;
    PieceRing<SLAB_R, 32> rg{smem + SLAB_RING + wave * SLAB_R * 1024, lane16};
    const unsigned char* const wo = reinterpret_cast<const unsigned char*>(p.wo) + (size_t)(2 * wave) * 16384;
    auto src = [&](const int n) -> const unsigned char* { return wo + (size_t)n * 1024; };
    float4v rs[2], gm[2], bt[2];
    const float* res = p.res + (size_t)(p.res_period > 0 ? rcl % p.res_period : rcl) * 256;
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
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = (2 * wave + j) * 16 + 4 * g;
            rs[j] = *reinterpret_cast<const float4v*>(p.bo + c) + *reinterpret_cast<const float4v*>(res + c);
            gm[j] = *reinterpret_cast<const float4v*>(p.ln_g + c);
            bt[j] = *reinterpret_cast<const float4v*>(p.ln_b + c);
        }
        rg.prime(src);
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
    DEC_LGKM0();
    __builtin_amdgcn_s_barrier();
    Acc3 a;
    acc3_zero(a);
    gemm256(rg, src, 0, Xhi, Xlo, lane16, g, li, a);
    float4v hn[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) hn[j][r] = acc3_get(a, j, r) + rs[j][r];
    layernorm_acc(hn, red, wave, g, li, gm, bt);
    if (ok) {
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<float4v*>(p.h + (size_t)row * 256 + (2 * wave + j) * 16 + 4 * g) = hn[j];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// dec_ffn_kernel: grid (slabs of 64 rows, F / 128 hidden chunks), 512 threads.
//   fc1: wave w owns hidden tile w of the chunk (16 channels) for the slab's four 16-row tiles; fc2: wave w owns output tiles 2w, 2w + 1.
// Piece sequence of a wave (ring of 11 slots): 8 rows of the slab (fp32: the wave converts rows 8w .. 8w + 7 to hi / lo), the 16 pieces
// of its W1 tile, then the 16 pieces of its two W2 tiles for this chunk's four k-steps, k-step major.  The hidden chunk (hi / lo)
// takes the place of the input rows in LDS.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int FFN_ROWS = 64;
constexpr int FFN_R = 11;
constexpr int FFN_RING = 2 * FFN_ROWS * XP;                 // 67 584 (1 KiB aligned)
constexpr int FFN_LDS = FFN_RING + 8 * FFN_R * 1024;        // 157 696

__global__ __launch_bounds__(512) void dec_ffn_kernel(DecFfnParams p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const Xhi = smem;
    unsigned char* const Xlo = smem + FFN_ROWS * XP;
    unsigned char* const Hhi = smem;                        // (after the barrier that ends fc1)
    unsigned char* const Hlo = smem + FFN_ROWS * HP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15, lane16 = lane * 16;
    const int row0 = blockIdx.x * FFN_ROWS, chunk = blockIdx.y;
    PieceRing<FFN_R, 40> rg{smem + FFN_RING + wave * FFN_R * 1024, lane16};
    const unsigned char* const w1 = reinterpret_cast<const unsigned char*>(p.w1) + (size_t)(chunk * 8 + wave) * 16384;
    const size_t w2tile = (size_t)(p.F / 32) * 2048;   // bytes of one 16-row tile of W2 (F / 32 k-steps x (hi, lo))
    const unsigned char* const w2 = reinterpret_cast<const unsigned char*>(p.w2) + (size_t)(2 * wave) * w2tile + (size_t)chunk * 4 * 2048;
    auto src = [&](const int n) -> const unsigned char* {
        if (n < 8) { const int r = row0 + 8 * wave + n; return reinterpret_cast<const unsigned char*>(p.h + (size_t)(r < p.M ? r : p.M - 1) * 256); }
        if (n < 24) return w1 + (size_t)(n - 8) * 1024;
        const int i = n - 24;   // k-step i / 4, tile (i / 2) & 1, hi / lo i & 1
        return w2 + (size_t)((i >> 1) & 1) * w2tile + (size_t)(i >> 2) * 2048 + (size_t)(i & 1) * 1024;
    };
    const float4v b1 = *reinterpret_cast<const float4v*>(p.b1 + chunk * OPD_DEC_FFN_CHUNK + wave * 16 + 4 * g);
    rg.prime(src);
    DEC_FENCE();
    // ---- rows 8w .. 8w + 7 -> hi / lo -----------------------------------------------------------------------------------------
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        rg.wait(n, n);
        float4v x = *reinterpret_cast<const float4v*>(rg.slot(n) + lane16);
        if (row0 + 8 * wave + n >= p.M) x = float4v{0.f, 0.f, 0.f, 0.f};
        store_split4(Xhi, Xlo, XP, 8 * wave + n, lane * 4, x);
        rg.advance(src, n, 1);
    }
    DEC_LGKM0();
    __builtin_amdgcn_s_barrier();
    // ---- hidden chunk = relu(x . W1^T + b1) ---------------------------------------------------------------------------------
    float4v ah[4], al[4], am[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) { ah[mt] = float4v{0.f, 0.f, 0.f, 0.f}; al[mt] = ah[mt]; am[mt] = ah[mt]; }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        rg.wait(8 + 2 * ks, 9 + 2 * ks);
        const half8 wh = *reinterpret_cast<const half8*>(rg.slot(8 + 2 * ks) + lane16), wl = *reinterpret_cast<const half8*>(rg.slot(9 + 2 * ks) + lane16);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const half8 xh = lds_frag(Xhi, XP, mt * 16 + li, ks, g), xl = lds_frag(Xlo, XP, mt * 16 + li, ks, g);
            mma3(wh, wl, xh, xl, ah[mt], al[mt], am[mt]);
        }
        rg.advance(src, 8 + 2 * ks, 2);
    }
    DEC_LGKM0();
    __builtin_amdgcn_s_barrier();   // every wave has read the input rows: the hidden chunk may overwrite them
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float4v hid;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { const float tv = ah[mt][rr] + (al[mt][rr] + am[mt][rr]) * LO_INV + b1[rr]; hid[rr] = tv > 0.f ? tv : 0.f; }
        store_split4(Hhi, Hlo, HP, mt * 16 + li, wave * 16 + 4 * g, hid);
    }
    DEC_LGKM0();
    __builtin_amdgcn_s_barrier();
    // ---- partial = hidden chunk . W2[:, chunk]^T ----------------------------------------------------------------------------
    Acc3 a[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc3_zero(a[mt]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        rg.wait(24 + 4 * ks, 27 + 4 * ks);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const half8 wh = *reinterpret_cast<const half8*>(rg.slot(24 + 4 * ks + 2 * j) + lane16), wl = *reinterpret_cast<const half8*>(rg.slot(25 + 4 * ks + 2 * j) + lane16);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const half8 xh = lds_frag(Hhi, HP, mt * 16 + li, ks, g), xl = lds_frag(Hlo, HP, mt * 16 + li, ks, g);
                mma3(wh, wl, xh, xl, a[mt].h[j], a[mt].l[j], a[mt].m[j]);
            }
        }
        rg.advance(src, 24 + 4 * ks, 4);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int row = row0 + mt * 16 + li;
        if (row < p.M) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float4v o;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) o[rr] = acc3_get(a[mt], j, rr);
                *reinterpret_cast<float4v*>(p.partials + ((size_t)chunk * p.M + row) * 256 + (2 * wave + j) * 16 + 4 * g) = o;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// heads2_kernel: the class head (256 -> ncls <= 128) and the box MLP (256 -> 256 -> 256 -> 4, ReLU, sigmoid) on split fp16 operands
// (HF:models/detr/modeling_detr.py:1284-1300, 1567-1583), grid = slabs of 16 rows, 512 threads.  The fp32-MFMA form (kernels_misc.hip::
// heads_kernel) pulls 620 KB of fp32 weights per workgroup through register loads in fragment shape -- the slow path of
// tools/microbench/oneshot.hip -- and takes 34 us; here the three 256-wide layers travel like every other decoder weight: fragment-order
// hi / lo pairs through the wave-private rings.  Prologue as heads_kernel's (the last layer's FFN sum + LN3, the final decoder LayerNorm).
// A wave's piece sequence: 16 pieces of its class tile (the class matrix is padded to 128 rows: waves 6, 7 multiply zeros), 32 of layer 1's
// two tiles, 32 of layer 2's.  The hidden activations pass through the slab's X area (barrier - rewrite - barrier); the last layer
// (256 -> 4) is 64 dot products split over the waves, in fp32 from LDS.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void heads2_kernel(HeadParams p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const Xhi = smem;
    unsigned char* const Xlo = smem + 16 * XP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15, lane16 = lane * 16;
    const int row0 = blockIdx.x * 16;
    const int row = row0 + li;
    const bool ok = row < p.rows;
    PieceRing<SLAB_R, 80> rg{smem + SLAB_RING + wave * SLAB_R * 1024, lane16};
    const unsigned char* const wc = reinterpret_cast<const unsigned char*>(p.wc_f) + (size_t)wave * 16384;
    const unsigned char* const w1 = reinterpret_cast<const unsigned char*>(p.w1_f) + (size_t)(2 * wave) * 16384;
    const unsigned char* const w2 = reinterpret_cast<const unsigned char*>(p.w2_f) + (size_t)(2 * wave) * 16384;
    auto src = [&](const int n) -> const unsigned char* { return n < 16 ? wc + (size_t)n * 1024 : n < 48 ? w1 + (size_t)(n - 16) * 1024 : w2 + (size_t)(n - 48) * 1024; };
    {   // rows -> X hi / lo: 32 threads per row, 8 consecutive columns each; every load that does not depend on computed data up front
        const int r = tid >> 5, c0 = (tid & 31) * 8;
        const size_t rc = (size_t)(row0 + r < p.rows ? row0 + r : p.rows - 1) * 256 + c0;
        float v[8];
        const float4v h0 = *reinterpret_cast<const float4v*>(p.hs + rc), h1 = *reinterpret_cast<const float4v*>(p.hs + rc + 4);
        float4v pa[16], pb[16];
        if (p.partials) {
#pragma unroll
            for (int sp = 0; sp < 16; ++sp) {
                const float* ps = p.partials + (size_t)(sp < p.nsplit ? sp : p.nsplit - 1) * p.rows * 256 + rc;
                pa[sp] = *reinterpret_cast<const float4v*>(ps);
                pb[sp] = *reinterpret_cast<const float4v*>(ps + 4);
            }
        }
        const float* z = p.ffn_b2 ? p.ffn_b2 : p.b1;   // (any valid address: unused without partials)
        const float4v fb0 = *reinterpret_cast<const float4v*>(z + c0), fb1 = *reinterpret_cast<const float4v*>(z + c0 + 4);
        const float* g3 = p.ln3_gamma ? p.ln3_gamma : p.b1;
        const float* b3 = p.ln3_beta ? p.ln3_beta : p.b1;
        const float4v g30 = *reinterpret_cast<const float4v*>(g3 + c0), g31 = *reinterpret_cast<const float4v*>(g3 + c0 + 4);
        const float4v b30 = *reinterpret_cast<const float4v*>(b3 + c0), b31 = *reinterpret_cast<const float4v*>(b3 + c0 + 4);
        const float* gf = p.ln_gamma ? p.ln_gamma : p.b1;
        const float* bf = p.ln_beta ? p.ln_beta : p.b1;
        const float4v gf0 = *reinterpret_cast<const float4v*>(gf + c0), gf1 = *reinterpret_cast<const float4v*>(gf + c0 + 4);
        const float4v bf0 = *reinterpret_cast<const float4v*>(bf + c0), bf1 = *reinterpret_cast<const float4v*>(bf + c0 + 4);
        rg.prime(src);
        DEC_FENCE();
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = h0[j]; v[4 + j] = h1[j]; }
        if (p.partials) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] += fb0[j]; v[4 + j] += fb1[j]; }
#pragma unroll
            for (int sp = 0; sp < 16; ++sp)
                if (sp < p.nsplit) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] += pa[sp][j]; v[4 + j] += pb[sp][j]; }
                }
            layernorm_row32(v, g30, g31, b30, b31);
        }
        if (p.ln_gamma) layernorm_row32(v, gf0, gf1, bf0, bf1);
        if (row0 + r >= p.rows) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
        store_split8(Xhi, Xlo, XP, r, c0, v);
    }
    DEC_LGKM0();
    __builtin_amdgcn_s_barrier();
    // ---- class logits: this wave's tile (columns 16 w .. + 15) -------------------------------------------------------------------------
    {
        Acc3 c;
        acc3_zero(c);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            rg.wait(8 * hf, 8 * hf + 7);
            gemm_group(rg, 8 * hf, Xhi, Xlo, li, 4 * hf, lane16, g, c.h[0], c.l[0], c.m[0]);
            rg.advance(src, 8 * hf, 8);
        }
        const int n = wave * 16 + 4 * g;
        if (ok) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < p.ncls) p.logits[(size_t)row * p.ncls + n + r] = acc3_get(c, 0, r) + p.bc[n + r];
        }
    }
    // ---- box MLP layer 1 ------------------------------------------------------------------------------------------------------------------
    Acc3 a;
    acc3_zero(a);
    gemm256(rg, src, 16, Xhi, Xlo, lane16, g, li, a);
    float4v t[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float4v b = *reinterpret_cast<const float4v*>(p.b1 + (2 * wave + j) * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float x = acc3_get(a, j, r) + b[r]; t[j][r] = x > 0.f ? x : 0.f; }
    }
    DEC_LGKM0();
    __builtin_amdgcn_s_barrier();          // every wave has read the rows: the first hidden layer takes their place
#pragma unroll
    for (int j = 0; j < 2; ++j) store_split4(Xhi, Xlo, XP, li, (2 * wave + j) * 16 + 4 * g, t[j]);
    DEC_LGKM0();
    __builtin_amdgcn_s_barrier();
    // ---- layer 2 ---------------------------------------------------------------------------------------------------------------------------
    acc3_zero(a);
    gemm256(rg, src, 48, Xhi, Xlo, lane16, g, li, a);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float4v b = *reinterpret_cast<const float4v*>(p.b2 + (2 * wave + j) * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float x = acc3_get(a, j, r) + b[r]; t[j][r] = x > 0.f ? x : 0.f; }
    }
    DEC_LGKM0();
    __builtin_amdgcn_s_barrier();          // the X area is free: the second hidden layer as fp32 [16][264]
    float* const h2 = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < 2; ++j) *reinterpret_cast<float4v*>(h2 + li * 264 + (2 * wave + j) * 16 + 4 * g) = t[j];
    DEC_LGKM0();
    __builtin_amdgcn_s_barrier();
    // ---- last layer: 64 outputs (row = lane >> 2, coordinate = lane & 3), the reduction split over the 8 waves (32 k each), summed in order ----
    float* const part = reinterpret_cast<float*>(smem + SLAB_RING);   // (the rings are drained: every piece has been consumed)
    {
        const int r = lane >> 2, c = lane & 3;
        float acc = 0.f;
#pragma unroll 8
        for (int kk = 0; kk < 32; ++kk) acc = fmaf(h2[r * 264 + wave * 32 + kk], p.w3[(wave * 32 + kk) * 4 + c], acc);
        part[wave * 64 + lane] = acc;
    }
    DEC_LGKM0();
    __builtin_amdgcn_s_barrier();
    if (tid < 64) {
        const int r = tid >> 2, c = tid & 3;
        float acc = part[tid];
#pragma unroll
        for (int w = 1; w < 8; ++w) acc += part[w * 64 + tid];
        if (row0 + r < p.rows) p.boxes[(size_t)(row0 + r) * 4 + c] = 1.0f / (1.0f + expf(-(acc + p.b3[c])));
    }
}

}  // namespace

// host: [N][K] fp32 -> the split pair in MFMA-fragment order: for every (16-row tile nt, 32-wide k-step ks) 1 KiB of hi = fp16(w) followed by
// 1 KiB of lo = fp16((w - hi) * 2048); inside a KiB lane L = 16 g + li holds w[16 nt + li][32 ks + 8 g .. + 7].  out: 2 * N * K halves.
void opd_split_f16_frag(const float* w, int N, int K, f16_t* out) {
    const int nks = K / 32;
    for (int nt = 0; nt < N / 16; ++nt)
        for (int ks = 0; ks < nks; ++ks) {
            f16_t* blk = out + ((size_t)nt * nks + ks) * 1024;
            for (int L = 0; L < 64; ++L)
                for (int e = 0; e < 8; ++e) {
                    const float v = w[(size_t)(nt * 16 + (L & 15)) * K + ks * 32 + (L >> 4) * 8 + e];
                    const _Float16 h = fabsf(v) < 6.103515625e-5f ? (_Float16)0.f : (_Float16)v;   // (no subnormal hi: see split1)
                    const _Float16 l = (_Float16)((v - (float)h) * 2048.0f);
                    __builtin_memcpy(&blk[L * 8 + e], &h, 2);
                    __builtin_memcpy(&blk[512 + L * 8 + e], &l, 2);
                }
        }
}

hipError_t opd_launch_heads2(const HeadParams& p, hipStream_t stream) {
    if (p.rows <= 0 || p.ncls <= 0 || p.ncls > 128 || !p.hs || !p.wc_f || !p.w1_f || !p.w2_f || !p.bc || !p.b1 || !p.b2 || !p.w3 || !p.b3 || !p.logits || !p.boxes)
        return hipErrorInvalidValue;
    if (p.partials && (p.nsplit < 1 || p.nsplit > 16 || !p.ffn_b2 || !p.ln3_gamma || !p.ln3_beta)) return hipErrorInvalidValue;
    if ((p.ln_gamma != nullptr) != (p.ln_beta != nullptr)) return hipErrorInvalidValue;
    OPD_SET_MAX_LDS_ONCE(heads2_kernel, SLAB_LDS);
    OPD_LAUNCH(heads2_kernel, dim3((p.rows + 15) / 16), dim3(512), SLAB_LDS, stream, p);
    return hipGetLastError();
}
hipError_t opd_launch_dec_qkv(const DecQkvParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.Q <= 0 || !p.h_out || !p.w || !p.bias || !p.q16 || !p.k16 || !p.vT) return hipErrorInvalidValue;
    if (p.partials && (!p.h_in || !p.b2 || !p.ln_g || !p.ln_b || p.nsplit < 1 || p.nsplit > QKV_MAXS)) return hipErrorInvalidValue;
    if (p.Q > 128 || p.M % p.Q != 0) return hipErrorInvalidValue;   // v^T rows hold 128 keys; rows are (frame, query)
    OPD_SET_MAX_LDS_ONCE(dec_qkv_kernel<true>, SLAB_LDS);
    OPD_SET_MAX_LDS_ONCE(dec_qkv_kernel<false>, SLAB_LDS);
    if (p.partials) OPD_LAUNCH(dec_qkv_kernel<true>, dim3((p.M + 15) / 16, 3), dim3(512), SLAB_LDS, stream, p);
    else OPD_LAUNCH(dec_qkv_kernel<false>, dim3((p.M + 15) / 16, 3), dim3(512), SLAB_LDS, stream, p);
    return hipGetLastError();
}
hipError_t opd_launch_dec_self(const DecSelfParams& p, hipStream_t stream) {
    if (p.B <= 0 || p.Q <= 0 || p.Q > 128 || (p.Q & 3) || !p.q16 || !p.k16 || !p.vT || !p.h || !p.wo || !p.bo || !p.ln_g || !p.ln_b || !p.wq || !p.rbq || !p.qc16)
        return hipErrorInvalidValue;
    OPD_SET_MAX_LDS_ONCE(dec_self_kernel, SLAB_LDS);
    OPD_LAUNCH(dec_self_kernel, dim3((p.Q + 15) / 16, p.B), dim3(512), SLAB_LDS, stream, p);
    return hipGetLastError();
}
hipError_t opd_launch_dec_cross_out(const DecCrossOutParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.splits < 1 || p.splits > CROSS_MAXS || !p.part_o || !p.part_ml || !p.res || !p.h || !p.wo || !p.bo || !p.ln_g || !p.ln_b || p.res_period < 0)
        return hipErrorInvalidValue;
    OPD_SET_MAX_LDS_ONCE(dec_cross_out_kernel, SLAB_LDS);
    OPD_LAUNCH(dec_cross_out_kernel, dim3((p.M + 15) / 16), dim3(512), SLAB_LDS, stream, p);
    return hipGetLastError();
}
hipError_t opd_launch_dec_ffn(const DecFfnParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.F <= 0 || p.F % OPD_DEC_FFN_CHUNK != 0 || !p.h || !p.w1 || !p.b1 || !p.w2 || !p.partials) return hipErrorInvalidValue;
    OPD_SET_MAX_LDS_ONCE(dec_ffn_kernel, FFN_LDS);
    OPD_LAUNCH(dec_ffn_kernel, dim3((p.M + FFN_ROWS - 1) / FFN_ROWS, p.F / OPD_DEC_FFN_CHUNK), dim3(512), FFN_LDS, stream, p);
    return hipGetLastError();
}
