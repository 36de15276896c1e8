// kernels_attn.hip — flash-style multi-head attention, head_dim 32, fp16 MFMA / fp32 softmax, gfx950.
//
// softmax(Q K^T * 32^-1/2) V for the encoder self-attention (1050x1050), decoder
// self-attention (100x100) and decoder cross-attention (100x1050)   (HF:models/detr/modeling_detr.py:402-427).
// Scores are never materialised in HBM (the reference's eager path writes [B,8,1050,1050] fp32 = 282 MB).
//
// One workgroup = 4 waves = 64 queries of one (batch, head); each wave owns 16 queries.  Both products run
// "swapped" so that every per-query quantity is lane-local (query index = lane & 15):
//   S^T[key][q]  = mfma_16x16x32(A = K tile rows, B = Q rows)      -> lane holds 4 keys x its query per 16-key tile
//   O^T[d][q]   += mfma_16x16x32(A = V^T,         B = P^T)         -> lane holds 4 head-dims x its query
// The P accumulator tile is re-used directly as the B operand of the second product (k-slot j<4 -> S-tile 2kb,
// j>=4 -> S-tile 2kb+1); V^T fragments come from the V tile in LDS through ds_read_b64_tr_b16 (hardware transpose).
//
// The loop is bound by vector-instruction issue (one v_exp_f32 per score and what surrounds it), not by the matrix pipe or by
// memory (tools/trace_attn.py: ~520 SIMD cycles per wave and 64-key tile, 160 of them MFMA), so the round-3 form removes
// instructions from the common tile:
//  * Lazy rescale.  The exponent reference m_ref of a query is NOT the running maximum: it only moves when a tile holds a score
//    more than LAZY_T (2^8) above it.  p = exp2(s*c - m_ref) may then exceed 1 (up to 2^8: far inside fp16 / fp32 range), the
//    softmax ratios are unchanged (a common reference cancels in O = sum p v / sum p), and the common tile carries no cross-lane
//    maximum, no alpha = exp2(m_old - m_new), no rescale of the accumulators: one wave-uniform branch skips them.
//  * The row sum comes out of the matrix pipe: a third MFMA per 32 keys with an all-ones A operand accumulates sum_k P[k][q] — of
//    the fp16-rounded weights the PV product actually uses, so the normalised weights sum to one exactly — instead of eight packed
//    adds; every lane ends with its query's complete sum, no cross-lane reduction.
//  * K / V tiles travel HBM -> LDS by LDS-DMA (buffer_load ... lds: 1 KiB = 16 key rows x 64 bytes per wave-instruction): no
//    staging registers, no ds_write, no predicates (rows past the operand read as zeros through the buffer descriptor).
//    An LDS-DMA writes lane L's 16 bytes to slot L of its 1-KiB block, so the bank-conflict-free images are made on the SOURCE
//    side — lane L fetches the (row, 16-byte chunk) cell that belongs in slot L:
//      K block (ds_read_b128 fragment reads, lane (g, li) takes row li, chunk g):        slot = 16 * chunk + row
//      V block (ds_read_b64_tr_b16, 32 lanes take 8 rows x 32 bytes):   slot = 32 * (chunk >> 1) + 16 * (row >> 3) + 2 * (row & 7) + (chunk & 1)
// Encoder launch at batch 8 (1088 workgroups): 30.1 -> 24.0 us; 2040 tokens (r101 at 1066x1920): 91 -> 72 us (profiles/r03_bench_attn.txt).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <math.h>
#include "opd_kernels.h"
#include "opd_elem.h"

typedef elem_t half8 __attribute__((ext_vector_type(8)));
typedef elem_t half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef short short4v __attribute__((__vector_size__(4 * sizeof(short))));

namespace {

constexpr float LAZY_T = 8.0f;   // log2 of the head-room a score may have over its query's exponent reference before the reference moves

__device__ __forceinline__ half4 lds_tr16(const unsigned char* p) {
    short4v t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) short4v*)(const_cast<unsigned char*>(p)));
    half4 r;
    __builtin_memcpy(&r, &t, 8);
    return r;
}

// max over lanes {l, l^16, l^32, l^48}: v_permlane16_swap / v_permlane32_swap exchange 16- / 32-lane rows between two registers, so
// each butterfly step is one swap + one v_max in the VALU (ds_bpermute, what __shfl_xor compiles to, is an LDS round trip)
__device__ __forceinline__ float xmax16_32(const float v) {
    // (inline asm: given the same value for both operands, hipcc folds max over the two results of __builtin_amdgcn_permlane16_swap to
    //  the first one, as if they were equal; s_nop 1 = the VALU-write -> permlane-read hazard the compiler would otherwise cover)
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1" : "+v"(a), "+v"(b));
    float c = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1" : "+v"(a), "+v"(c));
    return a;
}

// TRACE (tools/trace_attn.py only): wave 0 stamps the shader clock at five points of every key tile and writes the per-phase sums.
#define ATTN_STAMP(i)                                                                                        \
    do {                                                                                                     \
        if constexpr (TRACE) {                                                                               \
            unsigned long long now_;                                                                         \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");                     \
            tsum[i] += now_ - tlast;                                                                         \
            tlast = now_;                                                                                    \
        }                                                                                                    \
    } while (0)

// MASKED: per-frame key mask of a ragged batch (compile time: the unmasked loop carries no mask arithmetic).
// KT: keys per LDS tile.  64 for the encoder (many workgroups per CU: the smaller tile keeps registers low); 128 for the decoder's
// cross-attention (<= 128 queries: a handful of workgroups, each a latency chain of per-tile barriers).
// SPLIT (the fused decoder's cross-attention, kernels_dec.hip): the key tiles are cut into p.splits contiguous ranges, one workgroup each;
// a workgroup writes its unnormalised sum_k p v, its exponent reference and its sum_k p instead of the normalised output, and
// dec_cross_out_kernel combines the splits.  A 100 x 1050 cross-attention is a latency chain of nine 128-key tiles on 128 workgroups;
// three splits make it three tiles on 384.
template <bool MASKED, int KT, bool TRACE = false, bool SPLIT = false>
__global__ __launch_bounds__(256) void attention_kernel(AttnParams p) {
    constexpr int NKT = KT / 16;              // 16-key score tiles per LDS tile
    constexpr int RPT = KT / 64;              // 1-KiB blocks per wave, operand and tile
    constexpr int K_BYTES = KT * 64, TILE_BYTES = KT * 128;   // one buffer = K tile then V tile, 64 bytes per key row each
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * TILE_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;
    const int li = lane & 15;
    // XCD-aware tile map: workgroup L runs on XCD L % 8, so XCD x takes the x-th contiguous eighth of the (frame, head, query
    // tile) order.  The query tiles of one frame then share one L2: its K/V rows (all heads of a key share 128-byte lines) come
    // from HBM once instead of once per XCD.
    const int nq = (p.Lq + 63) >> 6;
    const int nsp = SPLIT ? p.splits : 1;
    const int total = nq * p.heads * p.B * nsp;
    const int chunk = (total + 7) >> 3;
    const int n0 = (int)(blockIdx.x & 7u) * chunk + (int)(blockIdx.x >> 3);
    if (n0 >= total) return;   // whole workgroup leaves before the first barrier
    const int split = SPLIT ? n0 % nsp : 0, n = SPLIT ? n0 / nsp : n0;   // (the splits of one query tile are neighbours: same XCD)
    const int b = n / (nq * p.heads), h = (n / nq) % p.heads;
    const int q = (n % nq) * 64 + wave * 16 + li;
    const bool q_ok = q < p.Lq;

    half8 qf;
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[j] = (elem_t)0.f;
    if (q_ok) qf = *reinterpret_cast<const half8*>(p.q + ((size_t)b * p.Lq + q) * p.ldq + h * 32 + g * 8);
    half8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (elem_t)1.f;

    // LDS-DMA staging: wave w moves key rows 16 w .. 16 w + 15 (+ 64 i) of the tile.  Rows past Lk of the LAST frame fall outside
    // the descriptor (zeros); of earlier frames they are the next frame's rows: finite, and masked like every key >= Lk.
    const __amdgpu_buffer_rsrc_t rsrc_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.k), 0, (unsigned)((size_t)p.B * p.Lk * p.ldk * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.v), 0, (unsigned)((size_t)p.B * p.Lk * p.ldv * 2), 0x00020000);
    const int vr = ((lane >> 4) & 1) * 8 + ((lane >> 1) & 7), vc = (lane >> 5) * 2 + (lane & 1);   // the V cell of this lane's slot
    const unsigned koff = (unsigned)(((size_t)b * p.Lk + wave * 16 + (lane & 15)) * p.ldk + h * 32 + (lane >> 4) * 8) * 2u;
    const unsigned voff = (unsigned)(((size_t)b * p.Lk + wave * 16 + vr) * p.ldv + h * 32 + vc * 8) * 2u;
    auto dma_tile = [&](int t, int buf) {
        unsigned char* base = lds + buf * TILE_BYTES;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int row0 = t * KT + i * 64;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_k, (__attribute__((address_space(3))) void*)(base + (i * 4 + wave) * 1024), 16, koff,
                                                     (unsigned)(row0 * p.ldk) * 2u, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_v, (__attribute__((address_space(3))) void*)(base + K_BYTES + (i * 4 + wave) * 1024), 16, voff,
                                                     (unsigned)(row0 * p.ldv) * 2u, 0, 0);
        }
    };

    unsigned long long tsum[5] = {0, 0, 0, 0, 0}, tlast = 0, tbegin = 0, rbegin = 0;
    if constexpr (TRACE) {
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rbegin)::"memory");   // 100 MHz wall clock
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tbegin)::"memory");
        tlast = tbegin;
    }
    float m_ref = -INFINITY;   // exponent reference of this lane's query (log2 domain, uniform over the 4 lanes that share the query)
    float4v oacc[2] = {float4v{0.f, 0.f, 0.f, 0.f}, float4v{0.f, 0.f, 0.f, 0.f}};
    float4v lacc = float4v{0.f, 0.f, 0.f, 0.f};   // every element: sum over all keys so far of this query's fp16 weights

    // ragged batch: keys outside the frame's valid (rows x cols) rectangle of the key map get -inf, like the additive
    // attention mask of the reference (HF:models/detr/modeling_detr.py:402-427, 933-991); key 0 is always valid
    const int kv_rows = MASKED ? p.key_valid[2 * b] : 0, kv_cols = MASKED ? p.key_valid[2 * b + 1] : 0;
    const int ntiles = (p.Lk + KT - 1) / KT;
    const int tps = SPLIT ? (ntiles + nsp - 1) / nsp : ntiles;                 // key tiles per split
    const int t0 = split * tps, t1 = SPLIT ? (t0 + tps < ntiles ? t0 + tps : ntiles) : ntiles;
    const float scale2 = p.scale * 1.44269504088896340736f;  // exponents are taken in base 2: scores are multiplied by scale * log2(e)
    if (SPLIT && t0 >= t1) {   // more splits than tiles: an empty range carries no weight (workgroup-uniform: nobody reaches a barrier)
        if (q_ok) {
            const size_t prow = (size_t)split * p.B * p.Lq + (size_t)b * p.Lq + q;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) *reinterpret_cast<float4v*>(p.part_o + prow * (p.heads * 32) + h * 32 + dt * 16 + g * 4) = float4v{0.f, 0.f, 0.f, 0.f};
            if (g == 0) *reinterpret_cast<float2v*>(p.part_ml + (prow * p.heads + h) * 2) = float2v{-INFINITY, 0.f};
        }
        return;
    }
    dma_tile(t0, 0);
    __syncthreads();   // (an LDS-DMA in flight is a pending LDS write: the barrier's fence waits for it)

    // One key tile.  LAST (compile time): the tile may hold keys >= Lk; every other tile of an unmasked launch is full and carries no
    // masking code at all.
    auto tile_step = [&](const int t, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int buf = (t - t0) & 1;
        if (!LAST && (!SPLIT || t + 1 < t1)) dma_tile(t + 1, buf ^ 1);
        ATTN_STAMP(0);
        const unsigned char* Kl = lds + buf * TILE_BYTES;
        const unsigned char* Vl = Kl + K_BYTES;

        // ---- S^T = K Q^T ----------------------------------------------------------------------------------------
        float4v s[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const half8 kf = *reinterpret_cast<const half8*>(Kl + kt * 1024 + lane * 16);
            s[kt] = OPD_MFMA_16x16x32(kf, qf, float4v{0.f, 0.f, 0.f, 0.f});
        }
        float mx = -INFINITY;
        if (MASKED) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = t * KT + kt * 16 + g * 4 + r;
                    const int kr = key / p.key_row, kc = key - kr * p.key_row;
                    const bool ok = key < p.Lk && kr < kv_rows && kc < kv_cols;
                    s[kt][r] = ok ? s[kt][r] : -INFINITY;
                    mx = fmaxf(mx, s[kt][r]);
                }
        } else if (!LAST) {  // full tile: no key masking needed
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
        } else {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = t * KT + kt * 16 + g * 4 + r;
                    s[kt][r] = key < p.Lk ? s[kt][r] : -INFINITY;
                    mx = fmaxf(mx, s[kt][r]);
                }
        }
        // ---- does some query of this wave see a score more than 2^LAZY_T above its reference?  (tile 0: m_ref = -inf, always) --
        if (__any(mx * scale2 > m_ref + LAZY_T)) {
            const float mq = xmax16_32(mx) * scale2;   // the query's exact tile maximum (uniform over its 4 lanes)
            const bool up = mq > m_ref + LAZY_T;
            const float m_new = up ? mq : m_ref;
            const float alpha = up ? __builtin_amdgcn_exp2f(m_ref - m_new) : 1.0f;   // m_ref = -inf: exp2(-inf) = 0, on sums that are zero
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) oacc[dt][r] *= alpha;
#pragma unroll
            for (int r = 0; r < 4; ++r) lacc[r] *= alpha;
            m_ref = m_new;
        }
        ATTN_STAMP(1);
        // ---- p = exp2(s * scale * log2(e) - m_ref): one packed fma per two scores, one v_exp_f32 per score ---------------
        // (a split of a masked frame may hold no valid key at all: its reference stays -inf and every weight must come out as 0, not NaN)
        const float mn = (SPLIT && MASKED && m_ref == -INFINITY) ? 0.f : -m_ref;
        const float2v sc2 = {scale2, scale2}, mneg = {mn, mn};
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const float2v a = __builtin_elementwise_fma(float2v{s[kt][r], s[kt][r + 1]}, sc2, mneg);   // masked: -inf -> p = 0
                s[kt][r] = __builtin_amdgcn_exp2f(a[0]);
                s[kt][r + 1] = __builtin_amdgcn_exp2f(a[1]);
            }
        // ---- O^T += V^T P^T,  l += 1^T P^T ----------------------------------------------------------------------
#pragma unroll
        for (int kb = 0; kb < NKT / 2; ++kb) {
            half8 pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[r] = (elem_t)s[2 * kb][r];
                pf[4 + r] = (elem_t)s[2 * kb + 1][r];
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                // 16 lanes take 4 key rows x 16 head dims: rows kb*32 + g*4 + (li >> 2) (lo) and + 16 (hi: the next 1-KiB block)
                const unsigned char* a0 = Vl + kb * 2048 + (dt * 32 + (g >> 1) * 16 + ((g & 1) * 4 + (li >> 2)) * 2 + ((li >> 1) & 1)) * 16 + (li & 1) * 8;
                const half4 lo = lds_tr16(a0);
                const half4 hi = lds_tr16(a0 + 1024);
                half8 vf;
#pragma unroll
                for (int r = 0; r < 4; ++r) { vf[r] = lo[r]; vf[4 + r] = hi[r]; }
                oacc[dt] = OPD_MFMA_16x16x32(vf, pf, oacc[dt]);
            }
            lacc = OPD_MFMA_16x16x32(ones, pf, lacc);
        }
        ATTN_STAMP(2);
        if (TRACE && !LAST) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        ATTN_STAMP(3);
        __syncthreads();
        ATTN_STAMP(4);
    };
    if constexpr (SPLIT) {   // only the last tile of the whole key range can hold keys >= Lk
        for (int t = t0; t < t1; ++t) {
            if (t + 1 == ntiles) tile_step(t, std::true_type{});
            else tile_step(t, std::false_type{});
        }
    } else {
        for (int t = 0; t + 1 < ntiles; ++t) tile_step(t, std::false_type{});
        tile_step(ntiles - 1, std::true_type{});
    }

    if constexpr (TRACE) {
        if (tid == 0 && p.trace) {
            unsigned long long rend;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rend)::"memory");
            unsigned long long* tr = p.trace + (size_t)blockIdx.x * 12;
            for (int i = 0; i < 5; ++i) tr[i] = tsum[i];
            tr[5] = 0;
            tr[6] = tlast - tbegin;
            tr[7] = (unsigned long long)ntiles;
            tr[8] = rbegin; tr[9] = rend;
            tr[10] = (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID bits 0..3
            tr[11] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
        }
    }
    if constexpr (SPLIT) {
        if (q_ok) {
            const size_t prow = (size_t)split * p.B * p.Lq + (size_t)b * p.Lq + q;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) *reinterpret_cast<float4v*>(p.part_o + prow * (p.heads * 32) + h * 32 + dt * 16 + g * 4) = oacc[dt];
            if (g == 0) *reinterpret_cast<float2v*>(p.part_ml + (prow * p.heads + h) * 2) = float2v{m_ref, lacc[0]};
        }
        return;
    }
    if (q_ok) {
        const float inv = 1.0f / lacc[0];
        f16_t* orow = p.o + ((size_t)b * p.Lq + q) * p.ldo + h * 32 + g * 4;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            half4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (elem_t)(oacc[dt][r] * inv);
            *reinterpret_cast<half4*>(orow + dt * 16) = o;
        }
    }
}

}  // namespace

hipError_t OPD_SYM(opd_launch_attention)(const AttnParams& p, hipStream_t stream) {
    if (p.B <= 0 || p.heads <= 0 || p.Lq <= 0 || p.Lk <= 0) return hipErrorInvalidValue;
    if (p.splits > 0) {   // key-split partials (fused decoder cross-attention): 128-key tiles
        if (!p.part_o || !p.part_ml || p.trace || (p.ldq % 8) || (p.ldk % 8) || (p.ldv % 8) || (p.key_valid && p.key_row < 1)) return hipErrorInvalidValue;
        if ((size_t)p.B * p.Lk * p.ldk * 2 >= (1ull << 32) || (size_t)p.B * p.Lk * p.ldv * 2 >= (1ull << 32)) return hipErrorInvalidValue;
        const int total = ((p.Lq + 63) / 64) * p.heads * p.B * p.splits;
        dim3 grid(8 * ((total + 7) / 8));
        if (p.key_valid) OPD_LAUNCH((attention_kernel<true, 128, false, true>), grid, dim3(256), 0, stream, p);
        else OPD_LAUNCH((attention_kernel<false, 128, false, true>), grid, dim3(256), 0, stream, p);
        return hipGetLastError();
    }
    if ((p.ldq % 8) || (p.ldk % 8) || (p.ldv % 8) || (p.ldo % 4)) return hipErrorInvalidValue;  // 16-byte row chunks
    if (p.key_valid && p.key_row < 1) return hipErrorInvalidValue;
    if ((size_t)p.B * p.Lk * p.ldk * 2 >= (1ull << 32) || (size_t)p.B * p.Lk * p.ldv * 2 >= (1ull << 32)) return hipErrorInvalidValue;   // buffer descriptors
    const int total = ((p.Lq + 63) / 64) * p.heads * p.B;
    dim3 grid(8 * ((total + 7) / 8));   // 8 XCDs x their share of the tiles (attention_kernel's tile map)
    const bool wide = p.Lq <= 128 && p.Lk > 128;   // few query tiles, long key loop: decoder cross-attention
    if (p.trace) {
        OPD_LAUNCH((attention_kernel<false, 64, true>), grid, dim3(256), 0, stream, p);
    } else if (p.key_valid) {
        if (wide) OPD_LAUNCH((attention_kernel<true, 128>), grid, dim3(256), 0, stream, p);
        else OPD_LAUNCH((attention_kernel<true, 64>), grid, dim3(256), 0, stream, p);
    } else {
        if (wide) OPD_LAUNCH((attention_kernel<false, 128>), grid, dim3(256), 0, stream, p);
        else OPD_LAUNCH((attention_kernel<false, 64>), grid, dim3(256), 0, stream, p);
    }
    return hipGetLastError();
}
