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
// j>=4 -> S-tile 2kb+1); V^T fragments come from the row-major V tile in LDS through ds_read_b64_tr_b16
// (hardware transpose), with a scalar-gather cross-check path selectable at run time.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <math.h>
#include "opd_kernels.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef short short4v __attribute__((__vector_size__(4 * sizeof(short))));

namespace {

constexpr int LDS_ROW = 80;  // bytes per K row in LDS (64 payload + 16 pad): 16-byte fragment reads of 16 rows hit 16 distinct slots
// V rows are read 4 rows x 32 bytes per 16 lanes by ds_read_b64_tr_b16: a 96-byte stride tiles those four pieces over the
// 128-byte bank line exactly (80 bytes leaves rows 0 and 3 overlapping); measured: attention -1 %, the loop is not LDS bound
#ifndef OPD_ATTN_V_ROW
#define OPD_ATTN_V_ROW 96
#endif
constexpr int V_ROW = OPD_ATTN_V_ROW;

__device__ __forceinline__ half4 lds_tr16(const unsigned char* p) {
    short4v t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) short4v*)(const_cast<unsigned char*>(p)));
    half4 r;
    __builtin_memcpy(&r, &t, 8);
    return r;
}

// TR: V^T fragments through ds_read_b64_tr_b16 (hardware transpose) or scalar LDS gathers (cross-check path).
// MASKED: per-frame key mask of a ragged batch.  Both are compile-time so that the inner loop carries no branches.
// KT: keys per LDS tile.  64 for the encoder (VALU bound, many workgroups per CU: the smaller tile keeps registers low);
// 128 for the decoder (<= 128 queries: a handful of workgroups, each a latency chain of per-tile barriers: 14.6 -> 11.8 us).
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

template <bool TR, bool MASKED, int KT>
__global__ __launch_bounds__(256) void attention_kernel(AttnParams p) {
    constexpr int NKT = KT / 16;              // 16-key score tiles per LDS tile
    constexpr int K_BYTES = KT * LDS_ROW, TILE_BYTES = KT * (LDS_ROW + V_ROW);   // one buffer = K tile then V tile
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * TILE_BYTES];  // [buf][K|V]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int g = lane >> 4;
    const int li = lane & 15;
    // XCD-aware tile map: workgroup L runs on XCD L % 8, so XCD x takes the x-th contiguous eighth of the (frame, head, query
    // tile) order.  The query tiles of one frame then share one L2: its K/V rows (all heads of a key share 128-byte lines) come
    // from HBM once instead of once per XCD.
    const int nq = (p.Lq + 63) >> 6;
    const int total = nq * p.heads * p.B;
    const int chunk = (total + 7) >> 3;
    const int n = (int)(blockIdx.x & 7u) * chunk + (int)(blockIdx.x >> 3);
    if (n >= total) return;   // whole workgroup leaves before the first barrier
    const int b = n / (nq * p.heads), h = (n / nq) % p.heads;
    const int q = (n % nq) * 64 + wave * 16 + li;
    const bool q_ok = q < p.Lq;

    half8 qf;
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[j] = (_Float16)0.f;
    if (q_ok) qf = *reinterpret_cast<const half8*>(p.q + ((size_t)b * p.Lq + q) * p.ldq + h * 32 + g * 8);

    const int skey = tid >> 2, schunk = tid & 3;
    const f16_t* kbase = p.k + (size_t)b * p.Lk * p.ldk + h * 32 + schunk * 8;
    const f16_t* vbase = p.v + (size_t)b * p.Lk * p.ldv + h * 32 + schunk * 8;
    constexpr int RPT = KT / 64;   // key rows per thread and tile (256 threads cover 64 rows x 4 chunks)
    uint4 rk[RPT], rv[RPT];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int key = t * KT + i * 64 + skey;
            rk[i] = make_uint4(0u, 0u, 0u, 0u);
            rv[i] = rk[i];
            if (key < p.Lk) {
                rk[i] = *reinterpret_cast<const uint4*>(kbase + (size_t)key * p.ldk);
                rv[i] = *reinterpret_cast<const uint4*>(vbase + (size_t)key * p.ldv);
            }
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* base = lds + buf * TILE_BYTES;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            *reinterpret_cast<uint4*>(base + (i * 64 + skey) * LDS_ROW + schunk * 16) = rk[i];
            *reinterpret_cast<uint4*>(base + K_BYTES + (i * 64 + skey) * V_ROW + schunk * 16) = rv[i];
        }
    };

    float m_run = -INFINITY;  // running max of this lane's query (uniform over the 4 lanes sharing the query)
    float l_run = 0.f;        // running sum over THIS lane's keys only (combined across the 4 lanes at the end)
    float4v oacc[2] = {float4v{0.f, 0.f, 0.f, 0.f}, float4v{0.f, 0.f, 0.f, 0.f}};

    // ragged batch: keys outside the frame's valid (rows x cols) rectangle of the key map get -inf, like the additive
    // attention mask of the reference (HF:models/detr/modeling_detr.py:402-427, 933-991); key 0 is always valid
    const int kv_rows = MASKED ? p.key_valid[2 * b] : 0, kv_cols = MASKED ? p.key_valid[2 * b + 1] : 0;
    const int ntiles = (p.Lk + KT - 1) / KT;
    const float scale2 = p.scale * 1.44269504088896340736f;  // scores are kept pre-multiplied by log2(e)
    load_tile(0);
    store_tile(0);
    __syncthreads();

    // One key tile.  LAST (compile time): the tile may hold keys >= Lk; every other tile of an unmasked launch is full and carries no
    // masking code at all (a run-time "is this the last tile" inside one loop body made the compiler copy all 16 score registers per
    // tile to merge the two paths).  Packed fp32 arithmetic (v_pk_fma_f32 / v_pk_add_f32: two scores per instruction) for the
    // exponent arguments and the row sum; the exponentials themselves are one v_exp_f32 per score.
    auto tile_step = [&](const int t, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int buf = t & 1;
        if (!LAST) load_tile(t + 1);
        const unsigned char* Kl = lds + buf * TILE_BYTES;
        const unsigned char* Vl = Kl + K_BYTES;

        // ---- S^T = K Q^T ----------------------------------------------------------------------------------------
        float4v s[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const half8 kf = *reinterpret_cast<const half8*>(Kl + (kt * 16 + li) * LDS_ROW + g * 16);
            s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf, float4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
        // ---- online softmax (fp32, base-2 domain: p = exp2(s * scale*log2(e) - m), one fma + one v_exp per score; the
        //      running maximum is taken on the RAW scores and scaled once: scale > 0 keeps the order) ----------------------
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
        mx = xmax16_32(mx);   // over the 4 lanes (16 apart) that share the query
        const float m_new = fmaxf(m_run, mx * scale2);  // finite: tile 0 always holds key 0 (the product rounds once, like s*scale2)
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        const float2v sc2 = {scale2, scale2}, mneg = {-m_new, -m_new};
        float2v psum2 = {0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const float2v a = __builtin_elementwise_fma(float2v{s[kt][r], s[kt][r + 1]}, sc2, mneg);   // masked: fma(-inf) = -inf -> 0
                float2v e;
                e[0] = __builtin_amdgcn_exp2f(a[0]);
                e[1] = __builtin_amdgcn_exp2f(a[1]);
                s[kt][r] = e[0];
                s[kt][r + 1] = e[1];
                psum2 += e;
            }
        l_run = l_run * alpha + (psum2[0] + psum2[1]);
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[dt][r] *= alpha;

        // ---- O^T += V^T P^T -------------------------------------------------------------------------------------
#pragma unroll
        for (int kb = 0; kb < NKT / 2; ++kb) {
            half8 pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[r] = (_Float16)s[2 * kb][r];
                pf[4 + r] = (_Float16)s[2 * kb + 1][r];
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                half8 vf;
                if (TR) {
                    const unsigned char* a0 = Vl + (kb * 32 + g * 4 + (li >> 2)) * V_ROW + (dt * 16 + (li & 3) * 4) * 2;
                    const half4 lo = lds_tr16(a0);
                    const half4 hi = lds_tr16(a0 + 16 * V_ROW);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { vf[r] = lo[r]; vf[4 + r] = hi[r]; }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int key = kb * 32 + (j < 4 ? g * 4 + j : 16 + g * 4 + (j - 4));
                        vf[j] = *reinterpret_cast<const _Float16*>(Vl + key * V_ROW + (dt * 16 + li) * 2);
                    }
                }
                oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, oacc[dt], 0, 0, 0);
            }
        }
        if (!LAST) store_tile(buf ^ 1);
        __syncthreads();
    };
    for (int t = 0; t + 1 < ntiles; ++t) tile_step(t, std::false_type{});
    tile_step(ntiles - 1, std::true_type{});

    float l_tot = l_run + __shfl_xor(l_run, 16, 64);
    l_tot += __shfl_xor(l_tot, 32, 64);
    if (q_ok) {
        const float inv = 1.0f / l_tot;
        f16_t* orow = p.o + ((size_t)b * p.Lq + q) * p.ldo + h * 32 + g * 4;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            half4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (_Float16)(oacc[dt][r] * inv);
            *reinterpret_cast<half4*>(orow + dt * 16) = o;
        }
    }
}

}  // namespace

hipError_t opd_launch_attention(const AttnParams& p, hipStream_t stream) {
    if (p.B <= 0 || p.heads <= 0 || p.Lq <= 0 || p.Lk <= 0) return hipErrorInvalidValue;
    if ((p.ldq % 8) || (p.ldk % 8) || (p.ldv % 8) || (p.ldo % 4)) return hipErrorInvalidValue;  // 16-byte row chunks
    if (p.key_valid && p.key_row < 1) return hipErrorInvalidValue;
    const int total = ((p.Lq + 63) / 64) * p.heads * p.B;
    dim3 grid(8 * ((total + 7) / 8));   // 8 XCDs x their share of the tiles (attention_kernel's tile map)
    const bool wide = p.Lq <= 128 && p.Lk > 128;   // few query tiles, long key loop: decoder cross-attention
#define OPD_ATTN_LAUNCH(TRV, MV)                                                                             \
    do {                                                                                                     \
        if (wide) hipLaunchKernelGGL((attention_kernel<TRV, MV, 128>), grid, dim3(256), 0, stream, p);       \
        else hipLaunchKernelGGL((attention_kernel<TRV, MV, 64>), grid, dim3(256), 0, stream, p);             \
    } while (0)
    if (p.key_valid) {
        if (p.use_tr_read) OPD_ATTN_LAUNCH(true, true);
        else OPD_ATTN_LAUNCH(false, true);
    } else {
        if (p.use_tr_read) OPD_ATTN_LAUNCH(true, false);
        else OPD_ATTN_LAUNCH(false, false);
    }
#undef OPD_ATTN_LAUNCH
    return hipGetLastError();
}
