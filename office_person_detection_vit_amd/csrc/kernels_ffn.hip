// kernels_ffn.hip — two chained 1x1 GEMMs through a wide hidden layer in ONE kernel (gfx950, fp16 MFMA, fp32 accumulate):
//
//   FFN   : y = LayerNorm(x + fc2(relu(fc1(x) + b1)) + b2)       the transformer feed-forward block, post-LN
//           (HF:models/detr/modeling_detr.py:576-590 DetrMLP, :629-639); hidden width F = 2048
//   ETAIL : y = relu(a1 . W2^T + b2 + res) ; z = relu(y . W3^T + b3)      the tail of a stage-3 bottleneck: 1x1 expand
//           (256 -> 1024) + residual + ReLU, fused with the NEXT block's 1x1 reduce (1024 -> 256)
//           (HF:models/resnet/modeling_resnet.py:139-178); y is stored (it is the next residual)
//
// Both are   out[256] = g( W_b . f( W_a . in[256] ) )   with a hidden vector of F channels per row, and in both the hidden
// vector never has to come back from HBM: FFN used to write the [M][2048] fp16 hidden tensor (34 MB at batch 8), run fc2
// split-K into fp32 slabs and reduce them in a third kernel (22 + 21 + 10.5 us per encoder layer); the stage-3 expand and the
// next reduce were two launches that moved the 69 MB block output three times (46 + 28 us per block).
//
// Decomposition.  A workgroup owns 64 rows (tokens / pixels) and streams both weight matrices once, in chunks of 64 hidden
// channels: chunk c needs the 64 rows of W_a ([F][256]) and the 64-column slice of W_b ([256][F]).  The four waves are
// 2 row groups x 2 channel halves: wave (rg, fh) owns rows 32 rg .. 32 rg + 31 and, of every chunk, hidden channels
// 32 fh .. 32 fh + 31:
//     H^T[32 ch][32 rows] = W_a[ch, :] . in^T  (K = 256)  ->  f(.) in fp32, ONE rounding to fp16 (the rounding the unfused path
//     applied when it stored the tensor)  ->  acc[256][32 rows] += W_b[:, ch] . H   (K = 32: one MFMA k-step)
// With the weights as the A operand of the 16x16x32 MFMA (as everywhere in this library) an accumulator tile holds
// 4 consecutive channels x one row per lane, and two such tiles, rounded to fp16, ARE the B operand of the next MFMA under the
// fixed k-permutation `opd_permute_k32` — applied to W_b's K index once at load.  No LDS round trip, no shuffle.
// The input rows of a wave (32 x 256 fp16) stay in 64 VGPRs as B fragments for the whole kernel; the two partial sums of a
// row group are exchanged through LDS at the end.  FFN: each wave finishes one 16-row tile with all 256 channels (two-pass
// LayerNorm across the 4 lane groups of a row, no cross-wave statistics).  ETAIL: each wave finishes 128 channels of both row
// tiles, so that `v_permlane16_swap` pairs them into 16-byte stores like every other fp16 epilogue of the library; the hidden
// tile y takes the same paired layout for its residual loads and its stores (kernels_btail.hip).
//
// Staging: LDS-DMA (`buffer_load ... lds`), 1-KiB pieces of 8 rows x 128 B with the XOR swizzle on the SOURCE side
// (kernels_gemm.hip), two 64-KiB stage buffers = 128 KiB of LDS, one workgroup per CU; chunk c + 1 is in flight while
// chunk c computes.  Every staged byte is read from LDS by exactly two waves (the two row groups).  ETAIL keeps its y stores
// in flight across the chunk barrier: they are the youngest vector-memory operations of a chunk step, so the counted
// `s_waitcnt vmcnt(2)` retires the next chunk's operands and residual and leaves the stores flying (kernels_btail.hip).
//
// What bounds it: a workgroup pulls all the weights (2 MiB FFN, 1 MiB ETAIL) through its CU's LDS-DMA path (33-42 B/clk,
// DESIGN.md §2) for 64 rows; the MFMAs of those rows need 13.7 / 6.8 us at the full rate.
#include <hip/hip_runtime.h>
#include "opd_kernels.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

namespace {

constexpr int ROW_BYTES = 128;
constexpr int TM = 64;                         // rows per workgroup
constexpr int FC = 64;                         // hidden channels per chunk
constexpr int WA_BYTES = FC * 256 * 2;         // 32 KiB: 4 sub-tiles [64 ch][64 k]
constexpr int WB_BYTES = 256 * FC * 2;         // 32 KiB: [256 out][64 ch]
constexpr int STAGE_BYTES = WA_BYTES + WB_BYTES;
constexpr int SUB = FC * ROW_BYTES;            // one [64 rows][64 k] sub-tile: 8 KiB

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROW_BYTES + ((chunk ^ (row & 7)) << 4); }

__device__ __forceinline__ unsigned pack2h(float a, float b) {
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    half2v h;
    h[0] = (_Float16)a;
    h[1] = (_Float16)b;
    unsigned u;
    __builtin_memcpy(&u, &h, 4);
    return u;
}
__device__ __forceinline__ void unpack2h(unsigned u, float& a, float& b) {
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    half2v h;
    __builtin_memcpy(&h, &u, 4);
    a = (float)h[0];
    b = (float)h[1];
}
__device__ __forceinline__ half8 as_half8(unsigned a, unsigned b, unsigned c, unsigned d) {
    uint4v u = {a, b, c, d};
    half8 h;
    __builtin_memcpy(&h, &u, 16);
    return h;
}
__device__ __forceinline__ float4v relu4(float4v v) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
    return v;
}

template <bool ETAIL>
__global__ __launch_bounds__(256, 1) void ffn_kernel(FfnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave >> 1, fh = wave & 1;
    const int g = lane >> 4, li = lane & 15;
    const int m_base = blockIdx.x * TM + rg * 32;
    const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;
    const int nchunks = (p.dbg & 16) ? 2 : p.F / FC;   // dbg 16: two chunks only (prologue / epilogue cost)

    const __amdgpu_buffer_rsrc_t rsrc_wa = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.w1), 0, (unsigned)((size_t)p.F * 256 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_wb = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.w2p), 0, (unsigned)((size_t)256 * p.F * 2), 0x00020000);
    // wave w stages sub-tile w of the W_a chunk (k = 64 w .. 64 w + 63 of its 64 rows) and rows 64 w .. 64 w + 63 of the W_b chunk
    const unsigned waoff = (unsigned)(lrow * 256 + wave * 64) * 2u + (unsigned)lchunk * 16u;
    const unsigned wboff = (unsigned)((wave * 64 + lrow) * p.F) * 2u + (unsigned)lchunk * 16u;
    const unsigned wbstep = (unsigned)(8 * p.F) * 2u;
    // piece i of this wave's share of chunk c: i < 8 -> W_a, else W_b.  Inside the chunk loop the pieces are issued ONE AT A TIME
    // between MFMAs (a piece costs the issuing wave 100-200 cycles of address generation; sixteen in a row in front of the MFMAs
    // left the matrix pipe idle for half of every step: one wave per SIMD, nobody else to cover)
    auto issue_piece = [&](int c, int buf, int i) {
        unsigned char* Was = smem + buf * STAGE_BYTES;
        unsigned char* Wbs = Was + WA_BYTES;
        if (i < 8)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_wa, (__attribute__((address_space(3))) void*)(Was + wave * SUB + i * 1024), 16,
                                                     waoff + (unsigned)i * (8u * 256u * 2u), c * (FC * 256 * 2), 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_wb, (__attribute__((address_space(3))) void*)(Wbs + (wave * 8 + i - 8) * 1024), 16,
                                                     wboff + (unsigned)(i - 8) * wbstep, c * (FC * 2), 0, 0);
    };
    auto issue = [&](int c, int buf) {
#pragma unroll
        for (int i = 0; i < 16; ++i) issue_piece(c, buf, i);
    };
    issue(0, 0);

    // b1 lives in LDS behind the stage buffers: a global load inside the chunk loop would be the youngest vector-memory operation
    // when its value is needed, and waiting for it drains the LDS-DMA of the next chunk that was issued just before it
    float* const b1s = reinterpret_cast<float*>(smem + 2 * STAGE_BYTES);
    for (int i = tid; i < p.F; i += 256) b1s[i] = p.b1[i];
    // the epilogue's per-channel vectors too: global loads interleaved with its stores made it a chain of 16 exposed L2 round trips
    float* const b2s = b1s + p.F;
    b2s[tid] = p.b2[tid];
    if constexpr (!ETAIL) {
        b2s[256 + tid] = p.gamma[tid];
        b2s[512 + tid] = p.beta[tid];
    }

    // ETAIL: hidden tile in the paired 16-byte layout: lane (g, li) <-> row (g & 1) * 16 + li, 8 channels at (g >> 1) * 8 of a
    // 16-channel tile (two row tiles of one accumulator register pair, `v_permlane16_swap`)
    const int pr_m = m_base + (g & 1) * 16 + li;
    const bool pr_ok = pr_m < p.M;
    const size_t pr_hid = (size_t)pr_m * p.F + fh * 32 + (g >> 1) * 8;     // + c * 64 + nt * 16
    // The residual loads and the y stores of the chunk loop are inline asm with a fixed count per step (rows >= M load the last
    // valid row and store to the caller's dump slot): the compiler must not see them, or it adds its own conservative waits
    // (vmcnt(0) in front of the first MFMA of a step: measured in the .s) — the explicit counted waits below, tied to the registers,
    // order them instead.
    const size_t ld_hid = (size_t)(pr_ok ? pr_m : p.M - 1) * p.F + fh * 32 + (g >> 1) * 8;
    auto load_res = [&](int c, uint4v (&r)[2]) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const f16_t* ptr = p.res16 + ld_hid + c * FC + nt * 16;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[nt]) : "v"(ptr) : "memory");
        }
    };
    uint4v res_a[2], res_b[2];   // residual of chunk c lives in res_a (c even) / res_b (c odd), fetched one chunk step ahead
    if constexpr (ETAIL) load_res(0, res_a);

    // this wave's 32 rows of the input as B fragments: xf[mt][ks] = in[m_base + 16 mt + li][32 ks + 8 g .. + 7]; rows >= M: zeros
    half8 xf[2][8];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = m_base + mt * 16 + li;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            half8 v = {};
            if (m < p.M) v = *reinterpret_cast<const half8*>(p.x16 + (size_t)m * 256 + ks * 32 + g * 8);
            xf[mt][ks] = v;
        }
    }
    float4v acc2[16][2];
#pragma unroll
    for (int nt = 0; nt < 16; ++nt) {
        acc2[nt][0] = float4v{0.f, 0.f, 0.f, 0.f};
        acc2[nt][1] = float4v{0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(res_a[0]), "+v"(res_a[1])::"memory");
    __syncthreads();   // chunk 0 and the biases are in LDS

    // one chunk step; `res_cur` / `res_next` are named registers (a run-time index would move them to scratch)
    auto step = [&](int c, uint4v (&res_cur)[2], uint4v (&res_next)[2]) {
        if constexpr (ETAIL) {
            if (c + 1 < nchunks) load_res(c + 1, res_next);   // (the first vector-memory operations of the step: see the wait at its end)
        }
        const bool dma = !(p.dbg & 2) || c == 0;
        if ((p.dbg & 4) && dma) issue(c + 1, (c + 1) & 1);
        // default: two pieces per k-step of GEMM a, so that the last of them has GEMM b's time to land (measured: 64 us against 72 us
        // for one piece per k-step / tile pair over both GEMMs, 70 us for all sixteen in front: tools/bench_ffn.py)
        const bool spread = (p.dbg & 8) && dma, early = !(p.dbg & 12) && dma;
        const unsigned char* Was = smem + (c & 1) * STAGE_BYTES;
        const unsigned char* Wbs = Was + WA_BYTES;
        // ---- GEMM a: this wave's 32 hidden channels of the chunk x its 32 rows, K = 256 -----------------------------------
        // (fragments are read one k-step ahead of the MFMAs that use them: with one wave per SIMD nobody else covers the LDS latency.
        //  The staging pieces of chunk c + 1 are issued unconditionally: past the last chunk the buffer descriptor's range check
        //  turns them into zero fills of a stage buffer nobody reads any more.)
        float4v h[2][2];
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            const float4v b = *reinterpret_cast<const float4v*>(b1s + c * FC + fh * 32 + ft * 16 + g * 4);
            h[ft][0] = b;
            h[ft][1] = b;
        }
        auto read_wa = [&](int ks, int ft) {
            return *reinterpret_cast<const half8*>(Was + (ks >> 1) * SUB + swz(fh * 32 + ft * 16 + li, (ks & 1) * 4 + g));
        };
        auto read_wb = [&](int nt) { return *reinterpret_cast<const half8*>(Wbs + swz(nt * 16 + li, fh * 4 + g)); };
        half8 wa[3][2];   // GEMM a fragments, read two k-steps ahead
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            wa[0][ft] = read_wa(0, ft);
            wa[1][ft] = read_wa(1, ft);
        }
        half8 wb[4];      // GEMM b fragments, read three tiles ahead (the first three during GEMM a's tail)
        const bool compute = !(p.dbg & 1);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (compute) {
#pragma unroll
                for (int ft = 0; ft < 2; ++ft)
                    if (ks + 2 < 8) wa[(ks + 2) % 3][ft] = read_wa(ks + 2, ft);
                if (ks >= 5) wb[ks - 5] = read_wb(ks - 5);
#pragma unroll
                for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) h[ft][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ks % 3][ft], xf[mt][ks], h[ft][mt], 0, 0, 0);
            }
            if (spread) issue_piece(c + 1, (c + 1) & 1, ks);
            if (early) { issue_piece(c + 1, (c + 1) & 1, 2 * ks); issue_piece(c + 1, (c + 1) & 1, 2 * ks + 1); }
            __builtin_amdgcn_sched_barrier(0);   // keep the reads two k-steps ahead: the scheduler would sink them next to their use
        }
        // ---- f(.): (+ residual,) ReLU, one fp16 rounding; two channel tiles = one B fragment (k order of opd_permute_k32) ----
        unsigned pk[2][2][2];   // [channel tile][row tile][register pair]
        uint4v yv[2];
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            float4v v0 = h[ft][0], v1 = h[ft][1];
            if constexpr (ETAIL) {
                {   // paired layout -> accumulator layout
                    const uint4v r = res_cur[ft];
                    const uint2v s0 = __builtin_amdgcn_permlane16_swap(r[0], r[2], false, false);
                    const uint2v s1 = __builtin_amdgcn_permlane16_swap(r[1], r[3], false, false);
                    float a, b;
                    unpack2h(s0[0], a, b); v0[0] += a; v0[1] += b;
                    unpack2h(s1[0], a, b); v0[2] += a; v0[3] += b;
                    unpack2h(s0[1], a, b); v1[0] += a; v1[1] += b;
                    unpack2h(s1[1], a, b); v1[2] += a; v1[3] += b;
                }
            }
            v0 = relu4(v0);
            v1 = relu4(v1);
            pk[ft][0][0] = pack2h(v0[0], v0[1]);
            pk[ft][0][1] = pack2h(v0[2], v0[3]);
            pk[ft][1][0] = pack2h(v1[0], v1[1]);
            pk[ft][1][1] = pack2h(v1[2], v1[3]);
            if constexpr (ETAIL) {   // store y: 16 bytes per lane
                const uint2v s0 = __builtin_amdgcn_permlane16_swap(pk[ft][0][0], pk[ft][1][0], false, false);
                const uint2v s1 = __builtin_amdgcn_permlane16_swap(pk[ft][0][1], pk[ft][1][1], false, false);
                yv[ft] = uint4v{s0[0], s1[0], s0[1], s1[1]};   // stored after GEMM b: the stores must be the step's youngest operations
            }
        }
        half8 hb[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) hb[mt] = as_half8(pk[0][mt][0], pk[0][mt][1], pk[1][mt][0], pk[1][mt][1]);
        // ---- GEMM b: all 256 output channels += W_b[:, these 32 hidden channels] . H ----------------------------------------
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
            if (compute) {
                if (nt + 3 < 16) wb[(nt + 3) & 3] = read_wb(nt + 3);
                acc2[nt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[nt & 3], hb[0], acc2[nt][0], 0, 0, 0);
                acc2[nt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[nt & 3], hb[1], acc2[nt][1], 0, 0, 0);
            }
            if ((nt & 1) && spread) issue_piece(c + 1, (c + 1) & 1, 8 + (nt >> 1));
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (ETAIL) {
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) {
                f16_t* ptr = pr_ok ? p.hid16 + pr_hid + c * FC + ft * 16 : p.dump;
                asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(ptr), "v"(yv[ft]) : "memory");   // (s_nop: the store reads its data
                // registers a little after issue; the compiler's next instruction may overwrite them — it pads nothing around asm)
            }
        }
        // chunk c + 1 has landed and every wave is done with chunk c's buffer.  ETAIL: this step's two y stores are its youngest
        // vector-memory operations (issued after the next chunk's DMA and residual loads), so they may stay in flight.
        if constexpr (ETAIL) asm volatile("s_waitcnt vmcnt(2)" : "+v"(res_next[0]), "+v"(res_next[1])::"memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    if constexpr (ETAIL) {
        for (int c = 0; c < nchunks; c += 2) {   // F % 128 == 0 (checked by the launcher)
            step(c, res_a, res_b);
            step(c + 1, res_b, res_a);
        }
    } else {
        for (int c = 0; c < nchunks; ++c) step(c, res_a, res_a);
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the last y stores)
    if (p.dbg & 32) return;                                // dbg 32: no epilogue
    float4v* ex = reinterpret_cast<float4v*>(smem);   // [wave][16][lane] float4: 16 KiB per wave (the stage buffers are free now)
    if constexpr (!ETAIL) {
        // ---- FFN: wave fh finishes row tile mt = fh with all 256 channels -----------------------------------------------------
        // (written once per value of fh with compile-time accumulator indices: a run-time index would move acc2 to scratch)
        float4v v[16];
        const int m = m_base + fh * 16 + li;
        const bool live = m < p.M;
        float4v rs[16];   // residual rows: all sixteen loads in flight across the exchange
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
            rs[nt] = float4v{0.f, 0.f, 0.f, 0.f};
            if (live) rs[nt] = *reinterpret_cast<const float4v*>(p.res32 + (size_t)m * 256 + nt * 16 + g * 4);
        }
        if (fh == 0) {
#pragma unroll
            for (int nt = 0; nt < 16; ++nt) ex[(wave * 16 + nt) * 64 + lane] = acc2[nt][1];
        } else {
#pragma unroll
            for (int nt = 0; nt < 16; ++nt) ex[(wave * 16 + nt) * 64 + lane] = acc2[nt][0];
        }
        __syncthreads();
        if (fh == 0) {
#pragma unroll
            for (int nt = 0; nt < 16; ++nt) v[nt] = acc2[nt][0] + ex[((wave ^ 1) * 16 + nt) * 64 + lane];   // half 0 + half 1
        } else {
#pragma unroll
            for (int nt = 0; nt < 16; ++nt) v[nt] = ex[((wave ^ 1) * 16 + nt) * 64 + lane] + acc2[nt][1];   // half 0 + half 1
        }
        float sum = 0.f;
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
            const int ch = nt * 16 + g * 4;
            const float4v t = v[nt] + *reinterpret_cast<const float4v*>(b2s + ch) + rs[nt];
            v[nt] = t;
            sum += t[0] + t[1] + t[2] + t[3];
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / 256.0f);
        float sq = 0.f;
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
            v[nt] -= mean;
            sq += v[nt][0] * v[nt][0] + v[nt][1] * v[nt][1] + v[nt][2] * v[nt][2] + v[nt][3] * v[nt][3];
        }
        sq += __shfl_xor(sq, 16);
        sq += __shfl_xor(sq, 32);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / 256.0f) + 1e-5f);
        if (live) {
#pragma unroll
            for (int nt = 0; nt < 16; ++nt) {
                const int ch = nt * 16 + g * 4;
                const float4v gm = *reinterpret_cast<const float4v*>(b2s + 256 + ch);
                const float4v bt = *reinterpret_cast<const float4v*>(b2s + 512 + ch);
                float4v o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = v[nt][q] * rstd * gm[q] + bt[q];
                if (p.y32) *reinterpret_cast<float4v*>(p.y32 + (size_t)m * 256 + ch) = o;
                if (p.y16) {
                    half4 hh;
                    hh[0] = (_Float16)o[0]; hh[1] = (_Float16)o[1]; hh[2] = (_Float16)o[2]; hh[3] = (_Float16)o[3];
                    *reinterpret_cast<half4*>(p.y16 + (size_t)m * 256 + ch) = hh;
                }
            }
        }
    } else {
        // ---- ETAIL: wave fh finishes output channels 128 fh .. 128 fh + 127 of both row tiles: z = relu(half 0 + half 1 + b) ----
        if (fh == 0) {
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                ex[(wave * 16 + 2 * nt) * 64 + lane] = acc2[8 + nt][0];
                ex[(wave * 16 + 2 * nt + 1) * 64 + lane] = acc2[8 + nt][1];
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                ex[(wave * 16 + 2 * nt) * 64 + lane] = acc2[nt][0];
                ex[(wave * 16 + 2 * nt + 1) * 64 + lane] = acc2[nt][1];
            }
        }
        __syncthreads();
        float4v v[8][2];
        if (fh == 0) {
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                v[nt][0] = acc2[nt][0] + ex[((wave ^ 1) * 16 + 2 * nt) * 64 + lane];
                v[nt][1] = acc2[nt][1] + ex[((wave ^ 1) * 16 + 2 * nt + 1) * 64 + lane];
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                v[nt][0] = ex[((wave ^ 1) * 16 + 2 * nt) * 64 + lane] + acc2[8 + nt][0];
                v[nt][1] = ex[((wave ^ 1) * 16 + 2 * nt + 1) * 64 + lane] + acc2[8 + nt][1];
            }
        }
        const size_t zrow = (size_t)pr_m * 256 + fh * 128 + (g >> 1) * 8;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            const float4v b = *reinterpret_cast<const float4v*>(b2s + fh * 128 + nt * 16 + g * 4);
            const float4v v0 = relu4(v[nt][0] + b), v1 = relu4(v[nt][1] + b);
            const uint2v s0 = __builtin_amdgcn_permlane16_swap(pack2h(v0[0], v0[1]), pack2h(v1[0], v1[1]), false, false);
            const uint2v s1 = __builtin_amdgcn_permlane16_swap(pack2h(v0[2], v0[3]), pack2h(v1[2], v1[3]), false, false);
            if (pr_ok) *reinterpret_cast<uint4*>(p.y16 + zrow + nt * 16) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
    }
#endif
}

template <bool ETAIL>
hipError_t launch_ffn_t(const FfnParams& p, hipStream_t stream) {
    const int LDS = 2 * STAGE_BYTES + p.F * 4 + 768 * 4;   // stage buffers + b1 + (b2, gamma, beta)
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_kernel<ETAIL>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES + 4096 * 4 + 768 * 4);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((ffn_kernel<ETAIL>), dim3((p.M + TM - 1) / TM), dim3(256), LDS, stream, p);
    return hipGetLastError();
}

}  // namespace

hipError_t opd_launch_ffn(const FfnParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.F < FC || p.F % FC != 0 || p.F > 4096 || !p.x16 || !p.w1 || !p.b1 || !p.w2p || !p.b2) return hipErrorInvalidValue;
    if ((size_t)p.F * 256 * 2 >= 0x7fffff00ull) return hipErrorInvalidValue;  // 31-bit buffer offsets
    if (p.etail) {
        if (!p.hid16 || !p.y16 || !p.res16 || !p.dump || p.F % (2 * FC) != 0) return hipErrorInvalidValue;
        return launch_ffn_t<true>(p, stream);
    }
    if (!p.res32 || !p.gamma || !p.beta) return hipErrorInvalidValue;
    return launch_ffn_t<false>(p, stream);
}
