// kernels_btail3.hip — fused ResNet bottleneck tail for the 256-channel blocks of stage 3 (gfx950):
//     x1 --3x3(256->256)+ReLU--> a1 --1x1(256->1024) + residual + ReLU--> y --1x1(1024->C3)+ReLU--> z        (C3 = 256 or 0)
//
// SURVEY.md §8(a) row a4.  Reference arithmetic: HF:models/resnet/modeling_resnet.py:139-178 (ResNetBottleNeckLayer), FrozenBN folded at
// load; z is the NEXT block's 1x1 reduce.  Same fusion boundary as kernels_btail.hip (the input of a 3x3 is the only place that needs a
// halo), different decomposition: in stage 3 a wave that owned ALL channels of its pixels (the stage 1-2 kernel) would need 128 + 128
// accumulator registers for a1 / z and run one wave per SIMD (the round-2 "ETAIL" form, measured slower than the three launches).  Here
//   * one workgroup = 128 pixels, EIGHT waves, one workgroup per CU (2 waves per SIMD, 160 KiB of LDS);
//   * the four wave PAIRS own 32 pixels each; inside a pair the two waves split the OUTPUT channels of every GEMM
//     (3x3: 2 x 128 channels; expand chunk of 64: 2 x 32; reduce: 2 x 128), so every wave holds 2 x 8 accumulator tiles at most;
//   * what one wave of a pair produces and the other needs as a B operand (a1: once; the 64 channels of a y chunk: per chunk) crosses
//     through LDS in operand form (16 bytes per lane, lane-linear: 8 KiB resp. 2 KiB per wave) — fp16-rounded 16x16 accumulator tiles
//     ARE B operands under the k-permutation of opd_permute_k32, exactly as in kernels_btail.hip;
//   * weights stream by LDS-DMA: the 3x3 loop stages 48 KiB per k-step (128 pixel rows + 256 weight rows of 64 halfs), the 16 chunk
//     steps stage W2[64 rows][256] and W3[256 rows][64-column slice] (32 KiB each) into separate double buffers, so ONE barrier per
//     chunk orders everything: W2 of chunk j+2 and W3 of chunk j+1 are requested right after barrier j.
// HBM traffic per block at batch 8 (M = 33 600): 172 MB (x1 17, residual 69, y 69, z 17) against 275 MB for the three launches; a1 and
// the re-read of y disappear, and so do two launches' fill / drain phases.
#include <hip/hip_runtime.h>
#include "opd_kernels.h"
#include "opd_elem.h"

typedef elem_t half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

namespace {

constexpr int ROW_BYTES = 128;

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROW_BYTES + ((chunk ^ (row & 7)) << 4); }

__device__ __forceinline__ int xcd_logical_block(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7;
    const int x = bid & 7, k = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}
__device__ __forceinline__ int xcd_logical_block_rev(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7;
    const int x = bid & 7, k = bid >> 3;
    return (x < r ? x * (q + 1) + q - k : r * (q + 1) + (x - r) * q + q - 1 - k);
}
__device__ __forceinline__ int fdiv(const int m, const FastDiv& f) { return f.one ? m : (int)(__umulhi((unsigned)m, f.mul) >> f.shift); }

__device__ __forceinline__ unsigned pack2h(float a, float b) {
    typedef elem_t half2v __attribute__((ext_vector_type(2)));
    half2v h;
    h[0] = (elem_t)a;
    h[1] = (elem_t)b;
    unsigned u;
    __builtin_memcpy(&u, &h, 4);
    return u;
}
__device__ __forceinline__ void unpack2h(unsigned u, float& a, float& b) {
    typedef elem_t half2v __attribute__((ext_vector_type(2)));
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
template <int N_OUTSTANDING>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N_OUTSTANDING >= 0 && N_OUTSTANDING <= 63, "vmcnt range");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_OUTSTANDING) : "memory");
}
__device__ __forceinline__ void compiler_fence() { asm volatile("" ::: "memory"); }
// LDS writes / reads of this wave retired, then the workgroup barrier; nothing moves across it
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS map (160 KiB):  [0, 32K) W2 buffer 0 | [32K, 64K) W2 buffer 1 | [64K, 96K) W3 buffer 0 | [96K, 128K) W3 buffer 1 | [128K, 160K) y exchange x 2
// 3x3 phase: a RING of three 48-KiB stages (pixels 16 KiB + weights 32 KiB) at [0, 144K), two k-steps of DMA in flight: a lone workgroup
// on a CU with ONE stage in flight waits ~2000 clocks per k-step for its 48 KiB (measured: 0.96 us per k-step against 0.47 of MFMA work).
// The last k-step (35) sits in stage 2 = [96K, 144K): W2 of chunk 0 is requested during k-step 34 into [0, 32K) (stage 0 is free by then),
// W2 of chunk 1 and W3 of chunk 0 during k-step 35 into [32K, 96K); the a1 exchange (64 KiB) then uses [96K, 160K).
constexpr int W2BUF = 0, W3BUF = 65536, XBUF = 131072, EXBUF = 98304;
constexpr int STAGE_BYTES = 49152;
constexpr int LDS_BYTES = 163840;

// TRACE (tools/trace_btail.py): wave 0 stamps the shader clock at the phase boundaries and writes p.trace[blockIdx.x][16] at the end:
// {wall clock in, entry, prologue done, 3x3 loop done, a1 exchanged, chunks 1 / 3 / .. / 15 done (8 stamps), z stored, stores retired, wall clock out}
template <int C3, bool TRACE = false>
__global__ __launch_bounds__(512) void btail256_kernel(BtailParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long tstamp[16] = {};
    auto stamp = [&](const int i) {
        if constexpr (TRACE) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tstamp[i])::"memory");
    };
    if constexpr (TRACE) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tstamp[0])::"memory");
    stamp(1);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int C1 = 256, C2 = 1024, NCH = C2 / 64;
    static_assert(C3 == 0 || C3 == 256, "reduce width");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = wave >> 1, half = wave & 1;
    const int g = lane >> 4, li = lane & 15;
    const int m_base = (p.rev ? xcd_logical_block_rev(blockIdx.x, gridDim.x) : xcd_logical_block(blockIdx.x, gridDim.x)) * 128;

    // ---- staging coordinates: piece = 8 tile rows x 128 B, lane -> (row lane>>3, 16-byte slot lane&7), source-side XOR swizzle
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ lrow;
    const unsigned backoff = (unsigned)(p.W + 1) * (unsigned)C1 * 2u;   // pad = 1: every in-image tap gets a non-negative offset
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x1)) - backoff, 0, (unsigned)((size_t)p.B * p.H * p.W * C1 * 2) + backoff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.w1), 0, (unsigned)(C1 * 9 * C1 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.w2p), 0, (unsigned)(C2 * C1 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(C3 ? p.w3p : p.w2p), 0, (unsigned)((C3 ? C3 : 1) * C2 * 2), 0x00020000);
    unsigned rowoff[2], rowmask[2], woff1[4], woff2[4], woff3[4];
    {
        const int ohw = p.OH * p.OW;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m_base + (wave * 2 + i) * 8 + lrow;
            const bool okm = m < p.M;
            const int mm = okm ? m : 0;
            const int b = fdiv(mm, p.fd_ohw);
            const int r = mm - b * ohw;
            const int oh = fdiv(r, p.fd_ow);
            const int ow = r - oh * p.OW;
            rowoff[i] = (unsigned)(((b * p.H + oh * p.stride) * p.W + ow * p.stride) * C1) * 2u + (unsigned)lchunk * 16u;
            unsigned kwmask = 0, mask = 0;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
                if ((unsigned)(ow * p.stride - 1 + kw) < (unsigned)p.W) kwmask |= 1u << kw;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
                if ((unsigned)(oh * p.stride - 1 + kh) < (unsigned)p.H) mask |= kwmask << (kh * 3);
            rowmask[i] = okm ? mask : 0u;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = wave * 4 + i;
            woff1[i] = (unsigned)((q * 8 + lrow) * (9 * C1)) * 2u + (unsigned)lchunk * 16u;                       // W1 rows 8q.., this k-step's 64 halfs
            woff2[i] = (unsigned)(((q & 7) * 8 + lrow) * C1 + (q >> 3) * 64) * 2u + (unsigned)lchunk * 16u;       // W2 chunk: sub-tile q>>3, rows 8(q&7)..
            woff3[i] = (unsigned)((q * 8 + lrow) * C2) * 2u + (unsigned)lchunk * 16u;                            // W3 rows 8q.., this chunk's 64 halfs
        }
    }
    constexpr int kpc = C1 / 64;   // k-steps per filter tap
    constexpr int nk = 9 * kpc;
    int tap_kh = 0, tap_kw = 0, tap_c = 0;
    // Requests are issued ONE PIECE AT A TIME between groups of MFMAs (`compute_main(.., between)`), never as a burst: the CU's address path
    // takes ~15 clocks per 1-KiB request, and eight waves that all issue their 6 requests right after a barrier stand in that queue for
    // ~740 clocks with the matrix pipe idle (measured: DMA alone 0.35 us and MFMA + LDS alone 0.61 us per k-step, both together 0.82).
    int is_tap = 0, is_soff_a = 0, is_ks = 0;
    auto begin_issue = [&](int ks) {   // k-steps are requested in order: the tap counters advance by one per call
        is_tap = tap_kh * 3 + tap_kw;
        is_soff_a = ((tap_kh * p.W + tap_kw) * C1 + tap_c * 64) * 2;
        is_ks = ks;
        if (++tap_c == kpc) {
            tap_c = 0;
            if (++tap_kw == 3) { tap_kw = 0; ++tap_kh; }
        }
    };
    auto issue_piece = [&](int i, int stage_off) {   // i = 0, 1: pixel rows; 2 .. 5: weight rows
        unsigned char* As = smem + stage_off;
        if (i < 2) {
            const unsigned vo = ((rowmask[i] >> is_tap) & 1u) ? rowoff[i] : 0x80000000u;   // out of range -> zero fill
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void*)(As + (wave * 2 + i) * 1024), 16, vo, is_soff_a, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w1, (__attribute__((address_space(3))) void*)(As + 16384 + (wave * 4 + i - 2) * 1024), 16, woff1[i - 2], is_ks * 128, 0, 0);
        }
    };
    auto issue_main = [&](int ks, int stage_off) {
        begin_issue(ks);
#pragma unroll
        for (int i = 0; i < 6; ++i) issue_piece(i, stage_off);
    };
    auto issue_w2_piece = [&](int j, int i) {
        unsigned char* dst = smem + W2BUF + (j & 1) * 32768;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w2, (__attribute__((address_space(3))) void*)(dst + (wave * 4 + i) * 1024), 16, woff2[i], j * (64 * C1 * 2), 0, 0);
    };
    auto issue_w3_piece = [&](int j, int i) {
        if constexpr (C3 > 0) {
            unsigned char* dst = smem + W3BUF + (j & 1) * 32768;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w3, (__attribute__((address_space(3))) void*)(dst + (wave * 4 + i) * 1024), 16, woff3[i], j * 128, 0, 0);
        }
    };

    // ---- residual / y in the paired 16-byte layout: lane (g, li) -> pixel (g&1)*16 + li of the pair's 32, 8 channels at (g>>1)*8 of a 16-channel tile
    // Loads and stores go through bounds-checked buffer descriptors: rows >= M read zeros / are not written, an absent residual is a
    // descriptor of zero records -- no branch around any of them, so every wave issues exactly 2 loads and 2 stores per chunk and the
    // counted waits below hold on ragged tiles too.
    const int pr_m = m_base + pair * 32 + (g & 1) * 16 + li;
    const bool has_res = p.res != nullptr && !(p.dbg & 4);
    const unsigned y_bytes = (unsigned)((size_t)p.M * C2 * 2);
    const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(has_res ? p.res : p.x1), 0, has_res ? y_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (p.dbg & 2) ? 0u : y_bytes, 0x00020000);
    const unsigned pr_off = (unsigned)pr_m * (unsigned)(C2 * 2) + (unsigned)(half * 32 + (g >> 1) * 8) * 2u;   // bytes; rows >= M land beyond y_bytes
    auto load_res = [&](int j, uint4v (&r)[2]) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) r[nt] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, pr_off + (unsigned)(j * 64 + nt * 16) * 2u, 0, 0);
    };

    // ---- 3x3 main loop: wave = (pixels 32*pair.., channels 128*half..): 2 x 8 accumulator tiles ----------------------------------
    stamp(2);
    issue_main(0, 0);
    issue_main(1, STAGE_BYTES);
    float4v acc1[8][2];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
        const float4v b = *reinterpret_cast<const float4v*>(p.b1 + half * 128 + nt * 16 + g * 4);
        acc1[nt][0] = b;
        acc1[nt][1] = b;
    }
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) asm volatile("" : "+v"(acc1[nt][0]), "+v"(acc1[nt][1]));   // (bias loads retire here, before the counted waits)
    // The 3x3 loop runs as TWO WAVE GROUPS staggered by one barrier (waves 0-3 = pairs 0, 1; waves 4-7 = pairs 2, 3; SIMD s hosts waves s
    // and s + 4, one of each group).  A k-step is two phases separated by barriers,
    //     P1(k): read k-step k's fragments (stage k % 3) -> request k-step k+2's six pieces (stage (k+2) % 3) -> vmcnt(6) (retires the
    //            wave's pieces of stage k+1) -> lgkmcnt(0) -> barrier          M(k): 32 MFMAs at raised priority -> barrier
    // and group 1 executes one extra barrier up front, so while one group's waves run M(k) their SIMD partners of the other group run P1:
    // LDS reads, DMA issue (60-185 clocks per piece to the issuing wave) and barrier waits no longer stop the matrix pipe -- with all
    // eight waves in lock-step the same loop ran at 1.75 x the MFMA time.  Ordering (barriers numbered B_i; group 0 runs P1(k) in front of
    // B_2k and M(k) behind it, group 1 one barrier later):
    //   RAW  a wave reads stage k after B_(2k-1) at the earliest; every wave retired its pieces of stage k in P1(k-1), i.e. in front of
    //        B_(2k-2) (group 0) or B_(2k-1) (group 1);
    //   WAR  stage (k+2) % 3 = stage (k-1) % 3 is requested after B_(2k-1) at the earliest; its last readers are the P1(k-1) of both groups,
    //        whose lgkmcnt(0) sits in front of B_(2k-2) resp. B_(2k-1).
    const int group = wave >> 2;
    half8 xf[2][2], wf[2][8];
    auto read_frags = [&](int stage_off) {
        const unsigned char* As = smem + stage_off;
        const unsigned char* Ws = As + 16384;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) xf[kk][mt] = *reinterpret_cast<const half8*>(As + swz(pair * 32 + mt * 16 + li, kk * 4 + g));
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) wf[kk][nt] = *reinterpret_cast<const half8*>(Ws + swz(half * 128 + nt * 16 + li, kk * 4 + g));
        }
    };
    auto mfma_phase = [&](auto&& between) {   // `between(slot)`: after every eighth MFMA (slots 0 .. 3)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc1[nt][mt] = OPD_MFMA_16x16x32(wf[kk][nt], xf[kk][mt], acc1[nt][mt]);
                if ((nt & 3) == 3) {
                    __builtin_amdgcn_sched_barrier(0);
                    between(kk * 2 + (nt >> 2));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        __builtin_amdgcn_s_setprio(0);
        lds_barrier();
    };
    auto no_issue = [](int) {};
    static_assert(nk % 3 == 0, "stage bookkeeping below assumes the last k-step sits in stage 2");
    wait_vmcnt<6>();   // stage 0
    lds_barrier();
    if (group == 1) lds_barrier();   // the stagger
    int st_cur = 0, st_next2 = 2 * STAGE_BYTES;
#pragma unroll 1
    for (int ks = 0; ks + 2 < nk; ++ks) {
        read_frags(st_cur);
        compiler_fence();
        // requests of k-step ks+2: two pieces here, four between the MFMAs of M(ks) (an LDS-DMA request costs its wave 60-185 clocks of
        // issue; all six in P1 made P1 twice as long as M).  The wait retires stage ks+1 (requested in P1(ks-1) and M(ks-1)): the 2 pieces
        // just issued may fly.  WAR of the late pieces: they go out after B_2k (group 0) at the earliest, later than the early ones.
        begin_issue(ks + 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) issue_piece(i, st_next2);
        wait_vmcnt<2>();
        lds_barrier();
        mfma_phase([&](int slot) { issue_piece(2 + slot, st_next2); });   // (measured per k-step: 6 + 0: 1900 clocks, 3 + 3: 1680, 2 + 4: 1630, 0 + 6: 1740)
        st_cur = st_cur == 2 * STAGE_BYTES ? 0 : st_cur + STAGE_BYTES;
        st_next2 = st_next2 == 2 * STAGE_BYTES ? 0 : st_next2 + STAGE_BYTES;
    }
    // (biases are fetched like operands, ahead of their use and IN FRONT of the DMA requests of their step: a load issued right before its
    //  use would make the compiler wait for everything older, i.e. drain the DMA queue in the middle of a step)
    float4v acc2[2][2];   // expand accumulators; [tile][0] receives the bias of the NEXT chunk one step ahead (it is dead from the y epilogue on)
    auto load_bias2 = [&](int j) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc2[nt][0] = *reinterpret_cast<const float4v*>(p.b2 + j * 64 + half * 32 + nt * 16 + g * 4);
    };
    float4v accz[C3 ? 8 : 1][2];
    // k-step nk-2 (stage 1): stage 0 is free (WAR rule above) -> W2 of chunk 0 into [0, 32K)
    read_frags(STAGE_BYTES);
    compiler_fence();
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_w2_piece(0, i);
    wait_vmcnt<4>();   // retires stage nk-1
    lds_barrier();
    mfma_phase(no_issue);
    // k-step nk-1 (stage 2): [0, 96K) is free -> W2 of chunk 1, W3 of chunk 0, the first biases and residuals
    read_frags(2 * STAGE_BYTES);
    compiler_fence();
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_w2_piece(1, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_w3_piece(0, i);
    compiler_fence();
    load_bias2(0);
    if constexpr (C3 > 0) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) accz[nt][0] = *reinterpret_cast<const float4v*>(p.b3 + half * 128 + nt * 16 + g * 4);
    }
    uint4v res[3][2];   // residual of chunk j lives in res[j % 3], fetched two chunk steps ahead
    load_res(0, res[0]);
    load_res(1, res[1]);
    compiler_fence();
    lds_barrier();
    mfma_phase(no_issue);
    if (group == 0) lds_barrier();   // the groups are aligned again: every wave has passed the same number of barriers
    wait_vmcnt<0>();
    if constexpr (C3 > 0) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            asm volatile("" : "+v"(accz[nt][0]));   // the bias is in its registers here, not wherever the scheduler would sink the load to
            accz[nt][1] = accz[nt][0];
        }
    }
    lds_barrier();   // chunk 0's operands (every wave's pieces) are in place
    stamp(3);

    // ---- a1 = relu(c1) as fp16 B operands; the pair exchanges halves through [96K, 160K) -----
    half8 a1[2][8];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            float4v u = acc1[2 * kb][mt], v = acc1[2 * kb + 1][mt];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                u[r] = u[r] > 0.f ? u[r] : 0.f;
                v[r] = v[r] > 0.f ? v[r] : 0.f;
            }
            a1[mt][kb] = as_half8(pack2h(u[0], u[1]), pack2h(u[2], u[3]), pack2h(v[0], v[1]), pack2h(v[2], v[3]));
            *reinterpret_cast<half8*>(smem + EXBUF + wave * 8192 + (mt * 4 + kb) * 1024 + lane * 16) = a1[mt][kb];
        }
    lds_barrier();
    // a1 in the GLOBAL k-block order 0 .. 7 (selects on the wave-uniform `half`, once per workgroup): both waves of a pair accumulate in the
    // same k-block order as the unfused launches.  (The two paths still differ in the last bit on general data: the k-permutation inside a
    // 32-block changes the order in which one MFMA sums its products -- 0.04 % of the fp16 outputs differ by one ulp; they are bit-identical
    // on exactly representable sums, which is what the kernel tests assert.  A layer therefore runs through ONE of the paths for every row
    // and every batch size of a handle: the choice is made from the handle's configuration, never from the batch at hand.)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const half8 own = a1[mt][kb];
            const half8 oth = *reinterpret_cast<const half8*>(smem + EXBUF + (wave ^ 1) * 8192 + (mt * 4 + kb) * 1024 + lane * 16);
            a1[mt][kb] = half ? oth : own;
            a1[mt][4 + kb] = half ? own : oth;
        }
    lds_barrier();   // the exchange area is free again (W3 buffer 1 and the y exchange live there)
    stamp(4);
    if (p.dbg & 8) return;

    // ---- chunk steps: 64 channels of y each; this wave computes 32 of them (2 tiles) for the pair's 32 pixels; all eight waves in step, one
    // barrier per chunk.  (Two de-synchronised forms were built and measured against this loop's 7500 clocks per two chunks, and removed:
    // Y(c) | barrier | Z(c) | barrier with the wave groups one interval apart as in the 3x3 loop: 8000; Z one chunk behind Y, one barrier
    // per chunk, the two groups running {Y(c), Z(c-1)} in opposite order: 8400-9100.  What holds the chunk step at 1.8 x its MFMA time is
    // not the waves of a SIMD being in the same phase.)
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const unsigned char* W2s = smem + W2BUF + (j & 1) * 32768;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc2[nt][1] = acc2[nt][0];
        {   // y: 8 k-blocks x (2 weight fragments, 4 MFMAs); fragments run two k-blocks ahead of their MFMAs
            auto w2frag = [&](int t, int nt) {
                const int kb = t;   // (a1 sits in global k-block order)
                return *reinterpret_cast<const half8*>(W2s + (kb >> 1) * 8192 + swz(half * 32 + nt * 16 + li, (kb & 1) * 4 + g));
            };
            half8 ring[3][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) ring[t][nt] = w2frag(t, nt);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (t + 2 < 8) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) ring[(t + 2) % 3][nt] = w2frag(t + 2, nt);
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc2[nt][mt] = OPD_MFMA_16x16x32(ring[t % 3][nt], a1[mt][t], acc2[nt][mt]);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        // residual (paired layout -> accumulator layout), ReLU, fp16; store y; the fp16 values are this wave's k-block of the chunk
        uint4v (&res_cur)[2] = res[j % 3];
        unsigned pk[2][2][2];
        uint4v yo[2];   // this step's y values in store layout: written behind the wait below
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float4v v0 = acc2[nt][0], v1 = acc2[nt][1];
            {
                uint4v r = res_cur[nt];
                asm volatile("" : "+v"(r));   // (pins the first touch of the loaded registers HERE: the copies v_permlane16_swap needs were otherwise
                                              //  hoisted a whole step up, in front of a wait that drained the DMA queue)
                const uint2v s0 = __builtin_amdgcn_permlane16_swap(r[0], r[2], false, false);
                const uint2v s1 = __builtin_amdgcn_permlane16_swap(r[1], r[3], false, false);
                float a, b;
                unpack2h(s0[0], a, b); v0[0] += a; v0[1] += b;
                unpack2h(s1[0], a, b); v0[2] += a; v0[3] += b;
                unpack2h(s0[1], a, b); v1[0] += a; v1[1] += b;
                unpack2h(s1[1], a, b); v1[2] += a; v1[3] += b;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v0[r] = v0[r] > 0.f ? v0[r] : 0.f;
                v1[r] = v1[r] > 0.f ? v1[r] : 0.f;
            }
            pk[nt][0][0] = pack2h(v0[0], v0[1]);
            pk[nt][0][1] = pack2h(v0[2], v0[3]);
            pk[nt][1][0] = pack2h(v1[0], v1[1]);
            pk[nt][1][1] = pack2h(v1[2], v1[3]);
            const uint2v s0 = __builtin_amdgcn_permlane16_swap(pk[nt][0][0], pk[nt][1][0], false, false);
            const uint2v s1 = __builtin_amdgcn_permlane16_swap(pk[nt][0][1], pk[nt][1][1], false, false);
            yo[nt] = uint4v{s0[0], s1[0], s0[1], s1[1]};
        }
        half8 yf[2][2];   // [own / partner][m-tile]
        if constexpr (C3 > 0) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                yf[0][mt] = as_half8(pk[0][mt][0], pk[0][mt][1], pk[1][mt][0], pk[1][mt][1]);
                *reinterpret_cast<half8*>(smem + XBUF + (j & 1) * 16384 + wave * 2048 + mt * 1024 + lane * 16) = yf[0][mt];
            }
        }
        // barrier j: W3_j and W2_{j+1} (requested after barrier j-1; before the loop for j = 0) have landed.  vmcnt(0): a counted wait proves an
        // LDS-DMA request only against YOUNGER LDS-DMA requests -- stores and register loads retire out of order with respect to it
        // (tools/microbench/vmorder.hip; rounds 2-3 let the residual loads of chunk j+1 and this step's stores stay in flight here, which
        // held only because the operands had been requested a whole step earlier).  The residual loads are a phase old by now, and this
        // step's y stores go out BEHIND the barrier, so the wait does not stand on them.
        wait_vmcnt<0>();
        lds_barrier();
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) __builtin_amdgcn_raw_buffer_store_b128(yo[nt], rsrc_y, pr_off + (unsigned)(j * 64 + nt * 16) * 2u, 0, 0);
        // the residual of chunk j + 2 right away (its registers were read two steps ago): it is a whole step old when barrier j + 1 drains it
        if (j + 2 < NCH) load_res(j + 2, res[(j + 2) % 3]);
        if (j + 1 < NCH) load_bias2(j + 1);
        compiler_fence();
        if constexpr (C3 > 0) {
            // z: 2 k-blocks x 8 tiles x 2 MFMAs; weight fragments run four tiles ahead; the 8 DMA requests of this step (W3 of chunk j+1, W2 of
            // chunk j+2) go out one per tile over the first half
            const unsigned char* W3s = smem + W3BUF + (j & 1) * 32768;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {   // the chunk's two k-blocks in order: block `half` is this wave's, the other one the partner's
                const half8 oth = *reinterpret_cast<const half8*>(smem + XBUF + (j & 1) * 16384 + (wave ^ 1) * 2048 + mt * 1024 + lane * 16);
                const half8 own = yf[0][mt];
                yf[0][mt] = half ? oth : own;
                yf[1][mt] = half ? own : oth;
            }
            auto w3frag = [&](int idx) {
                const int kb = idx >> 3;   // k-block (32 channels of the chunk)
                return *reinterpret_cast<const half8*>(W3s + swz(half * 128 + (idx & 7) * 16 + li, kb * 4 + g));
            };
            half8 ring[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) ring[i] = w3frag(i);
#pragma unroll
            for (int idx = 0; idx < 16; ++idx) {
                const half8 wf = ring[idx & 3];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) accz[idx & 7][mt] = OPD_MFMA_16x16x32(wf, yf[idx >> 3][mt], accz[idx & 7][mt]);
                if (idx + 4 < 16) ring[idx & 3] = w3frag(idx + 4);
                __builtin_amdgcn_sched_barrier(0);
                if (p.dbg & 16) {}   // timing ablation: no weight requests in the chunk loop (tools only)
                else if (idx < 4) { if (j + 1 < NCH) issue_w3_piece(j + 1, idx); }
                else if (idx < 8) { if (j + 2 < NCH) issue_w2_piece(j + 2, idx - 4); }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            if (j + 2 < NCH) {
#pragma unroll
                for (int i = 0; i < 4; ++i) issue_w2_piece(j + 2, i);
            }
            compiler_fence();
        }
        if (j & 1) stamp(5 + (j >> 1));
    }

    // ---- z = relu(c0') -----------------------------------------------------------------------------------------------------------
    if constexpr (C3 > 0) {
        const __amdgpu_buffer_rsrc_t rsrc_z = __builtin_amdgcn_make_buffer_rsrc(p.z, 0, (p.dbg & 2) ? 0u : (unsigned)((size_t)p.M * C3 * 2), 0x00020000);
        const unsigned zoff = (unsigned)pr_m * (unsigned)(C3 * 2) + (unsigned)(half * 128 + (g >> 1) * 8) * 2u;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            float4v v0 = accz[nt][0], v1 = accz[nt][1];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v0[r] = v0[r] > 0.f ? v0[r] : 0.f;
                v1[r] = v1[r] > 0.f ? v1[r] : 0.f;
            }
            const uint2v s0 = __builtin_amdgcn_permlane16_swap(pack2h(v0[0], v0[1]), pack2h(v1[0], v1[1]), false, false);
            const uint2v s1 = __builtin_amdgcn_permlane16_swap(pack2h(v0[2], v0[3]), pack2h(v1[2], v1[3]), false, false);
            const uint4v o = {s0[0], s1[0], s0[1], s1[1]};
            __builtin_amdgcn_raw_buffer_store_b128(o, rsrc_z, zoff + (unsigned)(nt * 16) * 2u, 0, 0);
        }
    }
    if constexpr (TRACE) {
        stamp(13);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(14);
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tstamp[15])::"memory");
        if (threadIdx.x == 0 && p.trace) {
#pragma unroll
            for (int i = 0; i < 16; ++i) p.trace[(size_t)blockIdx.x * 16 + i] = tstamp[i];
        }
    }
#endif
}

template <int C3, bool TRACE = false>
hipError_t launch_btail256_t(const BtailParams& p, hipStream_t stream) {
    OPD_SET_MAX_LDS_ONCE((btail256_kernel<C3, TRACE>), LDS_BYTES);
    OPD_LAUNCH((btail256_kernel<C3, TRACE>), dim3((p.M + 127) / 128), dim3(512), LDS_BYTES, stream, p);
    static const char* const kname = opd_kernel_name("btail256_kernel<%d, %s>", C3, OPD_BOOLSTR(TRACE));
    opd_last_kernel_name = kname;
    return hipGetLastError();
}

}  // namespace

#ifndef OPD_ELEM_BF16
bool opd_btail256_supported(int C1, int C3) { return C1 == 256 && (C3 == 0 || C3 == 256); }
#endif

// called by opd_launch_btail (kernels_btail.hip) for C1 == 256, with the FastDiv fields filled and the offset ranges checked
hipError_t OPD_SYM(opd_launch_btail256)(const BtailParams& p, hipStream_t stream) {
    if (!opd_btail256_supported(p.C1, p.C3) || p.xs) return hipErrorInvalidValue;
    if (p.trace) return p.C3 ? launch_btail256_t<256, true>(p, stream) : hipErrorInvalidValue;
    return p.C3 ? launch_btail256_t<256>(p, stream) : launch_btail256_t<0>(p, stream);
}
