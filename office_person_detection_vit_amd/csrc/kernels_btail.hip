// kernels_btail.hip — fused ResNet bottleneck tail for gfx950:   c1 (3x3) -> c2 (1x1 expand) + residual + ReLU -> c0' (1x1)
//
// SURVEY.md §8(a) row a4.  Reference arithmetic: HF:models/resnet/modeling_resnet.py:139-178 (ResNetBottleNeckLayer:
// shortcut + [1x1 reduce, 3x3, 1x1 expand] then activation), FrozenBN folded at load.
//
// In stages 1-2 of the trunk (C1 = 64 / 128 channels inside the block, C2 = 4*C1 outside) the unfused layers are HBM
// bound: the 3x3 output and the expand output each make a round trip through HBM only to be re-read by a 1x1 whose whole
// reduction dimension fits in one wave's registers.  The natural fusion boundary of a ResNet-v1.5 trunk is the INPUT of a
// 3x3 convolution (the only operator that needs a halo), so one kernel runs
//     x1 --3x3(C1->C1)+ReLU-->  a1  --1x1(C1->C2) + residual + ReLU-->  y  --1x1(C2->C3)+ReLU--> z
// where z is the NEXT block's reduce output (c0' of block i+1, or the first block of the next stage).  a1 never leaves
// registers, y is written once (it is the next residual) and never re-read for c0'.  HBM traffic per block drops from
// 1091 MB to 682 MB (stage 1, batch 8), and three launches' latency chains become one.
//
// Work decomposition: a workgroup owns 128 output pixels, wave w the 32 pixels 32w..32w+31 with ALL channels, so every
// reduction of the two 1x1s is wave-local.  MFMA orientation as in kernels_gemm.hip: weights are the A operand, pixels
// the B operand, D[row = channel][col = pixel]: lane (g = lane>>4, i = lane&15) holds channels 4g..4g+3 of pixel i.
// Channel ownership: the weight rows of every 32-channel block are STAGED into LDS in the order  LDS row 16t+4g+r <- channel 8g+4t+r
// (`own_row`; the staging offsets are per lane, so the permutation costs nothing), i.e. accumulator tile t of the block holds, in lane
// (g, i), channels 8g+4t .. +3 of pixel i.  The two tiles of a block, rounded to fp16, are then 8 CONSECUTIVE channels per lane:
// (1) they ARE the B operand of the next 16x16x32 MFMA in natural k order (no K-permuted weight copies), (2) y, z and the residual move as
// 16 bytes per lane with the four lanes of a pixel covering 64 contiguous bytes (16 pixels x 64 B per wave instruction; the accumulator
// layout of plain row order gives 8 bytes per lane, and a lane-pair exchange 32 rows x 32 B).
//
// Pipeline: the 3x3 main loop is the LDS-DMA loop of conv_gemm_dma_kernel (two stage buffers, buffer-descriptor
// staging, XOR-swizzled 128-byte rows).  It continues seamlessly into C2/64 "chunk steps": chunk j stages the 64 rows of
// W2 and the 64-column slice of W3 that chunk needs into the stage buffer the previous step has left, computes 64
// channels of y for the wave's 32 pixels, applies bias/residual/ReLU, stores them, and feeds them straight into the z
// accumulators.  The residual for chunk j+2 is fetched during chunk j; the wait that ends a chunk step counts only younger LDS-DMA
// requests (see `res_dma` below), and the step's y stores are issued behind it.
//
// Measured (tools/bench_btail.py, batch 8): stage-1 tail 188 us against 287 us for the three launches it replaces,
// stage-2 tail 135 against 170.  What does NOT move it further (each built, measured, removed):
// the residual staged through LDS by a dedicated fifth wave (SIMD imbalance: 238 us) or by one of the four waves with the
// operand staging split over the other three (190 us), whole-line y/z stores transposed through LDS (190 us), an
// input-stationary 3x3 loop on 8 x 16 pixel tiles with the halo patch of x1 in LDS (43 % fewer staged bytes: 188 us), a
// start-up stagger between workgroups sharing a CU (no phase locking), three instead of two workgroups per CU (185 vs
// 189 us).  PMC: no HBM credit stalls, the vector-memory address FIFO is full 45 % of the busy cycles.
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

// LDS row -> weight row (output channel) of the tails' channel ownership: bits [t][g1 g0][r1 r0] <- [g1 g0][t][r1 r0] within a 32-block
__device__ __forceinline__ int own_row(const int rho) { return (rho & ~31) | (((rho >> 2) & 3) << 3) | (((rho >> 4) & 1) << 2) | (rho & 3); }
// first of the four consecutive channels lane group g holds in accumulator tile nt
__device__ __forceinline__ int own_ch(const int nt, const int g) { return (nt >> 1) * 32 + g * 8 + (nt & 1) * 4; }

__device__ __forceinline__ int fdiv(const int m, const FastDiv& f) {   // m >= 0
    return f.one ? m : (int)(__umulhi((unsigned)m, f.mul) >> f.shift);
}

__device__ __forceinline__ int xcd_logical_block_rev(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7;
    const int x = bid & 7, k = bid >> 3;
    return (x < r ? x * (q + 1) + q - k : r * (q + 1) + (x - r) * q + q - 1 - k);
}

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

// RDMA = true (C1 == 64): the residual travels global -> LDS through LDS-DMA instead of global -> VGPR.  Loads through
// VGPRs move ~16 B/clk/CU, LDS-DMA 33-42 (tools/microbench/ldpath.hip); each wave stages only ITS OWN 32 rows (four
// 8-row x 128-byte pieces per 64-channel chunk) into a wave-private quarter of two 16-KiB buffers, so no barrier and no
// special wave are involved: chunk j+2 is requested right after the wave has read chunk j out of the same buffer.
// SC = true (first block of stage 1: C1 == 64, shortcut input of 64 channels at the output resolution): the block's 1x1 SHORTCUT
// convolution is a second GEMM into the same accumulators, y = relu(a1 . W2^T + xs . Wsc^T + (b2 + bsc)), instead of a residual
// that a separate launch wrote (274 MB at batch 8) and this kernel read back: the 64 shortcut channels of the wave's 32 pixels sit
// in 16 VGPRs as B fragments, the 64 x 64 slice of Wsc travels with the W2 / W3 chunk operands.  The shortcut is no longer
// rounded to fp16 on the way (one rounding fewer than the unfused path; the reference rounds nothing).
// TRACE (tools/trace_btail.py): wave 0 stamps the shader clock at the phase boundaries and writes p.trace[blockIdx.x][16] at the end:
// {wall clock in, entry, prologue done, 3x3 loop done, chunk 0 .. NCH-1 done, stores retired, ..., [15] wall clock out}.
// RC = 1 (round 5; second block of stage 1): the residual is not read but REBUILT.  The previous block's output y' = relu(W2' . a1' + Wsc . xs +
// b') is 256 channels wide, its ingredients a1' and xs 64 each, and a 1x1 needs no halo: this kernel reads a1' and xs of its own 32 pixels per
// wave as B fragments (32 KiB per workgroup instead of the 64-KiB residual), streams the 64 x 64 slices of W2' and Wsc with its own chunk
// operands and runs the previous tail's chunk arithmetic -- same operands, same MFMA order, same fp16 rounding -- so the residual it adds has
// the bits that tail would have stored.  The previous tail then stores a1' (68 MB at batch 8) instead of y' (274 MB), and nobody reads y'.
// NW: waves per workgroup = 32-pixel row groups per tile.  4: 128-pixel tiles, two workgroups per CU; 8: 256-pixel tiles, one workgroup per CU -- the
// same eight waves on a CU, but one set of weight tiles staged for 256 pixels instead of two sets for 128 each (round 5: the SQ counters of a
// stage-2 tail read SQ_VMEM_TA_CMD_FIFO_FULL for 80 % of the busy cycles -- the CU's vector-memory path is what the tails wait for, and
// 40 % of what they push through it are weight tiles).  A wave's arithmetic does not depend on NW: identical bits.
template <int C1, int C3, bool RDMA, bool SC = false, bool TRACE = false, int RC = 0, int NW = 4>
__global__ __launch_bounds__(64 * NW, 8 / NW) void btail_kernel(BtailParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long tstamp[16] = {};
    auto stamp = [&](const int i) {
        if constexpr (TRACE) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tstamp[i])::"memory");
    };
    if constexpr (TRACE) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tstamp[0])::"memory");
    stamp(1);
    static_assert(!RDMA || C1 == 64, "residual staging buffers are budgeted for C1 == 64 (80 KiB of LDS per workgroup)");
    static_assert(!RDMA || RC == 0, "a rebuilt residual is not staged");
    static_assert(!SC || (C1 == 64 && !RDMA), "fused shortcut: 64-channel tails only; it replaces the residual");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int C2 = 4 * C1;
    constexpr int NT1 = C1 / 16;               // c1 accumulator tiles per wave (all C1 channels)
    constexpr int KK1 = C1 / 32;               // 32-channel k-blocks of a1
    static_assert(NW == 4 || NW == 8, "waves per workgroup");
    constexpr int BM = 32 * NW;                // pixels per tile
    constexpr int A_BYTES = BM * ROW_BYTES;    // BM pixels x 64 halfs
    constexpr int RC_BYTES = RC ? 2 * 64 * ROW_BYTES : 0;    // chunks of the previous block's W2 and Wsc: [64 rows][64 halfs] each
    constexpr int W1_PIECES = C1 / (8 * NW);   // 1-KiB pieces of the W1 tile per wave
    constexpr int NCH = C2 / 64;               // 64-channel chunks of y
    constexpr int W2C_BYTES = 64 * C1 * 2;     // chunk of W2: C1/64 sub-tiles of [64 rows][64 halfs]
    constexpr int W2_PIECES = W2C_BYTES / (1024 * NW);
    constexpr int W3_PIECES = C3 / (8 * NW);   // slice of W3: [C3 rows][64 halfs]
    constexpr int SC_PIECES = 8 / NW;          // a [64 rows][64 halfs] slice (Wsc; RC: the previous block's W2 and Wsc): pieces per wave
    static_assert(W1_PIECES >= 1 && W2_PIECES >= 1 && (C3 == 0 || W3_PIECES >= 1), "every wave stages at least one piece of every operand");
    constexpr int NT3 = C3 / 16;
    constexpr int WSC_BYTES = SC ? 64 * ROW_BYTES : 0;   // chunk of Wsc: [64 rows][64 halfs]
    constexpr int CHUNK_BYTES = W2C_BYTES + C3 * ROW_BYTES + WSC_BYTES + RC_BYTES;
    constexpr int STAGE_BYTES = (A_BYTES + C1 * ROW_BYTES) > CHUNK_BYTES ? (A_BYTES + C1 * ROW_BYTES) : CHUNK_BYTES;   // a stage buffer holds a 3x3 k-step's tiles or a chunk's operands
    static_assert(RC == 0 || (RC == 1 && C1 == 64 && !RDMA && !SC), "residual rebuild: second block of stage 1");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    // p.rev: every XCD walks its range of tiles backwards, so that a kernel starts on the rows its predecessor wrote LAST (the ones still
    // in the 256-MiB Infinity Cache; a 274-MB tensor written and re-read in the same order never hits)
    const int m_base = (p.rev ? xcd_logical_block_rev(blockIdx.x, gridDim.x) : xcd_logical_block(blockIdx.x, gridDim.x)) * BM;
    const int wm0 = m_base + wave * 32;        // this wave's 32 pixels

    // ---- staging coordinates (see conv_gemm_dma_kernel): piece = 8 tile rows x 128 B, lane -> (row lane>>3, slot lane&7)
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ lrow;
    const unsigned backoff = (unsigned)(p.W + 1) * (unsigned)C1 * 2u;  // pad = 1: every in-image tap gets a non-negative offset
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x1)) - backoff, 0, (unsigned)((size_t)p.B * p.H * p.W * C1 * 2) + backoff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.w1), 0, (unsigned)(C1 * 9 * C1 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.w2p), 0, (unsigned)(C2 * C1 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(C3 ? p.w3p : p.w2p), 0, (unsigned)((C3 ? C3 : 1) * C2 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_sc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(SC ? p.wsc : RC ? p.rc_wsc : p.w2p), 0, (unsigned)(C2 * 64 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(RC ? p.rc_w2[0] : p.w2p), 0, (unsigned)(C2 * 64 * 2), 0x00020000);
    unsigned rowoff[4], rowmask[4], woff1[W1_PIECES], woff2[W2_PIECES], woff3[W3_PIECES ? W3_PIECES : 1], woffsc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)   // rows 8 (SC_PIECES wave + i) + lrow of the 64-row chunk (entry 1 is unused with eight waves)
        woffsc[i] = (unsigned)(own_row(((wave * SC_PIECES + i) & 7) * 8 + lrow) * 64) * 2u + (unsigned)lchunk * 16u;
    {
        const int ohw = p.OH * p.OW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m_base + (wave * 4 + i) * 8 + lrow;
            const bool okm = m < p.M;
            const int mm = okm ? m : 0;
            const int b = fdiv(mm, p.fd_ohw);   // (host-computed reciprocals: a runtime division is ~40 VALU instructions)
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
        for (int i = 0; i < W1_PIECES; ++i)
            woff1[i] = (unsigned)(own_row((wave * W1_PIECES + i) * 8 + lrow) * (9 * C1)) * 2u + (unsigned)lchunk * 16u;
#pragma unroll
        for (int i = 0; i < W2_PIECES; ++i) {
            const int q = wave * W2_PIECES + i;   // sub-tile q>>3 (64 k each), rows 8*(q&7)..+7 of the 64-row chunk
            woff2[i] = (unsigned)(own_row((q & 7) * 8 + lrow) * C1 + (q >> 3) * 64) * 2u + (unsigned)lchunk * 16u;
        }
#pragma unroll
        for (int i = 0; i < W3_PIECES; ++i)
            woff3[i] = (unsigned)(own_row((wave * W3_PIECES + i) * 8 + lrow) * C2) * 2u + (unsigned)lchunk * 16u;
    }

    constexpr int kpc = C1 / 64;   // k-steps per filter tap
    constexpr int nk = 9 * kpc;
    int tap_kh = 0, tap_kw = 0, tap_c = 0;
    auto issue_main = [&](int ks, int buf) {
        unsigned char* As = smem + buf * STAGE_BYTES;
        unsigned char* Ws = As + A_BYTES;
        const int tap = tap_kh * 3 + tap_kw;
        const int soff_a = ((tap_kh * p.W + tap_kw) * C1 + tap_c * 64) * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned vo = ((rowmask[i] >> tap) & 1u) ? rowoff[i] : 0x80000000u;  // out of range -> zero fill
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void*)(As + (wave * 4 + i) * 1024), 16, vo,
                                                     soff_a, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < W1_PIECES; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w1, (__attribute__((address_space(3))) void*)(Ws + (wave * W1_PIECES + i) * 1024),
                                                     16, woff1[i], ks * 128, 0, 0);
        if (++tap_c == kpc) {
            tap_c = 0;
            if (++tap_kw == 3) { tap_kw = 0; ++tap_kh; }
        }
    };
    auto issue_chunk = [&](int j, int buf) {
        unsigned char* W2s = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < W2_PIECES; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w2, (__attribute__((address_space(3))) void*)(W2s + (wave * W2_PIECES + i) * 1024),
                                                     16, woff2[i], j * (64 * C1 * 2), 0, 0);
        if constexpr (C3 > 0) {
            unsigned char* W3s = W2s + W2C_BYTES;
#pragma unroll
            for (int i = 0; i < W3_PIECES; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w3, (__attribute__((address_space(3))) void*)(W3s + (wave * W3_PIECES + i) * 1024),
                                                         16, woff3[i], j * 128, 0, 0);
        }
        if constexpr (SC) {
            unsigned char* Wscs = W2s + W2C_BYTES + C3 * ROW_BYTES;
#pragma unroll
            for (int i = 0; i < SC_PIECES; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_sc, (__attribute__((address_space(3))) void*)(Wscs + (wave * SC_PIECES + i) * 1024), 16,
                                                         woffsc[i], j * (64 * 64 * 2), 0, 0);
        }
        if constexpr (RC) {   // the previous block's expand and shortcut slices: [64 rows][64 k] each, rows in this kernel's ownership order
            unsigned char* Wr = W2s + W2C_BYTES + C3 * ROW_BYTES;
#pragma unroll
            for (int i = 0; i < SC_PIECES; ++i) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_rw, (__attribute__((address_space(3))) void*)(Wr + (wave * SC_PIECES + i) * 1024), 16,
                                                         woffsc[i], j * (64 * 64 * 2), 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_sc, (__attribute__((address_space(3))) void*)(Wr + 8192 + (wave * SC_PIECES + i) * 1024), 16,
                                                         woffsc[i], j * (64 * 64 * 2), 0, 0);
            }
        }
    };
    // y / z / residual: lane (g, li) moves 8 channels (16 B) at 8g of each 32-channel block, for its pixels li and 16 + li
    const bool pr_ok[2] = {wm0 + li < p.M, wm0 + 16 + li < p.M};
    // y: not stored at all (p.y == null: the consumer rebuilds it), or only where a stride-2 1x1 of the next stage reads it (even oh, ow)
    bool y_ok[2] = {pr_ok[0] && p.y != nullptr, pr_ok[1] && p.y != nullptr};
    if (p.y_stride2) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = wm0 + mt * 16 + li;
            const int r = m - fdiv(m, p.fd_ohw) * (p.OH * p.OW);
            const int oh = fdiv(r, p.fd_ow), ow = r - oh * p.OW;
            y_ok[mt] = y_ok[mt] && !((oh | ow) & 1);
        }
    }
    const size_t pr_row[2] = {(size_t)(wm0 + li) * C2 + g * 8, (size_t)(wm0 + 16 + li) * C2 + g * 8};
    // RDMA: wave-private residual staging: rows of this wave, whole 128-byte rows, swizzled like every other tile
    constexpr int RES_BUF = NW * 4096;
    unsigned char* const res_lds = smem + 2 * STAGE_BYTES + wave * 4096;   // + (j & 1) * RES_BUF for chunk j
    const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<f16_t*>(p.res ? p.res : p.x1), 0, p.res ? (unsigned)((size_t)p.M * C2 * 2) : 0u, 0x00020000);   // rows >= M: zeros
    const unsigned res_voff = (unsigned)((wm0 + lrow) * C2) * 2u + (unsigned)lchunk * 16u;
    auto issue_res = [&](int j) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_r, (__attribute__((address_space(3))) void*)(res_lds + (j & 1) * RES_BUF + i * 1024), 16,
                                                     res_voff + (unsigned)(i * 8 * C2 * 2), j * 128, 0, 0);
    };
    const bool has_res = RC ? true : (!SC && p.res != nullptr && !(p.dbg & 4));
    // Counted waits: `s_waitcnt vmcnt(N)` proves that an LDS-DMA request has landed only if the N operations allowed to stay in flight are
    // YOUNGER LDS-DMA requests.  Stores and loads into registers retire out of order with respect to an older LDS-DMA request
    // (tools/microbench/vmorder.hip: with 4 younger stores, or 4 younger register loads, vmcnt(4) returns while the older request's data is
    // still on its way in > 90 % of the cases; with 4 younger LDS-DMA requests in none), so they must not be counted -- rounds 2-3 did, and
    // were saved only by the operands having been requested a whole chunk step earlier.  Hence: N = the residual pieces of chunk j + 2 when
    // they travel by LDS-DMA, otherwise 0; and the y stores of a step are issued AFTER its wait, so that a vmcnt(0) never waits for them.
    const bool res_dma = RDMA && has_res;
    auto load_res = [&](int j, uint4 (&r)[4]) {
        if constexpr (RDMA) {
            if (has_res) issue_res(j);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // i = 2 q + mt: 32-channel block q of the chunk, pixel half mt
                r[i] = make_uint4(0u, 0u, 0u, 0u);
                if (has_res && pr_ok[i & 1]) r[i] = *reinterpret_cast<const uint4*>(p.res + pr_row[i & 1] + j * 64 + (i >> 1) * 32);
            }
        }
    };

    // ---- 3x3 main loop ------------------------------------------------------------------------------------------------
    const int ks_first = (p.dbg & 1) ? nk - 1 : 0;  // dbg: timing ablations (tools/bench_btail.py --ablate), never set by the model
    if (ks_first) { tap_kh = 2; tap_kw = 2; tap_c = kpc - 1; }
    // SC: the shortcut's input channels of this wave's 32 pixels as B fragments (k in natural order: Wsc is not permuted);
    // loaded in front of everything else, complete at the first barrier's vmcnt(0)
    half8 xs[2][2], a1p[2][2];   // RC: the same for the previous block: its shortcut input and its a1
    if constexpr (SC || RC != 0) {
        const f16_t* xsrc = SC ? p.xs : p.rc_xs;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = wm0 + mt * 16 + li;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                half8 v = {}, u = {};
                if (m < p.M) v = *reinterpret_cast<const half8*>(xsrc + (size_t)m * 64 + kk * 32 + g * 8);
                if constexpr (RC != 0)
                    if (m < p.M) u = *reinterpret_cast<const half8*>(p.rc_a1[0] + (size_t)m * 64 + kk * 32 + g * 8);
                xs[mt][kk] = v;
                a1p[mt][kk] = u;
            }
        }
    }
    stamp(2);
    issue_main(ks_first, ks_first & 1);
    float4v acc1[NT1][2];
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt) {
        const float4v b = *reinterpret_cast<const float4v*>(p.b1 + own_ch(nt, g));
        acc1[nt][0] = b;
        acc1[nt][1] = b;
    }
    OPD_DMA_BARRIER();   // (the bias loads above are younger than the first stage's requests: the compiler's own wait would let them fly)
    uint4 res[3][4];  // residual of chunk j lives in res[j % 3], fetched two chunk steps ahead
    auto compute_main = [&](int buf) {
        const unsigned char* As = smem + buf * STAGE_BYTES;
        const unsigned char* Ws = As + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 xf[2], wf[NT1];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) xf[mt] = *reinterpret_cast<const half8*>(As + swz(wave * 32 + mt * 16 + li, kk * 4 + g));
#pragma unroll
            for (int nt = 0; nt < NT1; ++nt) wf[nt] = *reinterpret_cast<const half8*>(Ws + swz(nt * 16 + li, kk * 4 + g));
#pragma unroll
            for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc1[nt][mt] = OPD_MFMA_16x16x32(wf[nt], xf[mt], acc1[nt][mt]);
        }
    };
#pragma unroll 1
    for (int ks = ks_first; ks + 1 < nk; ++ks) {
        issue_main(ks + 1, (ks + 1) & 1);
        compute_main(ks & 1);
        __syncthreads();
    }
    // last k-step: the free stage buffer receives chunk 0's operands; residual chunks 0 and 1 start their trip
    issue_chunk(0, nk & 1);
    compiler_fence();
    if constexpr (RC == 0) {
        load_res(0, res[0]);
        load_res(1, res[1]);
    }
    compiler_fence();
    compute_main((nk - 1) & 1);
    if (res_dma) wait_vmcnt<4>();   // chunk 0 operands and residual chunk 0 landed (chunk 1's 4 pieces, younger LDS-DMA requests, may still fly)
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    stamp(3);

    // ---- a1 = relu(c1) as fp16 B operands: k-block kk <- accumulator tiles 2kk, 2kk+1 -----------------------------------
    half8 a1[2][KK1];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int kk = 0; kk < KK1; ++kk) {
            float4v u = acc1[2 * kk][mt], v = acc1[2 * kk + 1][mt];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                u[r] = u[r] > 0.f ? u[r] : 0.f;
                v[r] = v[r] > 0.f ? v[r] : 0.f;
            }
            a1[mt][kk] = as_half8(pack2h(u[0], u[1]), pack2h(u[2], u[3]), pack2h(v[0], v[1]), pack2h(v[2], v[3]));
        }

    if (p.a1_out) {   // for the next tail, which rebuilds this block's output from it (lane (g, li): channels 32 kk + 8 g .. + 7 of its two pixels)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int kk = 0; kk < KK1; ++kk)
                if (pr_ok[mt]) *reinterpret_cast<half8*>(p.a1_out + (size_t)(wm0 + mt * 16 + li) * C1 + kk * 32 + g * 8) = a1[mt][kk];
    }

    float4v accz[NT3 ? NT3 : 1][2];
    if constexpr (C3 > 0) {
#pragma unroll
        for (int nt = 0; nt < NT3; ++nt) {
            const float4v b = *reinterpret_cast<const float4v*>(p.b3 + own_ch(nt, g));
            accz[nt][0] = b;
            accz[nt][1] = b;
        }
    }

    // ---- chunk steps: 64 channels of y each ------------------------------------------------------------------------------
    if (p.dbg & 8) return;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int buf = (nk + j) & 1;
        uint4 (&res_cur)[4] = res[j % 3];
        if (j + 1 < NCH) issue_chunk(j + 1, buf ^ 1);
        compiler_fence();
        if constexpr (RDMA) {
            if (has_res) {   // paired layout out of this wave's rows of the staged chunk; then the buffer is free for chunk j+2
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    res_cur[i] = *reinterpret_cast<const uint4*>(res_lds + (j & 1) * RES_BUF + swz((i & 1) * 16 + li, (i >> 1) * 4 + g));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (j + 2 < NCH) issue_res(j + 2);
            }
        } else if constexpr (RC == 0) {
            if (j + 2 < NCH) load_res(j + 2, res[(j + 2) % 3]);
        }
        compiler_fence();
        const unsigned char* W2s = smem + buf * STAGE_BYTES;
        if constexpr (RC != 0) {   // the previous block's chunk, as its own tail computes it (btail_kernel<64, 64, false, SC>): bias, W2' . a1', Wsc . xs
            const unsigned char* Wr = W2s + W2C_BYTES + C3 * ROW_BYTES;
            float4v accr[4][2];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float4v b = *reinterpret_cast<const float4v*>(p.rc_b[0] + j * 64 + own_ch(nt, g));
                accr[nt][0] = b;
                accr[nt][1] = b;
            }
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    half8 wf[4];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) wf[nt] = *reinterpret_cast<const half8*>(Wr + half * 8192 + swz(nt * 16 + li, kk * 4 + g));
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) accr[nt][mt] = OPD_MFMA_16x16x32(wf[nt], half ? xs[mt][kk] : a1p[mt][kk], accr[nt][mt]);
                }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    float4v v0 = accr[2 * q][mt], v1 = accr[2 * q + 1][mt];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v0[r] = v0[r] > 0.f ? v0[r] : 0.f;
                        v1[r] = v1[r] > 0.f ? v1[r] : 0.f;
                    }
                    res_cur[2 * q + mt] = make_uint4(pack2h(v0[0], v0[1]), pack2h(v0[2], v0[3]), pack2h(v1[0], v1[1]), pack2h(v1[2], v1[3]));
                }
        }
        float4v acc2[4][2];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const float4v b = *reinterpret_cast<const float4v*>(p.b2 + j * 64 + own_ch(nt, g));
            acc2[nt][0] = b;
            acc2[nt][1] = b;
        }
#pragma unroll
        for (int kk = 0; kk < KK1; ++kk) {
            half8 wf[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                wf[nt] = *reinterpret_cast<const half8*>(W2s + (kk >> 1) * 8192 + swz(nt * 16 + li, (kk & 1) * 4 + g));
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc2[nt][mt] = OPD_MFMA_16x16x32(wf[nt], a1[mt][kk], acc2[nt][mt]);
        }
        if constexpr (SC) {   // + Wsc[chunk] . xs: the block's shortcut convolution
            const unsigned char* Wscs = W2s + W2C_BYTES + C3 * ROW_BYTES;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                half8 wf[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) wf[nt] = *reinterpret_cast<const half8*>(Wscs + swz(nt * 16 + li, kk * 4 + g));
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc2[nt][mt] = OPD_MFMA_16x16x32(wf[nt], xs[mt][kk], acc2[nt][mt]);
            }
        }
        // residual, ReLU, fp16; store y; keep the fp16 values as the next B operand (block q of the chunk = k-block q of W3's slice)
        half8 yf[2][2];   // [q][mt]
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                float4v v0 = acc2[2 * q][mt], v1 = acc2[2 * q + 1][mt];
                if (has_res) {
                    const uint4 r = res_cur[2 * q + mt];
                    float a, b;
                    unpack2h(r.x, a, b); v0[0] += a; v0[1] += b;
                    unpack2h(r.y, a, b); v0[2] += a; v0[3] += b;
                    unpack2h(r.z, a, b); v1[0] += a; v1[1] += b;
                    unpack2h(r.w, a, b); v1[2] += a; v1[3] += b;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v0[r] = v0[r] > 0.f ? v0[r] : 0.f;
                    v1[r] = v1[r] > 0.f ? v1[r] : 0.f;
                }
                yf[q][mt] = as_half8(pack2h(v0[0], v0[1]), pack2h(v0[2], v0[3]), pack2h(v1[0], v1[1]), pack2h(v1[2], v1[3]));
            }
        if constexpr (C3 > 0) {
            const unsigned char* W3s = W2s + W2C_BYTES;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int nt = 0; nt < NT3; ++nt) {
                    const half8 wf = *reinterpret_cast<const half8*>(W3s + swz(nt * 16 + li, kk * 4 + g));
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) accz[nt][mt] = OPD_MFMA_16x16x32(wf, yf[kk][mt], accz[nt][mt]);
                }
            }
        }
        if (j + 1 < NCH) {
            // chunk j+1's operands (requested at the top of this step) have landed; only the residual pieces of chunk j+2 (LDS-DMA requests
            // younger than the operands') may stay in flight
            if (res_dma && j + 2 < NCH) wait_vmcnt<4>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
        // the y stores go out behind the wait: they fly during the next chunk step
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                if (y_ok[mt] && !(p.dbg & 2)) *reinterpret_cast<half8*>(p.y + pr_row[mt] + j * 64 + q * 32) = yf[q][mt];
        stamp(4 + j);
    }

    // ---- z = relu(c0') -----------------------------------------------------------------------------------------------------
    if constexpr (C3 > 0) {
#pragma unroll
        for (int q = 0; q < NT3 / 2; ++q)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                float4v v0 = accz[2 * q][mt], v1 = accz[2 * q + 1][mt];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v0[r] = v0[r] > 0.f ? v0[r] : 0.f;
                    v1[r] = v1[r] > 0.f ? v1[r] : 0.f;
                }
                if (pr_ok[mt] && !(p.dbg & 2))
                    *reinterpret_cast<uint4*>(p.z + (size_t)(wm0 + mt * 16 + li) * C3 + q * 32 + g * 8) =
                        make_uint4(pack2h(v0[0], v0[1]), pack2h(v0[2], v0[3]), pack2h(v1[0], v1[1]), pack2h(v1[2], v1[3]));
            }
    }
    if constexpr (TRACE) {
        stamp(4 + NCH);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(5 + NCH);
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tstamp[15])::"memory");
        if (threadIdx.x == 0 && p.trace) {
#pragma unroll
            for (int i = 0; i < 16; ++i) p.trace[(size_t)blockIdx.x * 16 + i] = tstamp[i];
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Third block of stage 1 with BOTH previous block outputs rebuilt (BtailParams::rc == 2; round 5).  With rc == 1 the second tail still stores
// its output y1 (274 MB at batch 8) for this tail to read back as its residual.  Here the second tail stores its a1 instead (68 MB), and this
// kernel rebuilds, per 32-channel half chunk,
//     y0 = relu(b0 + W2_0 . a1_0 + Wsc . xs)      (block 0's tail arithmetic)
//     y1 = relu(b1 + W2_1 . a1_1 + y0)            (block 1's)
//     y2 = relu(b2 + W2_2 . a1   + y1)            (its own; a1 from its own 3x3), z += W3 . y2
// from the three 64-channel tensors a1_0, a1_1, xs of its 32 pixels per wave (48 B-operand registers), each product in the order its own tail
// runs it, each y rounded to fp16 where that tail rounds it: same bits (tests/test_kernels_gpu.py: chain test, route 2).  Stage 1 then reads and
// writes 64-channel tensors only, plus the quarter of y2 that the next stage's stride-2 shortcut reads.
// Half chunks: a 64-channel chunk would need 48 KiB of operands per step (four 64 x 64 slices + the 128 x 64 slice of W3), twice that double
// buffered: one workgroup per CU.  A step therefore covers 32 channels: four [32][64] slices (16 KiB, double buffered in the two 3x3 stage
// buffers) while the W3 slice of a chunk (16 KiB) lives for two steps in a double buffer of its own: 80 KiB per workgroup, two per CU.
template <int C3>
__global__ __launch_bounds__(256, 2) void btail_rc2_kernel(BtailParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int C1 = 64, C2 = 256, NT1 = 4, KK1 = 2, NT3 = C3 / 16;
    constexpr int A_BYTES = 128 * ROW_BYTES, STAGE_BYTES = A_BYTES + C1 * ROW_BYTES;   // 24 KiB
    constexpr int QBUF0 = STAGE_BYTES, QBUF1 = 0;                                      // quads of even / odd steps: inside stage 1 / stage 0
    constexpr int W3BUF = 2 * STAGE_BYTES;                                             // + (chunk & 1) * 16 KiB
    static_assert(C3 == 128, "the W3 double buffer is laid out for 128 reduce channels");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int m_base = (p.rev ? xcd_logical_block_rev(blockIdx.x, gridDim.x) : xcd_logical_block(blockIdx.x, gridDim.x)) * 128;
    const int wm0 = m_base + wave * 32;

    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ lrow;
    const unsigned backoff = (unsigned)(p.W + 1) * (unsigned)C1 * 2u;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x1)) - backoff, 0, (unsigned)((size_t)p.B * p.H * p.W * C1 * 2) + backoff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.w1), 0, (unsigned)(C1 * 9 * C1 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.w3p), 0, (unsigned)(C3 * C2 * 2), 0x00020000);
    // this wave's [32][64] slice of every step: wave 0 block 0's expand, 1 block 0's shortcut, 2 block 1's expand, 3 this block's expand
    const f16_t* const qsrc = wave == 0 ? p.rc_w2[1] : wave == 1 ? p.rc_wsc : wave == 2 ? p.rc_w2[0] : p.w2p;
    const __amdgpu_buffer_rsrc_t rsrc_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(qsrc), 0, (unsigned)(C2 * 64 * 2), 0x00020000);
    unsigned rowoff[4], rowmask[4], woff1[2], woffq[4], woff3[4];
    {
        const int ohw = p.OH * p.OW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m_base + (wave * 4 + i) * 8 + lrow;
            const bool okm = m < p.M;
            const int mm = okm ? m : 0;
            const int b = fdiv(mm, p.fd_ohw);
            const int r = mm - b * ohw;
            const int oh = fdiv(r, p.fd_ow);
            const int ow = r - oh * p.OW;
            rowoff[i] = (unsigned)(((b * p.H + oh) * p.W + ow) * C1) * 2u + (unsigned)lchunk * 16u;
            unsigned kwmask = 0, mask = 0;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
                if ((unsigned)(ow - 1 + kw) < (unsigned)p.W) kwmask |= 1u << kw;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
                if ((unsigned)(oh - 1 + kh) < (unsigned)p.H) mask |= kwmask << (kh * 3);
            rowmask[i] = okm ? mask : 0u;
            // quad piece i: LDS rows 8 i + lrow of the 32-row block <- channel own_row(.) of the block (the block's base travels as the scalar offset)
            woffq[i] = (unsigned)(own_row(i * 8 + lrow) * 64) * 2u + (unsigned)lchunk * 16u;
            woff3[i] = (unsigned)(own_row((wave * 4 + i) * 8 + lrow) * C2) * 2u + (unsigned)lchunk * 16u;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) woff1[i] = (unsigned)(own_row((wave * 2 + i) * 8 + lrow) * (9 * C1)) * 2u + (unsigned)lchunk * 16u;
    }
    constexpr int nk = 9;
    int tap_kh = 0, tap_kw = 0;
    auto issue_main = [&](int ks, int buf) {
        unsigned char* As = smem + buf * STAGE_BYTES;
        unsigned char* Ws = As + A_BYTES;
        const int tap = tap_kh * 3 + tap_kw;
        const int soff_a = ((tap_kh * p.W + tap_kw) * C1) * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned vo = ((rowmask[i] >> tap) & 1u) ? rowoff[i] : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void*)(As + (wave * 4 + i) * 1024), 16, vo, soff_a, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w1, (__attribute__((address_space(3))) void*)(Ws + (wave * 2 + i) * 1024), 16, woff1[i], ks * 128, 0, 0);
        if (++tap_kw == 3) { tap_kw = 0; ++tap_kh; }
    };
    auto issue_quads = [&](int h) {   // step h = (chunk h >> 1, half h & 1): rows 32 h .. 32 h + 31 of the four [256][64] matrices
        unsigned char* dst = smem + ((h & 1) ? QBUF1 : QBUF0) + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_q, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, woffq[i], h * (32 * 64 * 2), 0, 0);
    };
    auto issue_w3 = [&](int j) {
        unsigned char* dst = smem + W3BUF + (j & 1) * 16384;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w3, (__attribute__((address_space(3))) void*)(dst + (wave * 4 + i) * 1024), 16, woff3[i], j * 128, 0, 0);
    };
    const bool pr_ok[2] = {wm0 + li < p.M, wm0 + 16 + li < p.M};
    const size_t pr_row[2] = {(size_t)(wm0 + li) * C2 + g * 8, (size_t)(wm0 + 16 + li) * C2 + g * 8};
    bool y_ok[2] = {pr_ok[0] && p.y != nullptr, pr_ok[1] && p.y != nullptr};
    if (p.y_stride2) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = wm0 + mt * 16 + li;
            const int r = m - fdiv(m, p.fd_ohw) * (p.OH * p.OW);
            const int oh = fdiv(r, p.fd_ow), ow = r - oh * p.OW;
            y_ok[mt] = y_ok[mt] && !((oh | ow) & 1);
        }
    }
    // the three 64-channel inputs of the rebuild for this wave's 32 pixels, as B fragments (natural k order); loaded in front of everything else
    half8 xs[2][2], a10[2][2], a11[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = wm0 + mt * 16 + li;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 v = {}, u0 = {}, u1 = {};
            if (m < p.M) {
                v = *reinterpret_cast<const half8*>(p.rc_xs + (size_t)m * 64 + kk * 32 + g * 8);
                u0 = *reinterpret_cast<const half8*>(p.rc_a1[1] + (size_t)m * 64 + kk * 32 + g * 8);
                u1 = *reinterpret_cast<const half8*>(p.rc_a1[0] + (size_t)m * 64 + kk * 32 + g * 8);
            }
            xs[mt][kk] = v; a10[mt][kk] = u0; a11[mt][kk] = u1;
        }
    }
    issue_main(0, 0);
    float4v acc1[NT1][2];
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt) {
        const float4v b = *reinterpret_cast<const float4v*>(p.b1 + own_ch(nt, g));
        acc1[nt][0] = b;
        acc1[nt][1] = b;
    }
    OPD_DMA_BARRIER();
    auto compute_main = [&](int buf) {
        const unsigned char* As = smem + buf * STAGE_BYTES;
        const unsigned char* Ws = As + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 xf[2], wf[NT1];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) xf[mt] = *reinterpret_cast<const half8*>(As + swz(wave * 32 + mt * 16 + li, kk * 4 + g));
#pragma unroll
            for (int nt = 0; nt < NT1; ++nt) wf[nt] = *reinterpret_cast<const half8*>(Ws + swz(nt * 16 + li, kk * 4 + g));
#pragma unroll
            for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc1[nt][mt] = OPD_MFMA_16x16x32(wf[nt], xf[mt], acc1[nt][mt]);
        }
    };
#pragma unroll 1
    for (int ks = 0; ks + 1 < nk; ++ks) {
        issue_main(ks + 1, (ks + 1) & 1);
        compute_main(ks & 1);
        __syncthreads();
    }
    // last k-step (stage 0): stage 1 is free -> the quads of step 0; the W3 buffers lie behind both stages
    issue_quads(0);
    issue_w3(0);
    compiler_fence();
    compute_main((nk - 1) & 1);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    half8 a1[2][KK1];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int kk = 0; kk < KK1; ++kk) {
            float4v u = acc1[2 * kk][mt], v = acc1[2 * kk + 1][mt];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                u[r] = u[r] > 0.f ? u[r] : 0.f;
                v[r] = v[r] > 0.f ? v[r] : 0.f;
            }
            a1[mt][kk] = as_half8(pack2h(u[0], u[1]), pack2h(u[2], u[3]), pack2h(v[0], v[1]), pack2h(v[2], v[3]));
        }
    float4v accz[NT3][2];
#pragma unroll
    for (int nt = 0; nt < NT3; ++nt) {
        const float4v b = *reinterpret_cast<const float4v*>(p.b3 + own_ch(nt, g));
        accz[nt][0] = b;
        accz[nt][1] = b;
    }
    // one [32][64] slice times a 64-channel B operand into two accumulator tiles per pixel half
    auto mma_slice = [&](const unsigned char* W, const half8 (&bop)[2][2], float4v (&acc)[2][2]) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 wf[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) wf[nt] = *reinterpret_cast<const half8*>(W + swz(nt * 16 + li, kk * 4 + g));
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = OPD_MFMA_16x16x32(wf[nt], bop[mt][kk], acc[nt][mt]);
        }
    };
    auto init_bias = [&](const float* bias, const int ch0, float4v (&acc)[2][2]) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float4v b = *reinterpret_cast<const float4v*>(bias + ch0 + g * 8 + nt * 4);
            acc[nt][0] = b;
            acc[nt][1] = b;
        }
    };
    // + residual (the previous block's rounded output, 8 consecutive channels per lane), ReLU, fp16
    auto finish = [&](float4v (&acc)[2][2], const uint4 (*res)[2], uint4 (&out)[2]) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float4v v0 = acc[0][mt], v1 = acc[1][mt];
            if (res) {
                const uint4 r = (*res)[mt];
                float a, b;
                unpack2h(r.x, a, b); v0[0] += a; v0[1] += b;
                unpack2h(r.y, a, b); v0[2] += a; v0[3] += b;
                unpack2h(r.z, a, b); v1[0] += a; v1[1] += b;
                unpack2h(r.w, a, b); v1[2] += a; v1[3] += b;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v0[r] = v0[r] > 0.f ? v0[r] : 0.f;
                v1[r] = v1[r] > 0.f ? v1[r] : 0.f;
            }
            out[mt] = make_uint4(pack2h(v0[0], v0[1]), pack2h(v0[2], v0[3]), pack2h(v1[0], v1[1]), pack2h(v1[2], v1[3]));
        }
    };
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        const int j = h >> 1, q = h & 1, ch0 = h * 32;
        if (h + 1 < 8) {
            issue_quads(h + 1);
            if (q == 1) issue_w3(j + 1);   // (its buffer was last read two steps ago)
        }
        compiler_fence();
        const unsigned char* Q = smem + (q ? QBUF1 : QBUF0);
        float4v acc[2][2];
        uint4 r0[2], r1[2], y2[2];
        init_bias(p.rc_b[1], ch0, acc);            // block 0: bias (expand + shortcut), W2_0 . a1_0, Wsc . xs
        mma_slice(Q, a10, acc);
        mma_slice(Q + 4096, xs, acc);
        finish(acc, nullptr, r0);
        init_bias(p.rc_b[0], ch0, acc);            // block 1: bias, W2_1 . a1_1, + y0
        mma_slice(Q + 8192, a11, acc);
        finish(acc, &r0, r1);
        init_bias(p.b2, ch0, acc);                 // this block: bias, W2_2 . a1, + y1
        mma_slice(Q + 12288, a1, acc);
        finish(acc, &r1, y2);
        half8 yf[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) yf[mt] = as_half8(y2[mt].x, y2[mt].y, y2[mt].z, y2[mt].w);
        {   // z += W3[:, 64 j + 32 q ..] . y2: k-block q of the chunk's slice
            const unsigned char* W3s = smem + W3BUF + (j & 1) * 16384;
#pragma unroll
            for (int nt = 0; nt < NT3; ++nt) {
                const half8 wf = *reinterpret_cast<const half8*>(W3s + swz(nt * 16 + li, q * 4 + g));
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) accz[nt][mt] = OPD_MFMA_16x16x32(wf, yf[mt], accz[nt][mt]);
            }
        }
        if (h + 1 < 8) {
            wait_vmcnt<0>();   // the next step's operands (requested at the top) have landed; the y stores of this step go out behind the wait
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            if (y_ok[mt]) *reinterpret_cast<half8*>(p.y + pr_row[mt] + ch0) = yf[mt];
    }
#pragma unroll
    for (int qz = 0; qz < NT3 / 2; ++qz)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float4v v0 = accz[2 * qz][mt], v1 = accz[2 * qz + 1][mt];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v0[r] = v0[r] > 0.f ? v0[r] : 0.f;
                v1[r] = v1[r] > 0.f ? v1[r] : 0.f;
            }
            if (pr_ok[mt])
                *reinterpret_cast<uint4*>(p.z + (size_t)(wm0 + mt * 16 + li) * C3 + qz * 32 + g * 8) =
                    make_uint4(pack2h(v0[0], v0[1]), pack2h(v0[2], v0[3]), pack2h(v1[0], v1[1]), pack2h(v1[2], v1[3]));
        }
#endif
}

template <int C1, int C3, bool RDMA, bool SC = false, bool TRACE = false, int RC = 0, int NW = 4>
hipError_t launch_btail_t(const BtailParams& p, hipStream_t stream) {
    constexpr int BM = 32 * NW;
    constexpr int MAIN = (BM + C1) * ROW_BYTES, CHUNK = 64 * C1 * 2 + C3 * ROW_BYTES + (SC ? 64 * ROW_BYTES : 0) + (RC ? 2 * 64 * ROW_BYTES : 0);
    constexpr int LDS = 2 * (MAIN > CHUNK ? MAIN : CHUNK) + (RDMA ? 2 * NW * 4096 : 0);
    static_assert(LDS <= 160 * 1024, "LDS per workgroup");
    OPD_SET_MAX_LDS_ONCE((btail_kernel<C1, C3, RDMA, SC, TRACE, RC, NW>), LDS);
    OPD_LAUNCH((btail_kernel<C1, C3, RDMA, SC, TRACE, RC, NW>), dim3((p.M + BM - 1) / BM), dim3(64 * NW), LDS, stream, p);
    static const char* const kname = opd_kernel_name("btail_kernel<%d, %d, %s, %s, %s, %d, %d>", C1, C3, OPD_BOOLSTR(RDMA), OPD_BOOLSTR(SC), OPD_BOOLSTR(TRACE), RC, NW);
    opd_last_kernel_name = kname;
    return hipGetLastError();
}

}  // namespace

#ifndef OPD_ELEM_BF16   // (shape predicates and the host-side weight permutation do not depend on the element type: defined once)
bool opd_btail_supported(int C1, int C3) {
    return (C1 == 64 && (C3 == 0 || C3 == 64 || C3 == 128)) || (C1 == 128 && (C3 == 0 || C3 == 128)) || opd_btail256_supported(C1, C3);
}
#endif

hipError_t OPD_SYM(opd_launch_btail)(const BtailParams& p_in, hipStream_t stream) {
    if (!opd_btail_supported(p_in.C1, p_in.C3) || p_in.OH <= 0 || p_in.OW <= 0) return hipErrorInvalidValue;
    BtailParams p = p_in;
    p.fd_ohw = opd_make_fastdiv((unsigned)p.OH * (unsigned)p.OW);
    p.fd_ow = opd_make_fastdiv((unsigned)p.OW);
    // 31-bit byte offsets in the buffer descriptors
    if ((size_t)p.B * p.H * p.W * p.C1 * 2 + (size_t)(p.W + 1) * p.C1 * 2 >= 0x7fffff00ull) return hipErrorInvalidValue;
    if ((size_t)p.M * p.C1 * 8 >= 0x7fffff00ull) return hipErrorInvalidValue;
    if (p.C1 == 256) return OPD_SYM(opd_launch_btail256)(p, stream);   // stage 3: kernels_btail3.hip
    if (p.y_stride2 && (p.stride != 1 || p.C3 == 0)) return hipErrorInvalidValue;   // (only next to a fused reduce: nobody else may need y)
    if (!p.y && !p.a1_out) return hipErrorInvalidValue;   // an output nobody could rebuild
    if (p.nw != 0 && p.nw != 4 && p.nw != 8) return hipErrorInvalidValue;
    if (p.nw == 8 && (p.trace || p.rc == 2)) return hipErrorInvalidValue;   // (the traced and the two-level instantiations exist with four waves only)
    if (p.rc == 2) {   // third block of stage 1: both previous outputs rebuilt (btail_rc2_kernel)
        if (p.C1 != 64 || p.C3 != 128 || p.stride != 1 || p.res || p.xs || !p.rc_a1[0] || !p.rc_a1[1] || !p.rc_xs || !p.rc_w2[0] || !p.rc_w2[1] || !p.rc_wsc ||
            !p.rc_b[0] || !p.rc_b[1] || !p.w3p || !p.z || p.trace || p.dbg || (size_t)p.M * 64 * 2 >= 0x7fffff00ull)
            return hipErrorInvalidValue;
        constexpr int LDS = 2 * (128 + 64) * ROW_BYTES + 2 * 16384;
        OPD_SET_MAX_LDS_ONCE(btail_rc2_kernel<128>, LDS);
        OPD_LAUNCH(btail_rc2_kernel<128>, dim3((p.M + 127) / 128), dim3(256), LDS, stream, p);
        return hipGetLastError();
    }
    if (p.rc) {   // residual rebuilt from the previous block's a1 and shortcut input (second block of stage 1)
        if (p.rc != 1 || p.C1 != 64 || p.C3 != 64 || p.stride != 1 || p.res || p.xs || !p.rc_a1[0] || !p.rc_xs || !p.rc_w2[0] || !p.rc_wsc || !p.rc_b[0] ||
            p.trace || (size_t)p.M * 64 * 2 >= 0x7fffff00ull)
            return hipErrorInvalidValue;
        return p.nw == 8 ? launch_btail_t<64, 64, false, false, false, 1, 8>(p, stream) : launch_btail_t<64, 64, false, false, false, 1>(p, stream);
    }
    if (p.xs) {   // fused shortcut convolution: stride 1, 64 -> 256 channels next to a 64-channel 3x3 (first block of stage 1)
        if (p.C1 != 64 || p.C3 != 64 || p.stride != 1 || !p.wsc || p.res || (size_t)p.M * 64 * 2 >= 0x7fffff00ull) return hipErrorInvalidValue;
        return p.nw == 8 ? launch_btail_t<64, 64, false, true, false, 0, 8>(p, stream) : launch_btail_t<64, 64, false, true>(p, stream);
    }
    if (p.trace) {   // tools/trace_btail.py: the two shapes that dominate stages 1 and 2
        if (p.C1 == 64 && p.C3 == 64 && (p.dbg & 16)) return launch_btail_t<64, 64, false, false, true>(p, stream);
        if (p.C1 == 64 && p.C3 == 64) return launch_btail_t<64, 64, true, false, true>(p, stream);
        if (p.C1 == 128 && p.C3 == 128) return launch_btail_t<128, 128, false, false, true>(p, stream);
        return hipErrorInvalidValue;
    }
    if (p.C1 == 64) {
        const bool rdma = !(p.dbg & 16);   // dbg 16: residual through VGPR loads (the first form: cross-check / timing)
        if (rdma && p.nw == 8) {           // 256-pixel tiles, eight waves (BtailParams::nw)
            if (p.C3 == 0) return launch_btail_t<64, 0, true, false, false, 0, 8>(p, stream);
            if (p.C3 == 64) return launch_btail_t<64, 64, true, false, false, 0, 8>(p, stream);
            return launch_btail_t<64, 128, true, false, false, 0, 8>(p, stream);
        }
        if (p.C3 == 0) return rdma ? launch_btail_t<64, 0, true>(p, stream) : launch_btail_t<64, 0, false>(p, stream);
        if (p.C3 == 64) return rdma ? launch_btail_t<64, 64, true>(p, stream) : launch_btail_t<64, 64, false>(p, stream);
        return rdma ? launch_btail_t<64, 128, true>(p, stream) : launch_btail_t<64, 128, false>(p, stream);
    }
    if (p.nw == 8) return p.C3 == 0 ? launch_btail_t<128, 0, false, false, false, 0, 8>(p, stream) : launch_btail_t<128, 128, false, false, false, 0, 8>(p, stream);
    if (p.C3 == 0) return launch_btail_t<128, 0, false>(p, stream);
    return launch_btail_t<128, 128, false>(p, stream);
}

#ifndef OPD_ELEM_BF16
// K-permutation of a [rows][K] fp16 weight matrix (K % 32 == 0) that makes two fp16-rounded 16x16 accumulator tiles a
// valid B operand: within each 32-block, slot 8g+e <- channel 4g+e, slot 8g+4+e <- channel 16+4g+e.
void opd_permute_k32(const f16_t* w, f16_t* out, int rows, int K) {
    for (int n = 0; n < rows; ++n)
        for (int b = 0; b < K; b += 32)
            for (int g = 0; g < 4; ++g)
                for (int e = 0; e < 4; ++e) {
                    out[(size_t)n * K + b + 8 * g + e] = w[(size_t)n * K + b + 4 * g + e];
                    out[(size_t)n * K + b + 8 * g + 4 + e] = w[(size_t)n * K + b + 16 + 4 * g + e];
                }
}
#endif
