// kernels_w8.hip — eight-wave implicit-GEMM convolution / linear layer for gfx950: 128 (m) x 256 (n) x 64 (k) tiles, three-stage LDS-DMA ring.
//
// SURVEY.md §8(a) row a4 (bottleneck 1x1 / 3x3 convolutions, FrozenBN folded, residual + ReLU in the epilogue).  Reference arithmetic:
// HF:models/resnet/modeling_resnet.py:139-178.  Same formulation and operand layouts as kernels_gemm.hip (out[m][n] = sum_k A[m][k] Wt[n][k],
// m = (b, oh, ow), k = (kh, kw, cin), NHWC fp16, weights [N][K]); what differs is the shape of a workgroup.
//
// Why a second GEMM kernel (round 5).  conv_gemm_dma_kernel runs 4-wave workgroups on 160 x 64 / 160 x 128 tiles, two per CU, one tile of
// prefetch, a drain + barrier per k-step.  With three batches in flight the whole-forward ablations (tools/abl_forward.sh) show what the
// headline rate is sensitive to: NOT the MFMAs of those launches (skipping them: +0 %), but their staging (skipping the tile DMA: +11 %,
// stage 4 alone 0.46 -> 0.30 ms).  Stage 4's 3x3 convolutions stage 855 MB per launch through the LDS-DMA path for 39.6 GFLOP: eight
// 64-column tiles each re-stage the im2col'd activation tile, and every CU of the chip is occupied for 66 us.  The stage-3 tail's 3x3 loop
// (kernels_btail3.hip) is the other shape: 128 x 256 tiles, eight waves, one workgroup per CU, three stages with two k-steps of DMA in flight,
// counted waits, two wave groups staggered by a barrier so that one group's LDS reads, request issue and barrier waits run beside the other
// group's MFMAs -- 1630 clocks per k-step of 4.2 MFLOP, i.e. 5.4 TFLOP/s per CU against 2.3 for the 4-wave kernel on this layer.  This file
// is that loop as a launch of its own: a 3x3 of stage 4 becomes 132 workgroups (half the CUs, 465 MB staged) that each run 72 k-steps; the
// other CUs belong to the other streams meanwhile.  Per accumulator element the k order and the MFMA order are those of
// conv_gemm_dma_kernel, so the two kernels give IDENTICAL bits (tests/test_kernels_gpu.py asserts it): which one runs a layer is a pure
// speed choice, made per handle configuration.
#include <hip/hip_runtime.h>
#include "opd_kernels.h"
#include "opd_elem.h"

typedef elem_t half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));

namespace {

constexpr int ROW_BYTES = 128;
constexpr int W8_BM = 128, W8_BN = 256;
constexpr int W8_A_BYTES = W8_BM * ROW_BYTES;                 // 16 KiB
constexpr int W8_STAGE = W8_A_BYTES + W8_BN * ROW_BYTES;      // 48 KiB
constexpr int W8_LDS = 3 * W8_STAGE;                          // 144 KiB: one workgroup per CU

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROW_BYTES + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ int xcd_logical_block(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7;
    const int x = bid & 7, k = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
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
template <int N_OUTSTANDING>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N_OUTSTANDING >= 0 && N_OUTSTANDING <= 63, "vmcnt range");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_OUTSTANDING) : "memory");
}
__device__ __forceinline__ void compiler_fence() { asm volatile("" ::: "memory"); }
// LDS reads / writes of this wave retired, then the workgroup barrier; nothing moves across it
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// PW: 1x1 stride-1 convolution / linear layer (row m of [M][Cin]; no tap bookkeeping).
// Requires (checked by the launcher): N % 256 == 0, Cin % 64 == 0, K / 64 >= 3, byte offsets below 2^31.
template <bool PW>
__global__ __launch_bounds__(512) void conv_w8_kernel(ConvGemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = wave >> 1, half = wave & 1;      // the wave's 32 pixels (of 128), its 128 output channels (of 256)
    const int g = lane >> 4, li = lane & 15;
    const int tiles_n = p.N / W8_BN;
    // XCD-aware tile order: the column tiles of a row tile run on one XCD back to back (they re-read the same activation rows)
    const int lbid = xcd_logical_block(blockIdx.x, gridDim.x);
    const int tile_m = fdiv(lbid, p.fd_tilesn), tile_n = lbid - tile_m * tiles_n;
    const int m_base = tile_m * W8_BM, n_base = tile_n * W8_BN;

    // ---- staging coordinates: piece = 8 tile rows x 128 B, lane -> (row lane >> 3, 16-byte slot lane & 7), XOR swizzle on the SOURCE side.
    //      Wave w requests pixel pieces 2 w, 2 w + 1 and weight pieces 4 w .. 4 w + 3 of every k-step.
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ lrow;
    const unsigned backoff = PW ? 0u : (unsigned)(p.pad * p.W + p.pad) * (unsigned)p.Cin * 2u;   // every in-image tap gets a non-negative offset
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x)) - backoff, 0, (unsigned)((size_t)p.B * p.H * p.W * p.Cin * 2) + backoff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.w), 0, (unsigned)((size_t)p.N * p.K * 2), 0x00020000);
    unsigned rowoff[2], rowmask[2], woff[4];
    {
        const int ohw = p.OH * p.OW;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m_base + (wave * 2 + i) * 8 + lrow;
            const bool okm = m < p.M;
            if constexpr (PW) {
                rowoff[i] = okm ? (unsigned)m * (unsigned)(p.Cin * 2) + (unsigned)lchunk * 16u : 0x80000000u;   // rows >= M: zeros (bounds check)
                rowmask[i] = 1u;
                continue;
            }
            const int mm = okm ? m : 0;
            const int b = fdiv(mm, p.fd_ohw);
            const int r = mm - b * ohw;
            const int oh = fdiv(r, p.fd_ow);
            const int ow = r - oh * p.OW;
            rowoff[i] = (unsigned)(((b * p.H + oh * p.stride) * p.W + ow * p.stride) * p.Cin) * 2u + (unsigned)lchunk * 16u;
            // valid taps in closed form (kernels_gemm.hip): contiguous ranges of kw and kh, replicated by p.tap_rep = sum 1 << kh * KW
            const int iw0 = ow * p.stride - p.pad, ih0 = oh * p.stride - p.pad;
            const int lo_w = max(0, -iw0), hi_w = min(p.KW - 1, p.W - 1 - iw0);
            const int lo_h = max(0, -ih0), hi_h = min(p.KH - 1, p.H - 1 - ih0);
            auto below = [](const int n) { return n > 0 ? 0xffffffffu >> (32 - n) : 0u; };
            const unsigned kwmask = hi_w >= lo_w ? below(hi_w + 1) & ~below(lo_w) : 0u;
            const unsigned hsel = hi_h >= lo_h ? below((hi_h + 1) * p.KW) & ~below(lo_h * p.KW) : 0u;
            rowmask[i] = okm ? kwmask * (p.tap_rep & hsel) : 0u;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            woff[i] = (unsigned)((n_base + (wave * 4 + i) * 8 + lrow) * p.K) * 2u + (unsigned)lchunk * 16u;
    }
    const int kpc = p.Cin / 64;      // k-steps per filter tap
    const int nk = p.K / 64;
    int tap_kh = 0, tap_kw = 0, tap_c = 0;
    // Requests go out one piece at a time between groups of MFMAs, never as a burst (kernels_btail3.hip: eight waves that all issue six
    // requests behind a barrier stand in the CU's address queue for ~740 clocks with the matrix pipe idle).
    int is_tap = 0, is_soff_a = 0, is_ks = 0;
    auto begin_issue = [&](int ks) {   // k-steps are requested in order: the tap counters advance by one per call
        is_ks = ks;
        if constexpr (PW) {
            is_tap = 0;
            is_soff_a = ks * 128;
        } else {
            is_tap = tap_kh * p.KW + tap_kw;
            is_soff_a = ((tap_kh * p.W + tap_kw) * p.Cin + tap_c * 64) * 2;
            if (++tap_c == kpc) {
                tap_c = 0;
                if (++tap_kw == p.KW) { tap_kw = 0; ++tap_kh; }
            }
        }
    };
    auto issue_piece = [&](int i, int stage_off) {   // i = 0, 1: pixel rows; 2 .. 5: weight rows
        unsigned char* As = smem + stage_off;
        if (i < 2) {
            const unsigned vo = ((rowmask[i] >> is_tap) & 1u) ? rowoff[i] : 0x80000000u;   // out of image -> zero fill
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void*)(As + (wave * 2 + i) * 1024), 16, vo, is_soff_a, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (__attribute__((address_space(3))) void*)(As + W8_A_BYTES + (wave * 4 + i - 2) * 1024), 16,
                                                     woff[i - 2], is_ks * 128, 0, 0);
        }
    };
    auto issue_main = [&](int ks, int stage_off) {
        begin_issue(ks);
#pragma unroll
        for (int i = 0; i < 6; ++i) issue_piece(i, stage_off);
    };

    // ---- main loop: wave = (pixels 32 pair .., channels 128 half ..): 8 x 2 accumulator tiles ------------------------------------------
    issue_main(0, 0);
    issue_main(1, W8_STAGE);
    const int wn0 = n_base + half * 128, wm0 = m_base + pair * 32;
    float4v acc[8][2];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
        const float4v b = *reinterpret_cast<const float4v*>(p.bias + wn0 + nt * 16 + g * 4);
        acc[nt][0] = b;
        acc[nt][1] = b;
    }
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) asm volatile("" : "+v"(acc[nt][0]), "+v"(acc[nt][1]));   // (the bias loads retire here, before the counted waits)
    // Two wave groups staggered by one barrier (waves 0-3 / 4-7; SIMD s hosts waves s and s + 4, one of each group).  A k-step is
    //     P1(k): read k-step k's fragments (stage k % 3) -> request the first two pieces of k-step k + 2 (stage (k + 2) % 3) -> vmcnt(2)
    //            (retires this wave's pieces of stage k + 1: only the two just issued may fly) -> lgkmcnt(0) -> barrier
    //     M(k):  32 MFMAs at raised priority with the other four pieces of k-step k + 2 between them -> lgkmcnt(0) -> barrier
    // and group 1 runs one extra barrier up front, so one group's P1 sits beside the other's M on every SIMD.  Barriers B_i; group 0 runs
    // P1(k) in front of B_2k and M(k) behind it, group 1 one barrier later:
    //   RAW  a wave reads stage k after B_(2k-1) at the earliest; every wave retired its pieces of stage k in P1(k-1), in front of B_(2k-2)
    //        (group 0) or B_(2k-1) (group 1);
    //   WAR  stage (k + 2) % 3 = stage (k - 1) % 3 is requested after B_(2k-1) at the earliest; its last readers are the P1(k-1) of both
    //        groups, whose lgkmcnt(0) sits in front of B_(2k-2) resp. B_(2k-1).
    // Every counted wait counts LDS-DMA requests only (opd_kernels.h, OPD_DMA_BARRIER: register loads and stores retire out of order with
    // respect to them): the bias loads are retired above, the residual loads go out behind the last counted wait.
    const int group = wave >> 2;
    half8 xf[2][2], wf[2][8];
    auto read_frags = [&](int stage_off) {
        const unsigned char* As = smem + stage_off;
        const unsigned char* Ws = As + W8_A_BYTES;
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
                for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = OPD_MFMA_16x16x32(wf[kk][nt], xf[kk][mt], acc[nt][mt]);
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
    wait_vmcnt<6>();   // stage 0
    lds_barrier();
    if (group == 1) lds_barrier();   // the stagger
    int st_cur = 0, st_next2 = 2 * W8_STAGE;
    auto rotate = [&]() {
        st_cur = st_cur == 2 * W8_STAGE ? 0 : st_cur + W8_STAGE;
        st_next2 = st_next2 == 2 * W8_STAGE ? 0 : st_next2 + W8_STAGE;
    };
#pragma unroll 1
    for (int ks = 0; ks + 2 < nk; ++ks) {
        read_frags(st_cur);
        compiler_fence();
        begin_issue(ks + 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) issue_piece(i, st_next2);
        wait_vmcnt<2>();
        lds_barrier();
        mfma_phase([&](int slot) { issue_piece(2 + slot, st_next2); });
        rotate();
    }
    // k-step nk - 2: nothing left to request; the wait retires stage nk - 1
    read_frags(st_cur);
    compiler_fence();
    wait_vmcnt<0>();
    lds_barrier();
    mfma_phase(no_issue);
    rotate();
    // k-step nk - 1: the fp16 residual (paired 16-byte layout, below) is requested here -- behind the last counted wait -- and lands during the MFMAs
    read_frags(st_cur);
    compiler_fence();
    uint4 res[8];
    const int my_m = wm0 + (g & 1) * 16 + li;          // paired layout: even g -> the wave's first 16 pixels, odd g -> its second 16
    const bool my_ok = my_m < p.M;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
        res[nt] = make_uint4(0u, 0u, 0u, 0u);
        if (p.res16 && my_ok) res[nt] = *reinterpret_cast<const uint4*>(p.res16 + (size_t)my_m * p.N + wn0 + nt * 16 + (g >> 1) * 8);
    }
    compiler_fence();
    lds_barrier();
    mfma_phase(no_issue);
    if (group == 0) lds_barrier();   // the groups are aligned again: every wave has passed the same number of barriers

    // ---- epilogue in registers (kernels_gemm.hip::epilogue_regs, direct form): + residual -> ReLU -> fp16; v_permlane16_swap pairs the 4-channel
    //      quads of the wave's two pixel tiles so that every lane loads / stores 8 consecutive channels (16 bytes) of one pixel
    f16_t* const o16 = reinterpret_cast<f16_t*>(p.out);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
        float4v v[2] = {acc[nt][0], acc[nt][1]};
        if (p.res16) {
            const uint4 r = res[nt];
            const uint2v s0 = __builtin_amdgcn_permlane16_swap(r.x, r.z, false, false);
            const uint2v s1 = __builtin_amdgcn_permlane16_swap(r.y, r.w, false, false);
            float a, b;
            unpack2h(s0[0], a, b); v[0][0] += a; v[0][1] += b;
            unpack2h(s1[0], a, b); v[0][2] += a; v[0][3] += b;
            unpack2h(s0[1], a, b); v[1][0] += a; v[1][1] += b;
            unpack2h(s1[1], a, b); v[1][2] += a; v[1][3] += b;
        }
        if (p.relu) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[h][r] = v[h][r] > 0.f ? v[h][r] : 0.f;
        }
        const uint2v s0 = __builtin_amdgcn_permlane16_swap(pack2h(v[0][0], v[0][1]), pack2h(v[1][0], v[1][1]), false, false);
        const uint2v s1 = __builtin_amdgcn_permlane16_swap(pack2h(v[0][2], v[0][3]), pack2h(v[1][2], v[1][3]), false, false);
        if (my_ok) *reinterpret_cast<uint4*>(o16 + (size_t)my_m * p.N + wn0 + nt * 16 + (g >> 1) * 8) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
    }
#endif
}

}  // namespace

#ifndef OPD_ELEM_BF16
// shape contract of the eight-wave kernel (the caller falls back to opd_launch_conv_gemm otherwise)
bool opd_conv_w8_supported(const ConvGemmParams& p) {
    if (p.M <= 0 || p.N <= 0 || (p.N % W8_BN) != 0 || p.Cin <= 0 || (p.Cin % 64) != 0 || p.K != p.KH * p.KW * p.Cin || p.K / 64 < 3) return false;
    if (p.KH < 1 || p.KW < 1 || p.KH * p.KW > 32 || p.stride < 1 || p.pad < 0 || p.OH <= 0 || p.OW <= 0) return false;
    if ((long long)p.B * p.OH * p.OW != p.M) return false;
    if (p.stem || p.x2 || p.x_alt || p.split_k > 1 || p.out_f32 || p.res32 || p.bias_period != 0 || p.bias_ptrs || p.out16_aux || !p.bias || !p.out) return false;
    if ((size_t)p.B * p.H * p.W * p.Cin * 2 + (size_t)(p.pad * p.W + p.pad) * p.Cin * 2 >= 0x7fffff00ull || (size_t)p.N * p.K * 2 >= 0x7fffff00ull) return false;
    return true;
}
#endif

hipError_t OPD_SYM(opd_launch_conv_w8)(const ConvGemmParams& p_in, hipStream_t stream) {
    if (!opd_conv_w8_supported(p_in)) return hipErrorInvalidValue;
    ConvGemmParams p = p_in;
    p.fd_ohw = opd_make_fastdiv((unsigned)p.OH * (unsigned)p.OW);
    p.fd_ow = opd_make_fastdiv((unsigned)p.OW);
    p.tap_rep = 0u;
    for (int kh = 0; kh < p.KH; ++kh) p.tap_rep |= 1u << (kh * p.KW);
    const int tiles_m = (p.M + W8_BM - 1) / W8_BM, tiles_n = p.N / W8_BN;
    p.fd_tilesn = opd_make_fastdiv((unsigned)tiles_n);
    const bool pw = p.KH == 1 && p.KW == 1 && p.pad == 0 && p.stride == 1 && p.H == p.OH && p.W == p.OW;
    if (pw) {
        OPD_SET_MAX_LDS_ONCE(conv_w8_kernel<true>, W8_LDS);
        OPD_LAUNCH(conv_w8_kernel<true>, dim3(tiles_m * tiles_n), dim3(512), W8_LDS, stream, p);
    } else {
        OPD_SET_MAX_LDS_ONCE(conv_w8_kernel<false>, W8_LDS);
        OPD_LAUNCH(conv_w8_kernel<false>, dim3(tiles_m * tiles_n), dim3(512), W8_LDS, stream, p);
    }
    return hipGetLastError();
}
