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
#include "opd_elem.h"

typedef elem_t half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128;
constexpr int BK = 64;           // halfs per k-step = 128 bytes per tile row
constexpr int ROW_BYTES = BK * 2;

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROW_BYTES + ((chunk ^ (row & 7)) << 4); }

// XCD-aware block -> tile map (cdna_hip_programming.md T1, bijective form).  Workgroups are dealt round-robin over the 8
// XCDs (blocks b and b+8 share an XCD and its private 4 MiB L2), so hand each XCD a CONTIGUOUS range of logical tile
// ids: the n-tiles of one m-tile (which re-read the same activation rows) then run on one XCD, back to back, and the
// rows are fetched from HBM / Infinity Cache once instead of once per n-tile.  Speed only, never correctness.
__device__ __forceinline__ int xcd_logical_block(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7;
    const int x = bid & 7, k = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

__device__ __forceinline__ int fdiv(const int m, const FastDiv& f) {   // m >= 0
    return f.one ? m : (int)(__umulhi((unsigned)m, f.mul) >> f.shift);
}

__device__ __forceinline__ uint4 ldg16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ uint2 ldg8(const void* p) { return *reinterpret_cast<const uint2*>(p); }

// ---------------------------------------------------------------------------------------------------------------------
// v2: LDS-DMA staging (global_load_lds_dwordx4: HBM/L2 -> LDS without passing through VGPRs or the ds_write path) and
// a register epilogue.  Profiling of v1 (profiles/r01_first_bench_*) showed the k-loop bound by ds_write_b128
// (~79 B/clk/CU: 32 KB of tile stores per k-step cost as much as the step's 32 MFMAs per wave) and the K=64 layers
// bound by the fp32 LDS round trip of the epilogue.  Here:
//   * every wave issues 1-KiB DMA pieces (8 tile rows x 128 B); the LDS image stays the v1 XOR-swizzled layout, the
//     swizzle is applied on the per-lane SOURCE address (lane l fills physical chunk l&7 of row l>>3, so it fetches
//     logical chunk (l&7)^(l>>3)); out-of-image taps / rows >= M fetch from a 16-byte zero page;
//   * accumulators go bias -> (+residual) -> ReLU -> fp16 in registers; v_permlane16_swap pairs the 4-wide n-quads of
//     two m-tiles into 8 consecutive channels per lane, so stores (and residual loads) are 16 bytes per lane.
// ---------------------------------------------------------------------------------------------------------------------
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void dma16(const void* gsrc, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
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

// Accumulators start from the bias (vector or row-periodic), so the bias loads overlap the first tile's DMA instead of
// sitting on the epilogue's critical path.
template <int NT, int MT = 4>
__device__ __forceinline__ void init_acc_bias(const ConvGemmParams& p, const float* bias, float4v (&acc)[NT][MT], const int m0,
                                              const int n0, const int lane) {
    const int g = lane >> 4, li = lane & 15;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int nq = n0 + nt * 16 + g * 4;
        // (bias_pcols: only columns with n mod bias_pmod < bias_pcols vary over the period -- the q / k thirds of a fused QKV
        //  projection; the rows of the table are identical in the other columns, so row 0 serves as a plain bias vector there)
        if (p.bias_period == 0 || (p.bias_pcols > 0 && (n0 % p.bias_pmod) >= p.bias_pcols)) {
            const float4v b = *reinterpret_cast<const float4v*>(bias + nq);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = b;
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = m0 + mt * 16 + li;
                const int fr = fdiv(m, p.fd_period), pr = m - fr * p.bias_period;   // frame, row within the period
                const float* brow = p.bias_ptrs && m < p.M ? p.bias_ptrs[fr] : bias;  // per-frame fold (ragged batch)
                acc[nt][mt] = m < p.M ? *reinterpret_cast<const float4v*>(brow + (size_t)pr * p.N + nq)
                                      : float4v{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
}

// fp16 residual in the paired 16-byte layout (see epilogue_regs), fetched BEFORE the last k-step's MFMAs.  With an odd
// number of m-tiles the last tile is unpaired: its residual is the lane's own 4-channel quad (8 bytes, in .x/.y).
template <int NT, int MT = 4>
__device__ __forceinline__ void prefetch_res16(const ConvGemmParams& p, uint4 (&res)[NT][(MT + 1) / 2], const int m0, const int n0,
                                               const int lane) {
    const int g = lane >> 4, li = lane & 15;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int mp = 0; mp < MT / 2; ++mp) {
            const int my_m = m0 + (mp * 2 + (g & 1)) * 16 + li;
            const int my_n = n0 + nt * 16 + (g >> 1) * 8;
            res[nt][mp] = make_uint4(0u, 0u, 0u, 0u);
            if (p.res16 && my_m < p.M) res[nt][mp] = *reinterpret_cast<const uint4*>(p.res16 + (size_t)my_m * p.N + my_n);
        }
        if constexpr (MT & 1) {
            const int m = m0 + (MT - 1) * 16 + li;
            res[nt][MT / 2] = make_uint4(0u, 0u, 0u, 0u);
            if (p.res16 && m < p.M) {
                const uint2 r = *reinterpret_cast<const uint2*>(p.res16 + (size_t)m * p.N + n0 + nt * 16 + g * 4);
                res[nt][MT / 2].x = r.x;
                res[nt][MT / 2].y = r.y;
            }
        }
    }
}

// Register epilogue: (+fp32 residual) (+fp16 residual) -> ReLU -> store.  v_permlane16_swap pairs the 4-channel
// accumulator quads of two m-tiles so each lane owns 8 consecutive channels of one row: 16-byte loads and stores.
// An unpaired last m-tile (odd MT) is stored as 8-byte quads.
// `wave_stage` (optional, fp16 output): a wave-private NT*16 x 64-row fp16 staging area in LDS.  The wave
// transposes its output tile through it so that every global store instruction writes whole contiguous row segments
// (8 lanes x 16 B = 128 B per row for BN = 128) instead of 32-byte pieces of 32 different rows: the 32-byte pattern is
// request-rate bound at ~2.7 TB/s on the store-heavy 64->256 / 128->512 layers.
template <int NT, int MT = 4>
__device__ __forceinline__ void epilogue_regs(const ConvGemmParams& p, void* out, float4v (&acc)[NT][MT],
                                              const uint4 (&res)[NT][(MT + 1) / 2], const int m0, const int n0, const int lane,
                                              unsigned char* wave_stage = nullptr) {
    const int g = lane >> 4, li = lane & 15;
    constexpr int WROW = NT * 32;  // bytes per staged row (NT*16 channels fp16)
    const bool staged = wave_stage != nullptr && !p.out_f32;
    // values of the m-tile pair (2*mp, 2*mp + 1) of column tile nt in accumulator layout: + residuals, ReLU
    auto pair_values = [&](const int nt, const int mp, float4v (&v)[2]) {
        const int nq = n0 + nt * 16 + g * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m0 + (mp * 2 + h) * 16 + li;
            v[h] = acc[nt][mp * 2 + h];
            if (p.res32 && m < p.M) v[h] += *reinterpret_cast<const float4v*>(p.res32 + (size_t)m * p.N + nq);
        }
        if (p.res16) {
            const uint4 r = res[nt][mp];
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
    };
    // the unpaired last m-tile of an odd MT (its fp16 residual was fetched as 8-byte quads in accumulator layout)
    auto last_values = [&](const int nt, float4v& v) {
        const int m = m0 + (MT - 1) * 16 + li;
        const int nq = n0 + nt * 16 + g * 4;
        v = acc[nt][MT - 1];
        if (p.res32 && m < p.M) v += *reinterpret_cast<const float4v*>(p.res32 + (size_t)m * p.N + nq);
        if (p.res16) {
            float a, b;
            unpack2h(res[nt][MT / 2].x, a, b); v[0] += a; v[1] += b;
            unpack2h(res[nt][MT / 2].y, a, b); v[2] += a; v[3] += b;
        }
        if (p.relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
        }
    };

    if (staged) {
        // Groups of up to four 16-row m-tiles pass through the wave's 64-row staging area one after the other (LDS operations
        // of one wave execute in order, so the area is re-used without a barrier).  Accumulator layout -> LDS [row][channel],
        // 16-byte chunks XOR-swizzled by row to spread the banks; read back row-contiguous: CPRW lanes cover one staged row.
        constexpr int CPRW = WROW / 16;         // 16-byte chunks per staged row (8 for BN = 128, 4 for BN = 64)
        constexpr int RPI = 64 / CPRW;          // rows per store instruction
        const int c = lane % CPRW, r0 = lane / CPRW;
        f16_t* o16 = reinterpret_cast<f16_t*>(out);
        // chunk swizzle: 128-byte rows take r & 7; two 64-byte rows share one 128-byte bank line, so there the row's low bit
        // already separates the halves and bits 1..2 pick the chunk (r & 3 left the 8-byte writes of a half-wave on 8 of the
        // 16 bank granules: SQ_LDS_BANK_CONFLICT 12 cycles per LDS instruction on the 64-channel tiles, tools/pmc_one_layer.sh)
        auto row_swz = [](const int r) { return CPRW == 8 ? (r & 7) : ((r >> 1) & (CPRW - 1)); };
        auto stage_quad = [&](const int nt, const int slot, const float4v& v) {
            const int r = slot * 16 + li;
            const int cb = nt * 32 + g * 8;  // byte offset of this quad inside the row
            // (rows r and r + 8 share swizzle and bank line: their 8-byte halves trade places, so the 16 lanes of a ds_write_b64 lane
            // group touch 32 different banks: the SQ pass of round 3 read 44-87 % conflict cycles on the pointwise variants without)
            const int off = r * WROW + ((((cb >> 4) ^ row_swz(r)) << 4) | ((cb & 8) ^ (r & 8)));
            *reinterpret_cast<uint2*>(wave_stage + off) = make_uint2(pack2h(v[0], v[1]), pack2h(v[2], v[3]));
        };
#pragma unroll
        for (int g0 = 0; g0 < MT; g0 += 4) {
            const int gt = (MT - g0) < 4 ? (MT - g0) : 4;   // m-tiles in this group (compile-time after unrolling)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int t = 0; t + 1 < gt; t += 2) {
                    float4v v[2];
                    pair_values(nt, (g0 + t) / 2, v);
                    stage_quad(nt, t, v[0]);
                    stage_quad(nt, t + 1, v[1]);
                }
                if (gt & 1) {
                    float4v v;
                    last_values(nt, v);
                    stage_quad(nt, gt - 1, v);
                }
            }
#pragma unroll
            for (int i = 0; i < gt * 16 / RPI; ++i) {
                const int r = r0 + i * RPI;
                const uint4 t = *reinterpret_cast<const uint4*>(wave_stage + r * WROW + ((c ^ row_swz(r)) << 4));
                const uint4 v = (r & 8) ? make_uint4(t.z, t.w, t.x, t.y) : t;
                const int m = m0 + g0 * 16 + r;
                if (m < p.M && !(p.dbg & 64)) *reinterpret_cast<uint4*>(o16 + (size_t)m * p.N + n0 + c * 8) = v;   // (dbg 64: timing ablation)
            }
        }
        return;
    }

#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int nq = n0 + nt * 16 + g * 4;  // this lane's 4 channels in accumulator layout
#pragma unroll
        for (int mp = 0; mp < MT / 2; ++mp) {
            float4v v[2];
            pair_values(nt, mp, v);
            // lane's row / channels in the paired (16-byte) layout: even g -> first m-tile of the pair, odd g -> second
            const int my_m = m0 + (mp * 2 + (g & 1)) * 16 + li;
            const int my_n = n0 + nt * 16 + (g >> 1) * 8;
            const size_t my_o = (size_t)my_m * p.N + my_n;
            if (p.out_f32) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int m = m0 + (mp * 2 + h) * 16 + li;
                    if (m < p.M) *reinterpret_cast<float4v*>(reinterpret_cast<float*>(out) + (size_t)m * p.N + nq) = v[h];
                }
            }
            f16_t* o16 = p.out_f32 ? p.out16_aux : reinterpret_cast<f16_t*>(out);
            if (o16) {
                const uint2v s0 = __builtin_amdgcn_permlane16_swap(pack2h(v[0][0], v[0][1]), pack2h(v[1][0], v[1][1]), false, false);
                const uint2v s1 = __builtin_amdgcn_permlane16_swap(pack2h(v[0][2], v[0][3]), pack2h(v[1][2], v[1][3]), false, false);
                if (my_m < p.M) *reinterpret_cast<uint4*>(o16 + my_o) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            }
        }
        if constexpr (MT & 1) {  // unpaired last m-tile: accumulator layout, 8-byte accesses
            const int m = m0 + (MT - 1) * 16 + li;
            const size_t o = (size_t)m * p.N + nq;
            float4v v;
            last_values(nt, v);
            if (m < p.M) {
                if (p.out_f32) *reinterpret_cast<float4v*>(reinterpret_cast<float*>(out) + o) = v;
                f16_t* o16 = p.out_f32 ? p.out16_aux : reinterpret_cast<f16_t*>(out);
                if (o16) *reinterpret_cast<uint2*>(o16 + o) = make_uint2(pack2h(v[0], v[1]), pack2h(v[2], v[3]));
            }
        }
    }
}

// BUF = true stages through BUFFER descriptors (`buffer_load_dwordx4 ... offen lds`): the per-lane part of an address is
// a 32-bit row offset computed ONCE, the filter-tap / channel-chunk displacement is a scalar (SGPR soffset), and padding
// or rows >= M are expressed by an out-of-range offset, which the descriptor's bounds check turns into zeros.  The
// flat-pointer form (BUF = false) spends ~25 VALU/SALU instructions per 1-KiB piece on 64-bit address arithmetic and
// zero-page selects — at 8 pieces per k-step that, not the MFMAs or the memory system, paced the k-loop.
// DUAL: a K-concatenated GEMM over TWO activation tensors — k-steps below K1 / 64 gather from `x` as usual, the others from `x2`, the
// input of a 1x1 convolution with its own stride at the same output positions.  Used for the first block of stages 3 and 4: the
// block's shortcut convolution becomes extra K of its 1x1 expand, y = relu([a1 | xs] . [W2 | Wsc]^T + (b2 + bsc)), instead of a
// launch of its own that writes a tensor the expand reads back as its residual.
// TRACE (tools only, tools/trace_gemm.py): wave 0 takes a shader-clock stamp at the phase boundaries of the workgroup's life and writes
// them to p.trace[blockIdx.x][8] when it ends: {100-MHz wall clock at entry, entry, prologue done, first tile landed, k-loop done,
// epilogue issued, stores retired, 100-MHz wall clock at the end}.
// PW ("pointwise"): linear layers and 1x1 stride-1 convolutions -- a tile row's source is row m of a [M][Cin] matrix, so the prologue
// needs no (frame, y, x) decomposition, no tap masks and no tap bookkeeping (tools/trace_gemm.py: the general prologue is ~950
// instructions, 4 400 clocks of a K = 256 workgroup's 19 000).
template <int BN, bool BUF, int MT, bool DUAL = false, bool TRACE = false, bool PW = false>
__global__ __launch_bounds__(256, 2) void conv_gemm_dma_kernel(ConvGemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // the buffer-resource type and builtins exist only in the gfx950 pass
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long tstamp[8];
    auto stamp = [&](const int i) {
        if constexpr (TRACE) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tstamp[i])::"memory");
    };
    if constexpr (TRACE) {
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tstamp[0])::"memory");
    }
    stamp(1);
    constexpr int BMT = 32 * MT;       // tile rows: 2 waves along m, MT 16-row MFMA tiles each (128 / 160 / 192)
    constexpr int A_BYTES = BMT * ROW_BYTES;
    constexpr int STAGE_BYTES = A_BYTES + BN * ROW_BYTES;
    constexpr int NT = BN / 32;
    constexpr int B_PIECES = BN / 32;  // 1-KiB pieces of the B tile per wave

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_n = p.N / BN;
    const int ntiles = tiles_n * ((p.M + BMT - 1) / BMT);
    const int lbid_all = xcd_logical_block(blockIdx.x, gridDim.x);
    const int zsplit = p.split_k > 1 ? fdiv(lbid_all, p.fd_ntiles) : 0;  // split-K slice
    const int lbid = lbid_all - zsplit * ntiles;
    const int tile_m = fdiv(lbid, p.fd_tilesn);
    const int tile_n = lbid - tile_m * tiles_n;
    const int m_base = tile_m * BMT;
    const int n_base = tile_n * BN;
    // (kernel arguments are never written: a modified ConvGemmParams would be demoted to scratch memory)
    void* out_ptr = p.out;
    const float* bias_ptr = p.bias;
    if (p.split_k > 1) {  // this slice writes its own fp32 slab; only slice 0 carries the bias
        out_ptr = reinterpret_cast<float*>(p.out) + (size_t)zsplit * p.M * p.N;
        if (zsplit > 0) bias_ptr = reinterpret_cast<const float*>(p.zero16);
    }

    // ---- DMA coordinates: piece q = wave*MT + i covers tile rows 8q..8q+7; lane -> (row 8q + (lane>>3), slot lane&7)
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ lrow;  // logical 16-byte chunk this lane fetches (XOR swizzle on the source side)
    long long a_base[MT];
    int a_ih0[MT], a_iw0[MT];
    bool a_ok[MT];
    {
        const int ohw = p.OH * p.OW;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m_base + (wave * MT + i) * 8 + lrow;
            a_ok[i] = m < p.M;
            const int mm = a_ok[i] ? m : 0;
            const int b = fdiv(mm, p.fd_ohw);
            const int r = mm - b * ohw;
            const int oh = fdiv(r, p.fd_ow);
            const int ow = r - oh * p.OW;
            a_base[i] = (long long)b * p.H * p.W;
            a_ih0[i] = oh * p.stride - p.pad;
            a_iw0[i] = ow * p.stride - p.pad;
        }
    }
    const f16_t* wrow[B_PIECES];
#pragma unroll
    for (int i = 0; i < B_PIECES; ++i)
        wrow[i] = p.w + (size_t)(n_base + (wave * B_PIECES + i) * 8 + lrow) * p.K + lchunk * 8;

    // ---- BUF: descriptors + per-lane 32-bit offsets + per-row tap-validity masks ---------------------------------------
    // activation descriptor is based `backoff` bytes BEFORE the tensor so that every in-image tap has a non-negative
    // offset: byte(ih, iw) = rowoff (centre-less pixel (oh*s, ow*s)) + soff(tap) with soff >= 0.
    const unsigned backoff = (unsigned)(p.pad * p.W + p.pad) * (unsigned)p.Cin * 2u;
    __amdgpu_buffer_rsrc_t rsrc_a, rsrc_b;
    unsigned rowoff[MT], rowmask[MT], woff[B_PIECES];
    const bool stem2 = p.stem == 2;
    if constexpr (BUF) {
        const unsigned a_bytes = stem2 ? (unsigned)((size_t)p.B * p.H * p.W * 8) : (unsigned)((size_t)p.B * p.H * p.W * p.Cin * 2) + backoff;
        const f16_t* xsrc = p.x;
        if constexpr (PW)
            if (p.x_alt && (n_base % p.alt_mod) >= p.alt_cols) xsrc = p.x_alt;   // (workgroup-uniform: a column tile lies in one group)
        rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(xsrc)) - backoff, 0, a_bytes, 0x00020000);
        rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.w), 0, (unsigned)((size_t)p.N * p.K * 2), 0x00020000);
        const int ohw = p.OH * p.OW;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m_base + (wave * MT + i) * 8 + lrow;
            const bool okm = m < p.M;
            if constexpr (PW) {   // row m of [M][Cin]; rows >= M read zeros through the descriptor's bounds check
                rowoff[i] = okm ? (unsigned)m * (unsigned)(p.Cin * 2) + (unsigned)lchunk * 16u : 0x80000000u;
                rowmask[i] = 1u;
                continue;
            }
            const int mm = okm ? m : 0;
            const int b = fdiv(mm, p.fd_ohw);
            const int r = mm - b * ohw;
            const int oh = fdiv(r, p.fd_ow);
            const int ow = r - oh * p.OW;
            // stem == 2: padded NHWC4 image (8 bytes / pixel); a k-step is 2 filter rows x 8 pixels, so the lane's 16-byte
            // chunk sits at (row lchunk>>2, pixel pair lchunk&3) of the window
            rowoff[i] = stem2 ? (unsigned)((b * p.H + oh * 2) * p.W + ow * 2) * 8u + (unsigned)((lchunk >> 2) * p.W) * 8u + (unsigned)(lchunk & 3) * 16u
                              : (unsigned)(((b * p.H + oh * p.stride) * p.W + ow * p.stride) * p.Cin) * 2u + (unsigned)lchunk * 16u;
            // separable validity in closed form (no loops, no branches): the valid kw form a contiguous range [lo_w, hi_w], likewise kh;
            // the row bits are replicated to every valid kh by a multiplication with the matching bits of p.tap_rep = sum 1 << kh*KW
            const int iw0 = ow * p.stride - p.pad, ih0 = oh * p.stride - p.pad;
            const int lo_w = max(0, -iw0), hi_w = min(p.KW - 1, p.W - 1 - iw0);
            const int lo_h = max(0, -ih0), hi_h = min(p.KH - 1, p.H - 1 - ih0);
            auto below = [](const int n) { return n > 0 ? 0xffffffffu >> (32 - n) : 0u; };   // bits [0, n), n <= 32
            const unsigned kwmask = hi_w >= lo_w ? below(hi_w + 1) & ~below(lo_w) : 0u;
            const unsigned hsel = hi_h >= lo_h ? below((hi_h + 1) * p.KW) & ~below(lo_h * p.KW) : 0u;
            rowmask[i] = okm ? kwmask * (p.tap_rep & hsel) : 0u;
        }
#pragma unroll
        for (int i = 0; i < B_PIECES; ++i)
            woff[i] = (unsigned)((n_base + (wave * B_PIECES + i) * 8 + lrow) * p.K) * 2u + (unsigned)lchunk * 16u;
    }
    __amdgpu_buffer_rsrc_t rsrc_a2 = rsrc_a;
    unsigned rowoff2[MT];
    int nk1 = 0x7fffffff;
    if constexpr (DUAL) {
        static_assert(BUF, "the dual-source form uses buffer-descriptor staging");
        nk1 = p.K1 / BK;
        rsrc_a2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.x2), 0, (unsigned)((size_t)p.B * p.H2 * p.W2 * p.Cin2 * 2), 0x00020000);
        const int ohw = p.OH * p.OW;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m_base + (wave * MT + i) * 8 + lrow;
            const int mm = m < p.M ? m : 0;
            const int b = fdiv(mm, p.fd_ohw);
            const int r = mm - b * ohw;
            const int oh = fdiv(r, p.fd_ow);
            const int ow = r - oh * p.OW;
            rowoff2[i] = m < p.M ? (unsigned)(((b * p.H2 + oh * p.stride2) * p.W2 + ow * p.stride2) * p.Cin2) * 2u + (unsigned)lchunk * 16u
                                 : 0x80000000u;
        }
    }

    const int kpc = p.Cin / BK;
    const int nk_all = p.K / BK;
    const int nk = p.split_k > 1 ? nk_all / p.split_k : nk_all;  // k-steps of this slice
    const int ks0 = zsplit * nk;
    // (a per-tile rotated k-order was tried to spread L2 channel load: no gain, and it makes a frame's result depend on
    //  its position in the batch through the fp32 summation order — removed; k-steps run in natural order.)
    int ks_cur = ks0;
    int tap_kh = 0, tap_kw = 0, tap_c = ks0;   // PW / single tap: the k-step IS the channel chunk
    if (!PW && ks0 != 0 && (p.KH | p.KW) != 1) { tap_kh = (ks_cur / kpc) / p.KW; tap_kw = (ks_cur / kpc) % p.KW; tap_c = ks_cur % kpc; }

    auto issue = [&](int, int buf) {
        const int ks = ks_cur;
        unsigned char* As = smem + buf * STAGE_BYTES;
        unsigned char* Bs = As + A_BYTES;
        if constexpr (BUF) {
            const int tap = PW ? 0 : tap_kh * p.KW + tap_kw;
            const int soff_a = PW ? ks * (BK * 2)
                                  : stem2 ? tap_c * 16 * p.W   // two padded image rows per k-step
                                          : ((tap_kh * p.W + tap_kw) * p.Cin + tap_c * BK) * 2;  // scalar displacement of this tap / chunk
            if (DUAL && ks >= nk1) {   // second source: channels 64 (ks - nk1) .. of the strided 1x1 input
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a2, (__attribute__((address_space(3))) void*)(As + (wave * MT + i) * 1024),
                                                             16, rowoff2[i], (ks - nk1) * (BK * 2), 0, 0);
            } else if (!(p.dbg & 8))
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const unsigned vo = (PW || ((rowmask[i] >> tap) & 1u)) ? rowoff[i] : 0x80000000u;  // out of range -> zeros
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void*)(As + (wave * MT + i) * 1024),
                                                         16, vo, soff_a, 0, 0);
            }
            if (!(p.dbg & 16))
#pragma unroll
            for (int i = 0; i < B_PIECES; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (__attribute__((address_space(3))) void*)(Bs + (wave * B_PIECES + i) * 1024),
                                                         16, woff[i], ks * (BK * 2), 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int ih = a_ih0[i] + tap_kh;
                const int iw = a_iw0[i] + tap_kw;
                const bool ok = a_ok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                const f16_t* src = ok ? p.x + (a_base[i] + (long long)ih * p.W + iw) * p.Cin + tap_c * BK + lchunk * 8
                                      : reinterpret_cast<const f16_t*>(p.zero16);
                dma16(src, As + (wave * MT + i) * 1024);
            }
#pragma unroll
            for (int i = 0; i < B_PIECES; ++i) dma16(wrow[i] + (size_t)ks * BK, Bs + (wave * B_PIECES + i) * 1024);
        }
        if constexpr (PW) {
            ++ks_cur;   // (the wrap below only serves the timing ablations that re-issue past the last k-step)
            if (ks_cur == ks0 + nk) ks_cur = ks0;
        } else {
            if (++tap_c == kpc) {
                tap_c = 0;
                if (++tap_kw == p.KW) { tap_kw = 0; ++tap_kh; }
            }
            if (++ks_cur == ks0 + nk) {  // wrap to the first k-step of this slice
                ks_cur = ks0;
                tap_kh = (ks0 / kpc) / p.KW; tap_kw = (ks0 / kpc) % p.KW; tap_c = ks0 % kpc;
            }
        }
    };

    const int wm0 = m_base + wm * (MT * 16), wn0 = n_base + wn * (BN / 2);
    stamp(2);
    issue(0, 0);
    // Cooperative L2 warm-up of the weights (round 5).  Inside the forward a layer's weights are in nobody's L2 when its launch starts, and all
    // workgroups of an XCD ask for the same weight tile at the same k-step: a cold miss (~2 us) in front of EVERY k-step of the first round,
    // with one tile of prefetch distance.  Each workgroup therefore touches a 1 / n-th share of the whole [N][K] matrix at once -- one dword
    // per 128-byte line, LDS-DMA requests whose bytes land in the second stage buffer before its first real tile is requested (the barrier
    // below drains them) -- so that the matrix is on its way into the XCD's L2 within the first microseconds.  n = workgroups sharing the XCD
    // (blockIdx mod 8; placement is speed only).
    if constexpr (BUF) {
        if (p.wprefetch && nk >= 2) {
            const unsigned wbytes = (unsigned)((size_t)p.N * p.K * 2);
            const unsigned nx = (gridDim.x + 7u) >> 3, xi = blockIdx.x >> 3;
            const unsigned seg = ((wbytes + nx - 1u) / nx + 127u) & ~127u;
            const unsigned lo = xi * seg, hi = lo + seg < wbytes ? lo + seg : wbytes;
            unsigned char* const dump = smem + STAGE_BYTES + wave * 256;
            for (unsigned off = lo + (unsigned)tid * 128u; off < hi; off += 256u * 128u)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (__attribute__((address_space(3))) void*)dump, 4, off, 0, 0, 0);
        }
    }
    float4v acc[NT][MT];
    init_acc_bias<NT, MT>(p, bias_ptr, acc, wm0, wn0, lane);
    uint4 res[NT][(MT + 1) / 2];
    __syncthreads();  // emits s_waitcnt vmcnt(0) for the DMA in flight, then s_barrier
    stamp(3);

    const int frow = lane & 15;
    const int fchk = lane >> 4;
    auto compute = [&](int buf) {
        const unsigned char* As = smem + buf * STAGE_BYTES;
        const unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 xf[MT], wf[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                xf[mt] = *reinterpret_cast<const half8*>(As + swz(wm * (MT * 16) + mt * 16 + frow, kk * 4 + fchk));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                wf[nt] = *reinterpret_cast<const half8*>(Bs + swz(wn * (BN / 2) + nt * 16 + frow, kk * 4 + fchk));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[nt][mt] = OPD_MFMA_16x16x32(wf[nt], xf[mt], acc[nt][mt]);
        }
    };
    for (int ks = 0; ks + 1 < nk; ++ks) {
        if ((p.dbg & 3) != 2) issue(ks + 1, (ks & 1) ^ 1);  // dbg: timing-only ablations (tools/microbench), never set by the model
        if ((p.dbg & 3) != 1) compute(ks & 1);
        __syncthreads();
    }
    prefetch_res16<NT, MT>(p, res, wm0, wn0, lane);  // in flight during the last tile's MFMAs
    compute((nk - 1) & 1);
    stamp(4);
    // the stage buffer NOT used by the last k-step is free: each wave takes a private quarter of it for the output transpose
    unsigned char* wave_stage = nullptr;
    // (single-k-step launches allocate ONE stage buffer: it is free once every wave has finished its fragment reads)
    if (!(p.dbg & 32)) {
        if (nk >= 2) {
            wave_stage = smem + (((nk - 1) & 1) ^ 1) * STAGE_BYTES + wave * (64 * NT * 32);
        } else {
            __syncthreads();
            wave_stage = smem + wave * (64 * NT * 32);
        }
    }
    epilogue_regs<NT, MT>(p, out_ptr, acc, res, wm0, wn0, lane, wave_stage);
    if constexpr (TRACE) {
        stamp(5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(6);
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tstamp[7])::"memory");
        if (tid == 0 && p.trace) {
#pragma unroll
            for (int i = 0; i < 8; ++i) p.trace[(size_t)blockIdx.x * 8 + i] = tstamp[i];
        }
    }
#endif
}

template <int BN, bool BUF, int MT, bool DUAL = false, bool TRACE = false, bool PW = false>
hipError_t launch_dma_t(const ConvGemmParams& p_in, hipStream_t stream) {
    constexpr int BMT = 32 * MT;
    constexpr int LDS = 2 * (BMT + BN) * ROW_BYTES;
    OPD_SET_MAX_LDS_ONCE((conv_gemm_dma_kernel<BN, BUF, MT, DUAL, TRACE, PW>), LDS);
    ConvGemmParams p = p_in;
    const int tiles_m = (p.M + BMT - 1) / BMT;
    const int tiles_n = p.N / BN;
    const int splits = p.split_k > 1 ? p.split_k : 1;
    p.fd_tilesn = opd_make_fastdiv((unsigned)tiles_n);
    p.fd_ntiles = opd_make_fastdiv((unsigned)(tiles_m * tiles_n));
    // a single k-step per workgroup (K = 64 layers, split-K slices of one step) needs no second stage buffer: half the
    // LDS -> three workgroups per CU instead of two, which is what hides the DMA / residual / store round trips there
    const int lds = ((p.K / BK) / splits == 1) ? LDS / 2 : LDS;
    OPD_LAUNCH((conv_gemm_dma_kernel<BN, BUF, MT, DUAL, TRACE, PW>), dim3(tiles_m * tiles_n * splits), dim3(256), lds, stream, p);
    static const char* const kname = opd_kernel_name("conv_gemm_dma_kernel<%d, %s, %d, %s, %s, %s>", BN, OPD_BOOLSTR(BUF), MT, OPD_BOOLSTR(DUAL), OPD_BOOLSTR(TRACE), OPD_BOOLSTR(PW));
    opd_last_kernel_name = kname;
    return hipGetLastError();
}

// Workgroups per CU that the LDS footprint of a <BN, MT> tile admits (160 KiB per CU), capped by the 2-waves/SIMD bound.
static inline int dma_blocks_per_cu(int bn, int mt) {
    const int lds = 2 * (32 * mt + bn) * ROW_BYTES;
    const int by_lds = (160 * 1024) / lds;
    return by_lds < 2 ? by_lds : 2;
}

// Tile height by wave quantisation: with T tiles on S = 256 CUs x blocks/CU slots the launch takes ceil(T/S) rounds, and a
// round lasts ~ (rows + cols) of the tile (DMA-throughput bound).  E.g. M = 33 600, N = 256: 128-row tiles give 526
// workgroups on 512 slots (2 rounds), 160-row tiles 420 (1 round).
static inline int pick_mt(int M, int N, int bn, int splits, int force_mt) {
    if (force_mt >= 4 && force_mt <= 6) return force_mt;   // (tools/sweep_tiles.py)
    {   // many rounds: quantisation is noise, keep the 128-row tile (tools/sweep_tiles.py: up to ~8 rounds the taller
        // tiles still win, e.g. the strided stage-2/3 shortcuts: 74 -> 65 us, 53 -> 47 us)
        const long long tiles128 = (long long)((M + 127) / 128) * (N / bn) * splits;
        if (tiles128 > 10LL * 256 * dma_blocks_per_cu(bn, 4)) return 4;
    }
    int best = 4;
    double best_cost = 1e30;
    for (int mt = 4; mt <= 6; ++mt) {
        const long long tiles = (long long)((M + 32 * mt - 1) / (32 * mt)) * (N / bn) * splits;
        const int bpc = dma_blocks_per_cu(bn, mt);
        const long long slots = 256LL * bpc;
        const long long rounds = (tiles + slots - 1) / slots;
        const double cost = (double)rounds * bpc * (32 * mt + bn) * (mt == 4 ? 1.0 : 1.03);  // prefer 128 rows on ties
        if (cost < best_cost) { best_cost = cost; best = mt; }
    }
    return best;
}

template <int BN>
hipError_t launch_dma(const ConvGemmParams& p, hipStream_t stream) {
    // buffer-descriptor staging needs 31-bit byte offsets and <= 32 filter taps; otherwise the flat-pointer form
    const size_t a_bytes = (size_t)p.B * p.H * p.W * p.Cin * 2 + (size_t)(p.pad * p.W + p.pad) * p.Cin * 2;
    const bool buf_ok = !p.flat_staging && a_bytes < 0x7fffff00ull && (size_t)p.N * p.K * 2 < 0x7fffff00ull && p.KH * p.KW <= 32;
    if (p.x2) {   // dual-source form (validated by opd_launch_conv_gemm): 128-column tiles, buffer staging
        if constexpr (BN == 128) {
            if (!buf_ok || (size_t)p.B * p.H2 * p.W2 * p.Cin2 * 2 >= 0x7fffff00ull) return hipErrorInvalidValue;
            switch (pick_mt(p.M, p.N, BN, 1, p.force_mt)) {
                case 5: return launch_dma_t<128, true, 5, true>(p, stream);
                case 6: return launch_dma_t<128, true, 6, true>(p, stream);
                default: return launch_dma_t<128, true, 4, true>(p, stream);
            }
        } else {
            return hipErrorInvalidValue;
        }
    }
    if (!buf_ok) return p.x_alt ? hipErrorInvalidValue : launch_dma_t<BN, false, 4>(p, stream);
    const bool pw = p.KH == 1 && p.KW == 1 && p.pad == 0 && p.stride == 1 && p.H == p.OH && p.W == p.OW && !p.stem;
    if (p.x_alt && (!pw || p.alt_mod <= 0 || (p.alt_mod % BN) != 0 || (p.alt_cols % BN) != 0)) return hipErrorInvalidValue;
    const int mt = pick_mt(p.M, p.N, BN, p.split_k > 1 ? p.split_k : 1, p.force_mt);
    if (p.trace) {   // tools/trace_gemm.py
        if (pw) {
            switch (mt) {
                case 5: return launch_dma_t<BN, true, 5, false, true, true>(p, stream);
                case 6: return launch_dma_t<BN, true, 6, false, true, true>(p, stream);
                default: return launch_dma_t<BN, true, 4, false, true, true>(p, stream);
            }
        }
        switch (mt) {
            case 5: return launch_dma_t<BN, true, 5, false, true>(p, stream);
            case 6: return launch_dma_t<BN, true, 6, false, true>(p, stream);
            default: return launch_dma_t<BN, true, 4, false, true>(p, stream);
        }
    }
    if (pw) {
        switch (mt) {
            case 5: return launch_dma_t<BN, true, 5, false, false, true>(p, stream);
            case 6: return launch_dma_t<BN, true, 6, false, false, true>(p, stream);
            default: return launch_dma_t<BN, true, 4, false, false, true>(p, stream);
        }
    }
    switch (mt) {
        case 5: return launch_dma_t<BN, true, 5>(p, stream);
        case 6: return launch_dma_t<BN, true, 6>(p, stream);
        default: return launch_dma_t<BN, true, 4>(p, stream);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused stem: 7x7 s2 convolution (+FrozenBN, ReLU) AND the 3x3 s2 p1 max-pool (HF:models/resnet/modeling_resnet.py:72-93)
// in one kernel — the 64-channel stem activation (273 MB at batch 8 x 800x1333, written and read back by the unfused
// pair) never reaches HBM.  One workgroup = a 5 x 32 patch of convolution outputs = 2 x 15 pooled pixels (+1 halo
// row/column recomputed by the neighbours: 1.33x the stem FLOPs, which are 2.5 % of the model).
//   * the implicit-GEMM part is conv_gemm_dma_kernel<64, BUF, 5> with a 2-D row map instead of the flat m index:
//     tile row r (0..159) <-> convolution pixel (2*py0 - 1 + r / 32, 2*px0 - 1 + r % 32) of the zero-bordered NHWC4 image;
//   * epilogue: bias -> fp16 patch in LDS (out-of-map pixels = -65504 so the max ignores them, which is exactly
//     MaxPool2d's implicit -inf padding) -> barrier -> 240 threads each reduce one (pooled pixel, 8-channel group), then ReLU.
// ---------------------------------------------------------------------------------------------------------------------
struct StemPoolParams {
    const f16_t* x4p;   // [B][Hp][Wp][4] zero-bordered normalised image
    const f16_t* w;     // [64][8][8][4]
    const float* bias;  // [64]
    f16_t* out;         // pooled [B][PH][PW][64]
    int B, Hp, Wp, OH, OW, PH, PW;
    int tiles_y, tiles_x;
    // U8 form (stem_pool2_kernel<true>): the raw frames instead of x4p -- pre-processing happens while the input patch is staged
    const uint8_t* frames;      // [B][H][W][3] uint8 BGR
    const int32_t* valid_hw;    // nullable [B][2]: frame sizes inside the canvas (ragged batch); outside -> zeros
    int H, W;
};

// 16-byte-chunk XOR swizzle that keeps ds_read_b128 fragment reads conflict free for rows of BKT halfs:
//   128-byte rows: chunk ^= row & 7;   64-byte rows: chunk ^= (-(row >> 2)) & 3   (4 tile rows share a 256-byte bank row).
template <int BKT>
__device__ __forceinline__ int swz_t(int row, int chunk) {
    if constexpr (BKT == 64) return row * 128 + ((chunk ^ (row & 7)) << 4);
    else return row * 64 + ((chunk ^ ((0 - (row >> 2)) & 3)) << 4);
}

// ---------------------------------------------------------------------------------------------------------------------
// Input-stationary stem + max-pool.  An im2col formulation (round 1: the LDS-DMA GEMM on a 2-D row map) sends every input pixel
// through the LDS-DMA path up to 16 times and the 64 x 256 weight tile once per 160 output pixels (112 KiB per tile); its
// k-loop is LDS-bandwidth bound, so this form stages what is unique instead:
//   * the folded weights without the all-zero eighth filter row ([7 kh][64 n][32 k] = 28 KiB) ONCE per workgroup, which
//     then walks up to STEM_TPW tiles along x;
//   * per tile the INPUT PATCH (15 rows x 70 NHWC4 pixels = 8.4 KiB, double buffered) instead of the 80 KiB im2col tile:
//     the B fragment of output pixel (r, c), filter row kh, column pair g is the 16 bytes at patch[(2r + kh)][2c + 2g], so
//     the 7 k-steps (K = 7 x 32 = 224 instead of 256) run straight from LDS without a barrier between them.
// Tile: 5 x 32 convolution outputs -> 2 x 15 pooled pixels (+1 halo row / column recomputed by the neighbours: 1.33x the stem
// FLOPs, which are 2.5 % of the model).  Epilogue: bias -> fp16 patch in LDS (out-of-map pixels = -65504 so the max ignores
// them, which is exactly MaxPool2d's implicit -inf padding) -> barrier -> 240 threads each reduce one (pooled pixel, 8 channels).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int STEM_TPW = 6;              // tiles per workgroup
constexpr int STEM_PROW = 70 * 8;        // bytes per patch row (70 NHWC4 pixels)
constexpr int STEM_PATCH = 9 * 1024;     // 15 rows x 560 B = 8400 B, staged as 9 one-KiB pieces
constexpr int STEM_W_BYTES = 7 * 64 * 64;

// U8 = true: the kernel reads the uint8 BGR frames itself.  A patch chunk (2 pixels = 6 source bytes at an arbitrary byte address)
// is fetched one tile ahead as three aligned dwords per chunk, realigned with v_alignbyte and normalised in registers:
// fp16(fma(v, A_c, -B_c)) with A_c = (1/255) * (1/std_c), B_c = mean_c * (1/std_c) in fp32 gives, for every byte value v = 0..255 and
// every channel, the same fp16 as preprocess_u8_kernel's (float(v) * (1/255) - mean_c) / std_c (enumerated: tests/test_host_cpu.py;
// on the device: the bit-identity test against the two-kernel path) -- so the staged patch is bit-identical to what the two-kernel path
// stages from the materialised NHWC4 image, and that image (69 MB written, 69 MB read at batch 8) and its launch disappear.
// (A 768-entry look-up table in LDS was the first form: six 2-byte LDS reads per chunk with random bank conflicts cost the
//  LDS-bound stem +43 us, more than the 34 us the removed launch took.)
constexpr float STEM_NA[3] = {0.017124755308032036f, 0.017507001757621765f, 0.01742919534444809f};   // RGB: (1/255) / std
constexpr float STEM_NB[3] = {2.1179039478302f, 2.0357141494750977f, 1.804444432258606f};            // RGB: mean / std

// Row placement of the pooling patch in LDS.  The pooling threads of one ds_read_b128 lane group read pixels 2 apart (pooled neighbours),
// i.e. rows of ONE parity: in plain row order those sit in the same half of the 256-byte bank line and collide two by two; the accumulator
// writes (ds_write_b64: 16 consecutive rows per lane group over a 128-byte bank line) collide between rows r and r + 8.  Row bits 0 and 2
// trade places (rows 2 apart alternate between the halves): free.  The write collision stays: swapping the 8-byte halves of rows with bit 3
// set removes it (the GEMM's staged epilogue does that) but costs the pooling 36 selects per tile, in a kernel bound by vector-ALU issue.
__device__ __forceinline__ int stem_prow(const int r) { return (r & ~5) | ((r & 1) << 2) | ((r >> 2) & 1); }

template <bool U8>
__global__ __launch_bounds__(256, 2) void stem_pool2_kernel(StemPoolParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Wl = smem;                              // [7][64][64 B], rows swizzled (swz_t<32>)
    unsigned char* const Pin = smem + STEM_W_BYTES;              // two input patches
    unsigned char* const patch = Pin + 2 * STEM_PATCH;           // [160 pixels][64 ch] fp16 convolution outputs for the pooling
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, li = lane & 15;

    // XCD-aware order: an XCD walks a contiguous run of tile rows top to bottom (one frame each at batch 8), so the 7 patch
    // rows that vertically adjacent tiles share come from its L2 instead of from HBM a second time
    const int nseg = (p.tiles_x + STEM_TPW - 1) / STEM_TPW;
    const int lbid = xcd_logical_block(blockIdx.x, gridDim.x);
    const int seg = lbid % nseg;
    const int ty = (lbid / nseg) % p.tiles_y;
    const int b = lbid / (nseg * p.tiles_y);
    const int tx_first = seg * STEM_TPW;
    const int tx_end = tx_first + STEM_TPW < p.tiles_x ? tx_first + STEM_TPW : p.tiles_x;
    const int py0 = ty * 2;
    const int cy0 = 2 * py0 - 1;                 // first convolution-output row of the tile (may be -1)
    const int iy0 = 2 * cy0;                     // first padded-image row of the patch

    const __amdgpu_buffer_rsrc_t rsrc_a =
        // (the frames' byte count rounded up to whole dwords: the bounds check works on dwords, and the last one may hold the final
        //  pixel's bytes next to <= 3 bytes of the allocation's own padding)
        U8 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.frames), 0, (unsigned)(((size_t)p.B * p.H * p.W * 3 + 3) & ~(size_t)3), 0x00020000)
           : __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.x4p), 0, (unsigned)((size_t)p.B * p.Hp * p.Wp * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(p.w), 0, 64 * 256 * 2, 0x00020000);
    const int vh = U8 ? (p.valid_hw ? p.valid_hw[2 * b] : p.H) : 0, vw = U8 ? (p.valid_hw ? p.valid_hw[2 * b + 1] : p.W) : 0;

    // ---- weights: 28 pieces of 16 rows x 64 B (piece q: filter row q>>2, output channels 16*(q&3)..+15), 7 per wave ------
    {
        const int wrow = lane >> 2, wslot = lane & 3;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int q = wave * 7 + i;
            const int kh = q >> 2, n = (q & 3) * 16 + wrow;
            const int chunk = wslot ^ ((0 - (n >> 2)) & 3);       // source-side swizzle of a 64-byte row
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (__attribute__((address_space(3))) void*)(Wl + q * 1024), 16,
                                                     (unsigned)((n * 256 + kh * 32 + chunk * 8) * 2), 0, 0, 0);
        }
    }
    // ---- input patch: 525 chunks of 16 B (2 pixels); chunk c sits at patch row c / 35, pixel pair c % 35 ------------------
    int prow[3], pcol[3];    // this wave stages pieces wave, wave + 4, (wave + 8 for wave 0)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = (wave + 4 * i) * 64 + lane;
        prow[i] = c / 35;
        pcol[i] = c - prow[i] * 35;
    }
    auto issue_patch = [&](int tx, int buf) {
        const int ix0 = 2 * (2 * (tx * 15) - 1);             // first padded-image column of the patch
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int q = wave + 4 * i;
            if (q < 9) {
                const int iy = iy0 + prow[i], ix = ix0 + 2 * pcol[i];
                const bool ok = prow[i] < 15 && (unsigned)iy < (unsigned)p.Hp && ix >= 0 && ix < p.Wp;
                const unsigned off = ok ? (unsigned)(((b * p.Hp + iy) * p.Wp + ix) * 8) : 0x80000000u;   // out of range -> zeros
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void*)(Pin + buf * STEM_PATCH + q * 1024),
                                                         16, off, 0, 0, 0);
            }
        }
    };
    // ---- U8: the same 525 chunks through registers.  Chunk c = patch row c / 35, pixel pair c % 35 <-> image pixels (y, x), (y, x + 1)
    //      with y = iy0 + row - 3, x = ix0 + 2 pair - 3; its 6 source bytes start at byte ((b H + y) W + x) 3 of the frames.
    unsigned pre[3][3];   // three aligned dwords per chunk, fetched TWO tiles ahead (right after the previous contents have been consumed)
    // Branch-free: a chunk that does not exist (piece >= 9, row >= 15) or lies outside the frame is fetched from an out-of-range offset, which
    // the descriptor's bounds check turns into zeros.  (A predicated load is a branch with its own wait; the wait counters of a block that a
    // wave may skip are unknown at the join, and the compiler then drains vmcnt(0) -- including the pooling's global stores -- at the loop top.)
    auto load_patch = [&](int tx) {
        const int ix0 = 2 * (2 * (tx * 15) - 1);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int y = iy0 + prow[i] - 3, x = ix0 + 2 * pcol[i] - 3;
            const bool ok = (tid >> 6) + 4 * i < 9 && prow[i] < 15 && (unsigned)y < (unsigned)vh && x + 1 >= 0 && x < vw;
            // (a pair that starts left of the image, x = -1, is fetched from its second pixel: byte addresses never go negative
            //  -- hipcc merges the three loads into one dwordx3, and a start before the buffer would zero all of it)
            const unsigned a = ok ? (unsigned)((((b * p.H + y) * p.W + (x < 0 ? 0 : x)) * 3) & ~3) : 0x80000000u;
#pragma unroll
            for (int d = 0; d < 3; ++d) pre[i][d] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_a, a + 4u * d, 0, 0);   // past the end: zeros
        }
    };
    unsigned char* const sink = patch + 160 * 128 + lane * 16;   // 1 KiB per workgroup that nobody reads: target of the chunks that do not exist
    auto store_patch = [&](int tx, int buf) {
        const int ix0 = 2 * (2 * (tx * 15) - 1);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int q = (tid >> 6) + 4 * i;
            const int y = iy0 + prow[i] - 3, x = ix0 + 2 * pcol[i] - 3;
            const bool row_ok = (unsigned)y < (unsigned)vh;
            const bool ok0 = row_ok && (unsigned)x < (unsigned)vw, ok1 = row_ok && (unsigned)(x + 1) < (unsigned)vw;
            const bool shifted = x < 0;   // the fetch started at the pair's second pixel
            const unsigned sh = (unsigned)(((b * p.H + y) * p.W + (shifted ? 0 : x)) * 3) & 3u;
            unsigned lo = __builtin_amdgcn_alignbyte(pre[i][1], pre[i][0], sh);   // source bytes 0..3: B0 G0 R0 B1
            unsigned hi = __builtin_amdgcn_alignbyte(pre[i][2], pre[i][1], sh);   // source bytes 4..7: G1 R1 . .
            if (shifted) { hi = lo >> 8; lo = lo << 24; }                         // bytes 0..2 are B1 G1 R1
            auto nrm = [&](const int c, const unsigned byte) { return __builtin_fmaf((float)byte, STEM_NA[c], -STEM_NB[c]); };
            uint4 o;
            o.x = ok0 ? pack2h(nrm(0, (lo >> 16) & 255u), nrm(1, (lo >> 8) & 255u)) : 0u;   // R0 G0
            o.y = ok0 ? pack2h(nrm(2, lo & 255u), 0.f) : 0u;                                // B0 0
            o.z = ok1 ? pack2h(nrm(0, (hi >> 8) & 255u), nrm(1, hi & 255u)) : 0u;           // R1 G1
            o.w = ok1 ? pack2h(nrm(2, lo >> 24), 0.f) : 0u;                                 // B1 0
            unsigned char* const dst = (q < 9 && prow[i] < 15) ? Pin + buf * STEM_PATCH + q * 1024 + lane * 16 : sink;
            *reinterpret_cast<uint4*>(dst) = o;
        }
    };
    if constexpr (U8) {
        load_patch(tx_first);
        store_patch(tx_first, 0);
        load_patch(tx_first + 1 < tx_end ? tx_first + 1 : tx_first);
    } else {
        issue_patch(tx_first, 0);
    }

    float4v bias2[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) bias2[nt] = *reinterpret_cast<const float4v*>(p.bias + wn * 32 + nt * 16 + g * 4);

    // The kernel is bound by LDS fragment reads (49 KiB per wave and tile against 70 MFMAs), and the weights are the same for every tile of
    // the workgroup: their 14 fragments per wave (7 filter rows x 2 column tiles, 56 VGPRs) are read ONCE, after the first barrier, and stay
    // in registers for the up to six tiles — 35 KiB of fragment reads per wave and tile instead of 49 (round 4).
    half8 wf[7][2];
    for (int tx = tx_first; tx < tx_end; ++tx) {
        const int buf = (tx - tx_first) & 1;
        // this tile's patch (and, first time, the weights) landed; the previous tile's pooling is done.  LDS-DMA data (the weights; the patch of
        // the fp16 form) is published behind an explicit drain; the uint8 form's later tiles have nothing of that kind in flight and must not
        // wait for the pooling's stores
        if (!U8 || tx == tx_first) OPD_DMA_BARRIER();
        else __syncthreads();
        if (tx == tx_first) {
#pragma unroll
            for (int kh = 0; kh < 7; ++kh)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    wf[kh][nt] = *reinterpret_cast<const half8*>(Wl + kh * 4096 + swz_t<32>(wn * 32 + nt * 16 + li, g));
        }
        if constexpr (!U8)
            if (tx + 1 < tx_end) issue_patch(tx + 1, buf ^ 1);
        const int px0 = tx * 15, cx0 = 2 * px0 - 1;
        float4v acc[2][5];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 5; ++mt) acc[nt][mt] = bias2[nt];
        const unsigned char* P = Pin + buf * STEM_PATCH + (li + g) * 16;
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            half8 xf[5];
#pragma unroll
            for (int mt = 0; mt < 5; ++mt) {
                const int m10 = wm * 5 + mt;      // m-tile: convolution row m10 >> 1, column half m10 & 1
                xf[mt] = *reinterpret_cast<const half8*>(P + (2 * (m10 >> 1) + kh) * STEM_PROW + (m10 & 1) * 256);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 5; ++mt) acc[nt][mt] = OPD_MFMA_16x16x32(wf[kh][nt], xf[mt], acc[nt][mt]);
        }
        // ---- fp16 patch [160 pixels][64 ch] (pixels outside the image: -65504 so that the max ignores them) -------
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 5; ++mt) {
                const int r = wm * 80 + mt * 16 + li;
                const int cy = cy0 + (r >> 5), cx = cx0 + (r & 31);
                const bool ok = (unsigned)cy < (unsigned)p.OH && (unsigned)cx < (unsigned)p.OW;
                const float4v v = acc[nt][mt];   // (ReLU: after the pooling, max(relu(x)) == relu(max(x)), on 8 values per thread instead of 40)
                const int cb = wn * 64 + nt * 32 + g * 8;  // byte offset of the quad in the 128-byte pixel row
                const int rr = stem_prow(r);
                const int off = rr * 128 + ((((cb >> 4) ^ (rr & 7)) << 4) | (cb & 8));
                // (the out-of-map select on the two packed words, not on the four floats: the kernel is bound by vector-ALU issue)
                const unsigned lowest = pack2h(-65504.f, -65504.f);
                *reinterpret_cast<uint2*>(patch + off) = make_uint2(ok ? pack2h(v[0], v[1]) : lowest, ok ? pack2h(v[2], v[3]) : lowest);
            }
        // U8: the next tile's patch goes into the buffer this tile does not use (last read in the k-loop of the tile before: every wave has
        // passed this tile's first barrier since); its source bytes were requested a tile ago, and the request for the tile after next follows
        // at once.  The barrier at the top of the next iteration publishes the patch.
        if constexpr (U8)
            if (tx + 1 < tx_end) {
                store_patch(tx + 1, buf ^ 1);
                load_patch(tx + 2 < tx_end ? tx + 2 : tx + 1);
            }
        __syncthreads();
        // ---- 3x3 s2 max over the patch: thread -> (pooled pixel 0..29, 8-channel group 0..7) -------------------------------
        if (tid < 240) {
            const int c8 = tid & 7, pp = tid >> 3;
            const int ly = pp / 15, lx = pp - ly * 15;
            const int py = py0 + ly, px = px0 + lx;
            if (py < p.PH && px < p.PW) {
                half8 m;
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = (elem_t)(-65504.f);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int r = (2 * ly + dy) * 32 + 2 * lx + dx;
                        const int rr = stem_prow(r);
                        const half8 v = *reinterpret_cast<const half8*>(patch + rr * 128 + ((c8 ^ (rr & 7)) << 4));
#ifdef OPD_ELEM_BF16
#pragma unroll
                        for (int j = 0; j < 8; ++j) m[j] = v[j] > m[j] ? v[j] : m[j];
#else
                        m = __builtin_elementwise_max(m, v);   // four v_pk_max_f16 (the element-wise select form: 8 compares + 8 selects)
#endif
                    }
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = m[j] > (elem_t)0.f ? m[j] : (elem_t)0.f;   // ReLU (a -0 from a rounded tiny negative becomes +0, as in the unfused pair)
                *reinterpret_cast<half8*>(p.out + (((size_t)b * p.PH + py) * p.PW + px) * 64 + c8 * 8) = m;
            }
        }
    }
#endif
}

}  // namespace

hipError_t OPD_SYM(opd_launch_conv_gemm)(const ConvGemmParams& p_in, hipStream_t stream) {
    if (p_in.OH <= 0 || p_in.OW <= 0 || p_in.bias_period < 0) return hipErrorInvalidValue;
    ConvGemmParams p = p_in;   // (+ the launch constants' reciprocals; M < 2^31 is the FastDiv range)
    p.fd_ohw = opd_make_fastdiv((unsigned)p.OH * (unsigned)p.OW);
    p.fd_ow = opd_make_fastdiv((unsigned)p.OW);
    p.fd_period = opd_make_fastdiv((unsigned)(p.bias_period > 0 ? p.bias_period : 1));
    p.tap_rep = 0u;
    if (p.KH >= 1 && p.KW >= 1 && p.KH * p.KW <= 32)
        for (int kh = 0; kh < p.KH; ++kh) p.tap_rep |= 1u << (kh * p.KW);
    // host-side shape contract of the kernel (checked before every launch: a violated assumption would fault the GPU)
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.N % 64) != 0 || (p.K % BK) != 0) return hipErrorInvalidValue;
    if (p.stem == 2) {  // padded-NHWC4 stem through the LDS-DMA kernel: [B][H = 2*OH+6][W = 2*OW+6][4], zero borders
        if (p.K != 256 || p.Cin != 256 || p.KH != 1 || p.KW != 1 || p.stride != 2 || p.pad != 0 || p.N != 64 || p.split_k > 1 ||
            p.H < 2 * p.OH + 6 || p.W < 2 * p.OW + 6 || (p.W & 1) || !p.zero16 || (size_t)p.B * p.H * p.W * 8 >= 0x7fffff00ull)
            return hipErrorInvalidValue;
        if ((long long)p.B * p.OH * p.OW != p.M) return hipErrorInvalidValue;
        return launch_dma_t<64, true, 4>(p, stream);
    }
    if (p.stem) {
        if (p.K != 256 || p.KH != 7 || p.KW != 7 || p.stride != 2 || p.pad != 3) return hipErrorInvalidValue;
    } else if (p.x2) {   // dual source: [primary conv | strided 1x1 of x2], see conv_gemm_dma_kernel<.., DUAL>
        if ((p.Cin % BK) != 0 || (p.Cin2 % BK) != 0 || p.K1 != p.KH * p.KW * p.Cin || p.K != p.K1 + p.Cin2 || p.stride2 < 1 || p.split_k > 1 ||
            p.bias_ptrs || (p.N % 128) != 0 || !p.zero16 || p.H2 < (p.OH - 1) * p.stride2 + 1 ||
            p.W2 < (p.OW - 1) * p.stride2 + 1)
            return hipErrorInvalidValue;
        if ((long long)p.B * p.OH * p.OW != p.M) return hipErrorInvalidValue;
        return launch_dma<128>(p, stream);
    } else {
        if ((p.Cin % BK) != 0 || p.K != p.KH * p.KW * p.Cin) return hipErrorInvalidValue;
    }
    if ((long long)p.B * p.OH * p.OW != p.M) return hipErrorInvalidValue;
    if (p.x_alt && (p.split_k > 1 || p.stem || !p.zero16)) return hipErrorInvalidValue;
    if (p.split_k > 1) {  // split-K: linear fp32 partial slabs only, reduced by opd_launch_reduce_ln
        if (!p.out_f32 || p.relu || p.res16 || p.res32 || p.out16_aux || p.bias_period != 0 || p.stem ||
            ((p.K / BK) % p.split_k) != 0)
            return hipErrorInvalidValue;
        if (!p.zero16) return hipErrorInvalidValue;
        return (p.N % 128 == 0 && (long long)((p.M + BM - 1) / BM) * (p.N / 128) * p.split_k >= 384) ? launch_dma<128>(p, stream)
                                                                                                     : launch_dma<64>(p, stream);
    }
    const int tiles_m = (p.M + BM - 1) / BM;
    static const int wide_min = [] { const char* v = getenv("OPD_WIDE_MIN"); return v ? atoi(v) : 384; }();   // (A/B switch: 128-column tiles from this many tiles on)
    const bool wide = (p.N % 128 == 0) && ((long long)tiles_m * (p.N / 128) >= wide_min);
    if (p.bias_ptrs) {  // per-frame periodic bias: implemented by the LDS-DMA kernel's accumulator initialisation only
        if (p.bias_period <= 0 || p.stem || !p.zero16) return hipErrorInvalidValue;
        return wide ? launch_dma<128>(p, stream) : launch_dma<64>(p, stream);
    }
    if (p.stem || !p.zero16) return hipErrorInvalidValue;   // (the 7x7 stem runs in stem_pool2_kernel, or as stem == 2 above)
    return wide ? launch_dma<128>(p, stream) : launch_dma<64>(p, stream);
}

// x4p: zero-bordered NHWC4 image [B][Hp = 2*OH+6][Wp = 2*OW+6][4]; out: pooled [B][PH][PW][64]

// Pre-processing + stem + max-pool in one launch: frames [B][H][W][3] uint8 BGR (valid_hw nullable [B][2]), geometry as below with
// Hp = 2 OH + 6, Wp = 2 OW + 6 the size the materialised padded image would have.
hipError_t OPD_SYM(opd_launch_stem_pool_u8)(const uint8_t* frames, const int32_t* valid_hw, const f16_t* w, const float* bias, f16_t* out, int B, int H,
                                   int W, int OH, int OW, int PH, int PW, hipStream_t stream) {
    if (B <= 0 || H <= 0 || W <= 0 || OH != (H - 1) / 2 + 1 || OW != (W - 1) / 2 + 1 || PH != (OH - 1) / 2 + 1 || PW != (OW - 1) / 2 + 1 ||
        (size_t)B * H * W * 3 >= 0x7fffff00ull)
        return hipErrorInvalidValue;
    StemPoolParams p{};
    p.frames = frames; p.valid_hw = valid_hw; p.H = H; p.W = W;
    p.w = w; p.bias = bias; p.out = out; p.B = B; p.Hp = 2 * OH + 6; p.Wp = 2 * OW + 6; p.OH = OH; p.OW = OW; p.PH = PH; p.PW = PW;
    p.tiles_y = (PH + 1) / 2;
    p.tiles_x = (PW + 14) / 15;
    constexpr int LDS2 = STEM_W_BYTES + 2 * STEM_PATCH + 160 * ROW_BYTES + 1024;   // (+ the sink of the U8 form)
    OPD_SET_MAX_LDS_ONCE(stem_pool2_kernel<true>, LDS2);
    const int nseg = (p.tiles_x + STEM_TPW - 1) / STEM_TPW;
    OPD_LAUNCH(stem_pool2_kernel<true>, dim3(B * p.tiles_y * nseg), dim3(256), LDS2, stream, p);
    return hipGetLastError();
}

hipError_t OPD_SYM(opd_launch_stem_pool)(const f16_t* x4p, const f16_t* w, const float* bias, f16_t* out, int B, int Hp, int Wp, int OH,
                                int OW, int PH, int PW, hipStream_t stream) {
    if (Hp < 2 * OH + 6 || Wp < 2 * OW + 6 || (Wp & 1) || PH != (OH - 1) / 2 + 1 || PW != (OW - 1) / 2 + 1 ||
        (size_t)B * Hp * Wp * 8 >= 0x7fffff00ull)
        return hipErrorInvalidValue;
    StemPoolParams p{};
    p.x4p = x4p; p.w = w; p.bias = bias; p.out = out; p.B = B; p.Hp = Hp; p.Wp = Wp; p.OH = OH; p.OW = OW; p.PH = PH; p.PW = PW;
    p.tiles_y = (PH + 1) / 2;
    p.tiles_x = (PW + 14) / 15;
    constexpr int LDS2 = STEM_W_BYTES + 2 * STEM_PATCH + 160 * ROW_BYTES + 1024;   // (+ the sink of the U8 form)
    OPD_SET_MAX_LDS_ONCE(stem_pool2_kernel<false>, LDS2);
    const int nseg = (p.tiles_x + STEM_TPW - 1) / STEM_TPW;
    OPD_LAUNCH(stem_pool2_kernel<false>, dim3(B * p.tiles_y * nseg), dim3(256), LDS2, stream, p);
    return hipGetLastError();
}
