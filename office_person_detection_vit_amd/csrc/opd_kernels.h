// opd_kernels.h — host-callable launchers of the gfx950 kernels (device code lives in kernels_*.hip).
// All launchers enqueue on `stream` and return the hipError_t of the launch; none of them synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

typedef uint16_t f16_t;  // raw 16-bit operand bits on the host side: IEEE half (default) or bfloat16 (OPD_FLAG_BF16)
enum { OPD_DT_F16 = 0, OPD_DT_BF16 = 1 };   // operand type of a launch: every kernel file with 16-bit operands is compiled for both (opd_elem.h)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel instantiation and device, from whichever thread launches it first
// (HipDetrDetector(streams = N) drives several handles from worker threads); used by the launchers in kernels_*.hip.
#include <mutex>
#define OPD_SET_MAX_LDS_ONCE(kernel, bytes)                                                                                  \
    do {                                                                                                                     \
        static std::once_flag once_[16];                                                                                     \
        static hipError_t err_[16];                                                                                          \
        int dev_ = 0;                                                                                                        \
        if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 16) return hipErrorInvalidDevice;   /* one node: <= 16 GPUs */ \
        std::call_once(once_[dev_], [&] { err_[dev_] = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)); }); \
        if (err_[dev_] != hipSuccess) return err_[dev_];                                                                     \
    } while (0)

// Every launch notes the kernel's name for the per-kernel timing table (opd_detr_kernel_table: profiling mode 1).  Templates whose arguments are
// template parameters of the launcher format the instantiation's name the way rocprofv3 prints it (opd_kernel_name).
extern thread_local const char* opd_last_kernel_name;          // opd_host.cpp
extern thread_local int opd_dbg_skip_launch;                   // timing ablation (OPD_DBG_SKIP, tools/abl_forward.sh): launches are noted, not made
const char* opd_kernel_name(const char* fmt, ...);             // a process-lifetime string (call once per instantiation: function-local static)
#define OPD_LAUNCH(kernel, ...)                       \
    do {                                              \
        opd_last_kernel_name = #kernel;               \
        if (!opd_dbg_skip_launch) hipLaunchKernelGGL(kernel, __VA_ARGS__); \
    } while (0)
#define OPD_BOOLSTR(b) ((b) ? "true" : "false")

// Publishing LDS-DMA data through a workgroup barrier.  hipcc models vector memory as ONE in-order queue: behind [LDS-DMA requests, loads into
// registers] it guards `__syncthreads()` with e.g. `s_waitcnt vmcnt(2)` -- "everything but the two youngest loads".  On gfx950 loads into registers
// (and stores) retire out of order with respect to an older LDS-DMA request (tools/microbench/vmorder.hip), so that wait proves nothing about the
// requests.  Every barrier that publishes LDS-DMA data with younger register loads possibly in flight is therefore written OPD_DMA_BARRIER():
// an explicit drain, then the barrier (tools/scan_dma_waits.py checks the compiled code for the pattern).
#if defined(__HIP_DEVICE_COMPILE__)
#define OPD_DMA_BARRIER()                                        \
    do {                                                         \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         \
        __syncthreads();                                         \
    } while (0)
#else
#define OPD_DMA_BARRIER() __syncthreads()
#endif

// ---- implicit-GEMM convolution / linear layer (kernels_gemm.hip) ---------------------------------------------------
// out[m][n] = act( sum_k A[m][k] * Wt[n][k] + bias + residual ),  m = (b,oh,ow), k = (kh,kw,cin), NHWC fp16 input.
// Division of a 31-bit unsigned value by a launch constant: q = one ? m : umulhi(m, mul) >> shift, exact for m < 2^31 (mul =
// ceil(2^(31+s) / d), s = ceil(log2 d), shift = s - 1).  A runtime 32-bit division costs ~40 VALU instructions; the conv / GEMM
// prologue did ten of them per lane (tools/trace_gemm.py: 5 000 of a workgroup's 20 000 clocks on the K = 256 layers).
struct FastDiv { unsigned mul, shift, one; };
inline FastDiv opd_make_fastdiv(unsigned d) {
    FastDiv f{0u, 0u, 1u};
    if (d <= 1u) return f;
    int s = 0;
    while ((1ull << s) < d) ++s;
    f.mul = (unsigned)(((1ull << (31 + s)) + d - 1) / d);
    f.shift = (unsigned)(s - 1);
    f.one = 0u;
    return f;
}

struct ConvGemmParams {
    const f16_t* x;      // [B][H][W][Cin] fp16 (Cin % 64 == 0), or NHWC4 for the stem
    const f16_t* w;      // [N][K] fp16, K = KH*KW*Cin (stem: [N][8][8][4])
    const float* bias;   // [N], or [bias_period][N] when bias_period > 0 (row-periodic bias, e.g. pos-embedding fold)
    const float* const* bias_ptrs;  // optional (device array, bias_period > 0): row m uses bias_ptrs[m / bias_period] instead of
                                    // `bias` — one periodic bias matrix per frame (ragged batches: the fold depends on the mask)
    const f16_t* res16;  // optional fp16 residual [M][N]
    const float* res32;  // optional fp32 residual [M][N]
    void* out;           // [M][N] fp16 (out_f32 == 0) or fp32
    f16_t* out16_aux;    // optional second fp16 copy of an fp32 output (unused when null)
    const void* zero16;  // >= 16 bytes of zeros on the device (source of padded / out-of-range LDS-DMA rows)
    int B, H, W, Cin, OH, OW, N, KH, KW, stride, pad;
    int M, K;
    int relu, bias_period, out_f32;
    int bias_pmod, bias_pcols;  // bias_pcols > 0: the periodic table varies only in columns n with n mod bias_pmod < bias_pcols (both
                                // multiples of 256); elsewhere all its rows are equal and the kernel reads row 0 only
    int stem;            // 0: NHWC conv / linear; 1: stem on a plain NHWC4 image (v1 kernel); 2: stem on the padded NHWC4 image
    int dbg;             // timing ablation for tools (0 = normal; 1 = skip MFMAs, 2 = skip all but the first tile DMA)
    int split_k;         // > 1: K is cut into split_k slices, slice z writes fp32 partials to out + z*M*N (bias in slice 0)
    // optional second activation source (K-concatenated GEMM): k >= K1 reads x2 [B][H2][W2][Cin2] at (oh*stride2, ow*stride2), i.e. a
    // 1x1 convolution of x2 added into the same accumulators; w is then [N][K1 + Cin2], K = K1 + Cin2, K1 = KH*KW*Cin
    const f16_t* x2;
    int H2, W2, Cin2, stride2, K1;
    FastDiv fd_ohw, fd_ow, fd_period;  // filled by opd_launch_conv_gemm: division by OH*OW, OW, bias_period
    // pointwise launches only: column tiles n with n mod alt_mod >= alt_cols read x_alt instead of x (same [M][K] shape) -- the fused
    // QKV projection reads "x + position embedding" for its q / k columns and x for its v columns
    const f16_t* x_alt;
    int alt_mod, alt_cols;
    unsigned tap_rep;                  // filled by opd_launch_conv_gemm: sum over kh of 1 << kh*KW (tap-validity masks)
    FastDiv fd_tilesn, fd_ntiles;      // filled by the LDS-DMA launcher: column tiles, tiles per split-K slice
    unsigned long long* trace;  // tools only: per-workgroup phase stamps [grid][8] (conv_gemm_dma_kernel<..., TRACE>); null in the model
    int force_mt;        // tools only (tools/sweep_tiles.py): 4 / 5 / 6 = tile height 128 / 160 / 192 rows instead of the quantisation-aware choice
    int wprefetch;       // 1: the workgroups of an XCD first touch a share each of the weight matrix (L2 warm-up at launch start; speed only)
    int flat_staging;    // tools / tests: 1 = stage tiles through flat global addresses (the path tensors beyond 2 GiB take) instead of buffer descriptors
    int dtype;           // OPD_DT_F16 / OPD_DT_BF16: the 16-bit operand type of x / w / res16 / out (opd_elem.h)
};
hipError_t opd_launch_conv_gemm(const ConvGemmParams& p, hipStream_t stream);
// eight-wave form for wide layers (kernels_w8.hip: 128 x 256 tiles, one workgroup per CU, three-stage ring); bit-identical results
bool opd_conv_w8_supported(const ConvGemmParams& p);
hipError_t opd_launch_conv_w8(const ConvGemmParams& p, hipStream_t stream);
// fused stem: 7x7 s2 conv + FrozenBN + ReLU + 3x3 s2 max-pool on the zero-bordered NHWC4 image -> pooled NHWC fp16
hipError_t opd_launch_stem_pool_u8(const uint8_t* frames, const int32_t* valid_hw, const f16_t* w, const float* bias, f16_t* out, int B, int H,
                                   int W, int OH, int OW, int PH, int PW, hipStream_t stream, int dtype = 0);   // pre-processing inside the stem
hipError_t opd_launch_stem_pool(const f16_t* x4p, const f16_t* w, const float* bias, f16_t* out, int B, int Hp, int Wp, int OH,
                                int OW, int PH, int PW, hipStream_t stream, int dtype = 0);
// fused bottleneck tail (kernels_btail.hip):  a1 = relu(conv3x3(x1, w1) + b1) ; y = relu(a1*w2 + b2 + res) ; z = relu(y*w3 + b3)
// x1 [B][H][W][C1] fp16, y/res [M][4*C1], z [M][C3]  (M = B*OH*OW, 3x3 pad 1, stride 1 or 2).  w2p / w3p are the 1x1
// weights: plain K order for C1 = 64 / 128, opd_permute_k32 applied along K for C1 = 256 (kernels_btail3.hip).  C3 == 0: no z.  (C1, C3) must satisfy opd_btail_supported.
struct BtailParams {
    const f16_t* x1;
    const f16_t* w1;   // [C1][3][3][C1]
    const float* b1;
    const f16_t* w2p;  // [4*C1][C1]
    const float* b2;
    const f16_t* res;  // [M][4*C1] or null
    const f16_t* xs;   // optional fused shortcut (C1 == 64, C3 == 64, stride 1, res == null): its input [M][64] fp16 ...
    const f16_t* wsc;  // ... and its folded 1x1 weights [4*C1][64] (plain K order); b2 then holds b2 + the shortcut's bias
    f16_t* y;          // [M][4*C1]
    const f16_t* w3p;  // [C3][4*C1] (C3 > 0)
    const float* b3;
    f16_t* z;          // [M][C3]
    int B, H, W, OH, OW, stride, M, C1, C3;
    int dbg;           // timing ablations for tools (0 = normal): 1 skip the 3x3 loop, 2 skip stores, 4 skip residual, 8 stop after the 3x3
    FastDiv fd_ohw, fd_ow;   // filled by opd_launch_btail
    unsigned long long* trace;   // tools only: per-workgroup phase stamps [grid][16] (btail_kernel<..., TRACE>); null in the model
    int rev;           // 1: each XCD walks its tiles in descending order (results identical; see kernels_btail.hip)
    int dtype;         // OPD_DT_F16 / OPD_DT_BF16
    // Round 5, stage 1 (C1 == 64): the block output y = relu(W2 . a1 + residual) is 4x wider than a1, and a 1x1 has no halo, so the NEXT
    // tail can rebuild the y it needs as its residual from the 64-channel tensors that made it instead of reading 274 MB back:
    //   y == null      -> y is not stored at all (the consumer recomputes it);
    //   a1_out         -> [M][C1] fp16: this block's a1 = relu(3x3), stored for that consumer (natural channel order);
    //   rc = 1         -> residual = relu(rc_b[0] + rc_w2[0] . rc_a1[0] + rc_wsc . rc_xs), the previous block's output rebuilt per 64-channel
    //                     chunk with that block's own instruction order (bit-identical to what its tail would have stored); res must be null;
    //   rc = 2         -> two levels: index [1] = the block before the previous one (the one with the shortcut: rc_b[1] = its b2 + bsc), index [0] = the
    //                     previous block; residual = relu(rc_b[0] + rc_w2[0] . rc_a1[0] + relu(rc_b[1] + rc_w2[1] . rc_a1[1] + rc_wsc . rc_xs)); C3 = 128;
    //   y_stride2 = 1  -> y is stored only at pixels with even (oh, ow): the only ones a stride-2 1x1 shortcut of the next stage reads
    //                     (valid when the next block's reduce is fused as z, i.e. nobody else reads y).
    f16_t* a1_out;
    const f16_t* rc_a1[2];
    const f16_t* rc_xs;
    const f16_t* rc_w2[2];
    const f16_t* rc_wsc;
    const float* rc_b[2];
    int rc;
    int y_stride2;
    int nw;   // waves per workgroup of the C1 = 64 / 128 kernels: 0 / 4 = 128-pixel tiles, two workgroups per CU; 8 = 256-pixel tiles, one per CU (identical bits)
};
bool opd_btail_supported(int C1, int C3);
hipError_t opd_launch_btail(const BtailParams& p, hipStream_t stream);
void opd_permute_k32(const f16_t* w, f16_t* out, int rows, int K);  // host
// 256-channel blocks (stage 3): eight-wave form, kernels_btail3.hip; reached through opd_launch_btail
bool opd_btail256_supported(int C1, int C3);
hipError_t opd_launch_btail256(const BtailParams& p, hipStream_t stream);

// ---- Linear(K -> 256) + bias + residual + LayerNorm in one kernel (kernels_rowln.hip) ---------------------------------
// y = LayerNorm(x . W^T + bias + res32) * gamma + beta over rows of 256; writes fp32 y and/or an fp16 copy.  y32 may alias
// res32 (a workgroup reads its own rows before it writes them).
struct GemmLnParams {
    const f16_t* x;      // [M][K] fp16, K % 64 == 0
    const f16_t* w;      // [256][K] fp16
    const float* bias;   // [256]
    const float* res32;  // [M][256] or null
    const float* gamma;  // [256]
    const float* beta;   // [256]
    float* y32;          // [M][256] or null
    f16_t* y16;          // [M][256] or null
    int M, K;
    int kloop;           // tests only: 1 = the k-loop kernel also for K == 256 (default there: the one-shot kernel)
    int deep_k;          // 1: the row-owner ring kernel for deep reductions (any K % 64 == 0; the encoder's FFN-2, K = 2048)
    // deep_k only: optional second fp16 output yp16 = fp16(y + pos[row % pos_period]) (pos: one [pos_period][256] table, or pos_ptrs:
    // one table per frame of pos_period rows) -- the position-embedding shadow the next encoder layer's q / k projection reads
    const float* pos;
    const float* const* pos_ptrs;
    int pos_period;
    f16_t* yp16;
    int dtype;           // OPD_DT_F16 / OPD_DT_BF16: x, w, y16, yp16
};
hipError_t opd_launch_gemm_ln(const GemmLnParams& p, hipStream_t stream);
// The encoder's FFN block in one launch (kernels_rowln.hip::enc_ffn_kernel): y = LayerNorm(res32 + relu(x . W1^T + b1) . W2^T + b2) over rows
// of 256; wpack = opd_encffn_pack(W1, b1, W2): the weights in MFMA-fragment order as eight per-wave streams.  Outputs as GemmLnParams' deep_k
// form (y32 may alias res32, y16 may alias x: a workgroup reads its own rows before it writes them).
struct EncFfnParams {
    const f16_t* x;               // [M][256]
    const unsigned char* wpack;   // opd_encffn_pack_bytes(F)
    const float* b2;              // [256]
    const float* res32;           // [M][256] or null
    const float* gamma;           // [256]
    const float* beta;
    float* y32;
    f16_t* y16;
    f16_t* yp16;                  // optional fp16(y + pos[row % pos_period])
    const float* pos;
    const float* const* pos_ptrs;
    int pos_period;
    int M, F;                     // F % 128 == 0
    int dtype;
    int dbg;                      // tools only: timing ablations (0 in the model)
    int wprefetch;                // 1: the workgroups of an XCD warm its L2 with the weight stream at launch start (speed only)
    // optional tail projection of the block's output, `tail` passes of 256 output columns (the next layer's q / k / v, or the decoder's memory
    // keys / values): tail_out[m][tail_col[t] .. + 255] = fp16((t < tail_pos ? y + pos : y)[m] . Wt_t^T + bias_t); Wt and the biases travel in wpack
    // optional FRONT phase: x = LayerNorm1(res32 + attn . Wo^T + bo) computed inside from the attention output (then `x` above is unused, y32 must be
    // res32: the front phase writes x there and the epilogue reads it back as the FFN's residual); Wo leads wpack (pack_front)
    const f16_t* attn;            // [M][256] or null
    const float* bo;              // [256]
    const float* gamma1;          // [256]
    const float* beta1;
    int pack_front;
    int pack_tail;                // the number of tail passes wpack was built with (stream stride); tail is 0 or pack_tail
    int tail, tail_pos;           // 0 .. 16 passes; the first tail_pos of them multiply y + pos (pos / pos_ptrs / pos_period as for yp16)
    f16_t* tail_out;
    int tail_ld;
    int tail_col[16];
};
hipError_t opd_launch_enc_ffn(const EncFfnParams& p, hipStream_t stream);
size_t opd_encffn_pack_bytes(int F, int tail, int front);
void opd_encffn_pack(const uint16_t* w1, const float* b1, const uint16_t* w2, int F, const uint16_t* wt, const float* bt, int tail, const uint16_t* wo,
                     unsigned char* out);   // host
// Small-M linear layer, reduction cut into 256-wide slices: slice z computes x[:, 256z : 256z+256] . w[:, 256z : 256z+256]^T.
// slices == 1: out = act(. + bias) as fp16 (out16) or fp32 (out32).  slices > 1: fp32 slabs out32[z][M][N], bias in slab 0
// (summed by opd_launch_reduce_ln).  bias_period > 0: row-periodic bias [period][N].
struct GemmK256Params {
    const f16_t* x;     // [M][ldx]
    const f16_t* w;     // [N][ldw], N % 64 == 0
    const float* bias;  // [N] or [bias_period][N]
    f16_t* out16;       // [M][N] or null
    float* out32;       // [slices][M][N] or null
    int M, N, ldx, ldw, slices, bias_period, relu;
    int dtype;          // OPD_DT_F16 / OPD_DT_BF16
};
hipError_t opd_launch_gemm_k256(const GemmK256Params& p, hipStream_t stream);

// ---- element-wise / small kernels (kernels_misc.hip) ----------------------------------------------------------------
// uint8 BGR HWC frames -> normalised fp16 NHWC4 (channel 3 = 0): (x/255 - mean)/std, RGB order, written into a
// zero-bordered image [B][Hp][Wp][4] with the frame at offset (3, 3) (Hp >= H + 6, Wp >= W + 6): the stem's padding.
// valid_hw (device, nullable): [B][2] = (h, w) of each frame inside the H x W canvas (ragged batch); the rest is zero.
hipError_t opd_launch_preprocess_u8(const uint8_t* frames, f16_t* out, int B, int H, int W, int Hp, int Wp, const int32_t* valid_hw,
                                    hipStream_t stream, int dtype = 0);
// float32 NCHW pixel_values -> the same padded fp16 NHWC4 image.
hipError_t opd_launch_preprocess_f32(const float* pv, f16_t* out, int B, int H, int W, int Hp, int Wp, const int32_t* valid_hw,
                                     hipStream_t stream, int dtype = 0);
// Pillow-exact bilinear resize of uint8 HxWx3 frames (two 8-bit passes, 22-bit fixed-point taps); tables from opd_resize_coeffs.
void opd_resize_coeffs(int in_size, int out_size, std::vector<int32_t>* bounds, std::vector<int32_t>* coeffs, int* ksize_out);  // host
hipError_t opd_launch_resize_u8(const uint8_t* in, uint8_t* out, int B, int h, int w, int oh, int ow, const int32_t* bounds_h,
                                const int32_t* coeff_h, int ksize_h, const int32_t* bounds_v, const int32_t* coeff_v, int ksize_v,
                                hipStream_t stream);
// 3x3 stride-2 pad-1 max-pool, NHWC fp16, C % 8 == 0.
hipError_t opd_launch_maxpool(const f16_t* x, f16_t* out, int B, int H, int W, int C, int OH, int OW, hipStream_t stream, int dtype = 0);
// y = LayerNorm(x) * gamma + beta over the last dim (D == 256); writes fp32 y and optional fp16 copy.
hipError_t opd_launch_layernorm(const float* x, const float* gamma, const float* beta, float* y, f16_t* y16,
                                int rows, hipStream_t stream, int dtype = 0);
// y = LayerNorm( sum_z partial[z] + residual ) (gamma == nullptr: no normalisation, plain sum): the deterministic
// reduction of split-K GEMM slabs fused with the residual add and the post-LN of the transformer layers.  D == 256.
hipError_t opd_launch_reduce_ln(const float* partials, int nsplit, size_t slab_stride, const float* residual,
                                const float* gamma, const float* beta, float* y, f16_t* y16, int rows, hipStream_t stream, int dtype = 0);
hipError_t opd_launch_reduce_ln_pos(const float* partials, int nsplit, size_t slab_stride, const float* residual, const float* gamma,
                                    const float* beta, float* y, f16_t* y16, int rows, const float* pos, const float* const* pos_ptrs,
                                    int period, f16_t* yp16, hipStream_t stream, int dtype = 0);   // + yp16 = fp16(y + position embedding)
// y[row][0..255] = c[0..255] for every row: fp32 y and its fp16 copy.
hipError_t opd_launch_broadcast_rows(const float* c, float* y, f16_t* y16, int rows, hipStream_t stream, int dtype = 0);
// fp32 -> fp16 cast of n elements (n % 8 == 0 not required).
hipError_t opd_launch_cast_f16(const float* x, f16_t* y, size_t n, hipStream_t stream, int dtype = 0);
// split-K convolution: out[i] = act(sum_z partials[z * slab_stride + i]) rounded once to the operand type (n % 8 == 0)
hipError_t opd_launch_reduce_act16(const float* partials, int nsplit, size_t slab_stride, f16_t* out, size_t n, int relu, hipStream_t stream, int dtype = 0);
// diagnostic tap: position-weighted 64-bit sums of `bytes / 4` words, OPD_TAP_BLOCKS partials written to slots[0 .. OPD_TAP_BLOCKS)
#define OPD_TAP_BLOCKS 64
hipError_t opd_launch_checksum(const void* buf, size_t bytes, unsigned long long* slots, hipStream_t stream);
// naive fp32 GEMM used once at plan-build time: C[m][n] = sum_k A[m][k]*Wt[n][k] + bias[n]  (Wt fp32 [N][K])
hipError_t opd_launch_gemm_f32(const float* A, const float* Wt, const float* bias, float* C, int M, int N, int K,
                               int ldc, hipStream_t stream);
// heads: logits = hs*Wc^T+bc ; boxes = sigmoid(W3 relu(W2 relu(W1 hs))) ; one block per (b, query); fp32 weights,
// passed TRANSPOSED ([in = 256][out]).
struct HeadParams {
    const float* hs;  // [rows][256]
    // optional (fused decoder): hs is the state before the last layer's FFN; the rows become LN3(hs + b2 + sum partials) first
    const float* partials;   // [nsplit][rows][256]
    int nsplit;
    const float *ffn_b2, *ln3_gamma, *ln3_beta;   // that FFN-2's bias, LN3
    const float *ln_gamma, *ln_beta;  // optional: hs is the decoder state BEFORE its final LayerNorm, applied here first
    const float *wc, *bc, *w1, *b1, *w2, *b2, *w3, *b3;   // weights transposed: wc [256][ncls], w1 / w2 [256][256], w3 [256][4]
    // optional: the three 256-wide layers as split fp16 pairs in MFMA-fragment order (opd_split_f16_frag; the class matrix padded with zero
    // rows to 128): all three set -> kernels_dec.hip::heads2_kernel (wc / w1 / w2 are then unused)
    const f16_t *wc_f, *w1_f, *w2_f;
    float* logits;    // [rows][ncls]
    float* boxes;     // [rows][4]
    int rows, ncls;
};
hipError_t opd_launch_heads(const HeadParams& p, hipStream_t stream);
hipError_t opd_launch_heads2(const HeadParams& p, hipStream_t stream);   // kernels_dec.hip (reached through opd_launch_heads)
// post-process: softmax / max over first ncls-1 / cxcywh->xyxy*scale / threshold -> fixed-slot records + counts.
struct PostParams {
    const float* logits;  // [B][Q][ncls]
    const float* boxes;   // [B][Q][4]
    const int32_t* orig_hw;  // [B][2] device, (h, w)
    void* records;        // opd_det [B][Q] (compacted per frame, query order)
    int32_t* counts;      // [B]
    int B, Q, ncls;
    float threshold;
};
hipError_t opd_launch_postprocess(const PostParams& p, hipStream_t stream);
// ROI mean-pool + L2 normalise on the encoder map [h][w][256] of one frame.
hipError_t opd_launch_roi_features(const float* enc, const int32_t* rois /*[n][4] x0,y0,x1,y1 map coords*/, float* out,
                                   int n, int h, int w, hipStream_t stream);
// the same for the device records of a batch (opd_det [B][Q], counts [B], orig_hw [B][2]): out [B][Q][256], row = query_index, rows the kernel
// does not visit (other classes, other queries) keep what they held
hipError_t opd_launch_roi_features_records(const float* enc, const void* records, const int32_t* counts, const int32_t* orig_hw, int label,
                                           float* out, int B, int Q, int h, int w, hipStream_t stream);

// cross-attention map of one frame and one decoder layer: mean over heads and the `nsel` selected queries of the softmax rows, fp32;
// q rows [query][ldq] (this frame's), k rows [key][ldk] (this frame's, this layer's), stat: >= nsel * heads * 8 bytes of scratch,
// key_valid2 (device, nullable): (rows, cols) of the frame's valid key rectangle; out [Lk]
hipError_t opd_launch_attention_map(const f16_t* q, int ldq, const f16_t* k, int ldk, const int32_t* sel, int nsel, int heads, int Lk, float scale,
                                    const int32_t* key_valid2, int key_row, void* stat, float* out, hipStream_t stream, int dtype = 0);

// tracker cost matrix: similarity (or 1 - similarity) of n1 x n2 (features [n][D] nullable, xywh boxes [n][4], per-row feature flags)
hipError_t opd_launch_similarity_matrix(const float* f1, const float* b1, const uint8_t* has1, int n1, const float* f2, const float* b2,
                                        const uint8_t* has2, int n2, int D, double aw, double mw, int as_distance, float* out,
                                        hipStream_t stream);

// ---- fused decoder (kernels_dec.hip) -----------------------------------------------------------------------------------------------
// The decoder (M = batch x queries rows, 2 % of the FLOPs) is a latency chain and the part of the path whose fp16 operand rounding
// moves the boxes most (tools/drift_split.py), so it runs as FIVE launches per layer on SPLIT operands: every GEMM input x and weight
// W travels as two fp16 numbers, x = hi + lo / 2048 with hi = fp16(x), lo = fp16((x - hi) * 2048), and a product is three MFMAs,
// W.x = Whi.xhi + (Whi.xlo + Wlo.xhi) / 2048 (two fp32 accumulators, combined once): ~22 mantissa bits, no fp16 rounding of weights
// or activations left on the decoder's residual path.  Weights are stored in MFMA-FRAGMENT ORDER (opd_split_f16_frag, host): per (16-row
// tile, 32-wide k-step) 1 KiB of hi then 1 KiB of lo, lane L's 16 bytes at offset 16 L -- what a wave streams by LDS-DMA and reads back
// lane-linear (kernels_dec.hip).  All `w*` pointers of the structs below are such [N][K] matrices (2 * N * K halves).
void opd_split_f16_frag(const float* w, int N, int K, f16_t* out);   // host
struct DecQkvParams {   // [previous layer: h = LN3(h_in + b2 + sum partials)] ; q | k | v = h . Wqkv^T + bias[row % Q]
    const float* h_in;       // [M][256] residual stream before the previous layer's FFN (null with partials == null: h_out is the input)
    const float* partials;   // [nsplit][M][256] fp32 partial sums of the previous layer's FFN-2 (null: no reduce / LayerNorm, h = h_out as it is)
    int nsplit;
    const float* b2;         // [256] that FFN-2's bias
    const float *ln_g, *ln_b;
    float* h_out;            // [M][256] the layer's input state (written by the q workgroups when partials != null; read otherwise)
    const f16_t* w;          // [768][256] = [Wq; Wk; Wv], fragment order
    const float* bias;       // [Q][768] row-periodic: query-position fold + biases (fp32)
    f16_t* q16;              // [M][256]
    f16_t* k16;              // [B][8 heads][8 key tiles][512]: k in MFMA-fragment order (kernels_dec.hip::dec_qkv_kernel)
    f16_t* vT;               // [B][8 heads][4][2][512]: v^T in MFMA-fragment order
    int M, Q;
};
hipError_t opd_launch_dec_qkv(const DecQkvParams& p, hipStream_t stream);
struct DecSelfParams {  // self-attention of one (frame, 16-query slab) + o-proj + residual + LayerNorm + the cross-attention query projection
    const f16_t *q16, *k16, *vT;
    float* h;                // [M][256] in: the layer's input state (residual), out: LayerNorm output
    const f16_t* wo;         // [256][256], fragment order
    const float *bo, *ln_g, *ln_b;
    const f16_t* wq;         // cross-attention q_proj [256][256], fragment order
    const float* rbq;        // [Q][256]: query-position fold + bias of that projection
    f16_t* qc16;             // [M][256] out: fp16, or bf16 when qc_bf16 (the cross-attention kernel multiplies it with K / V of the handle's operand type)
    int qc_bf16;
    int B, Q;
    float scale;
    unsigned long long* trace;   // tools only (tools/trace_dec.py): per-workgroup shader-clock stamps [grid][8] of wave 0; null in the model
};
hipError_t opd_launch_dec_self(const DecSelfParams& p, hipStream_t stream);
struct DecCrossOutParams {   // combine the key splits of the cross-attention, o-proj + residual + LayerNorm
    const float* part_o;     // [splits][M][256] unnormalised sum_k p v per split
    const float* part_ml;    // [splits][M][8][2]: (exponent reference in the log2 domain, sum_k p) per split and head
    int splits;
    const float* res;        // residual rows [M][256], or [res_period][256] repeated per frame (layer 0: the constant state)
    int res_period;
    float* h;                // [M][256] out (may alias res when res_period == 0)
    const f16_t* wo;         // [256][256], fragment order
    const float *bo, *ln_g, *ln_b;
    int M;
};
hipError_t opd_launch_dec_cross_out(const DecCrossOutParams& p, hipStream_t stream);
#define OPD_DEC_FFN_CHUNK 128   // hidden channels per workgroup of dec_ffn_kernel; partial sums: ffn / 128 slabs
struct DecFfnParams {   // partial[c] = relu(h . W1[chunk c]^T + b1[chunk c]) . W2[:, chunk c]^T for 64-row slabs (bias b2 added by the consumer)
    const float* h;          // [M][256]
    const f16_t* w1;         // [F][256], fragment order
    const float* b1;         // [F]
    const f16_t* w2;         // [256][F], fragment order
    float* partials;         // [F / 128][M][256]
    int M, F;
};
hipError_t opd_launch_dec_ffn(const DecFfnParams& p, hipStream_t stream);

// ---- attention (kernels_attn.hip) -----------------------------------------------------------------------------------
// O[b][q][h*32 + d] = softmax(Q K^T * scale) V, head_dim 32; Q/K/V are fp16 row-major with independent leading dims:
// Q at q_ptr[(b*Lq + i)*ldq + h*32 + d] etc.  Output fp16 [B*Lq][ldo].
struct AttnParams {
    const f16_t *q, *k, *v;
    f16_t* o;
    int B, heads, Lq, Lk;
    int ldq, ldk, ldv, ldo;
    float scale;
    const int32_t* key_valid;  // optional (device) [B][2] = (rows, cols) of the valid top-left rectangle of each frame's key map
    int key_row;               // key map row length (key k sits at (k / key_row, k % key_row)); used with key_valid
    unsigned long long* trace; // tools only (tools/trace_attn.py): per-workgroup phase sums [grid][12]; null in the model
    // key-split form (the fused decoder's cross-attention): splits > 0 cuts the key tiles into `splits` contiguous ranges, one workgroup
    // each; a workgroup writes its UNNORMALISED partial result part_o[split][b*Lq + q][h*32 + d] = sum_k p v (fp32) and
    // part_ml[split][b*Lq + q][h][0..1] = (exponent reference of p in the log2 domain, sum_k p); `o` is unused
    int splits;
    float *part_o, *part_ml;
    int dtype;                 // OPD_DT_F16 / OPD_DT_BF16: q, k, v, o
};
hipError_t opd_launch_attention(const AttnParams& p, hipStream_t stream);

// ---- per-element-type entry points (kernels_*.hip compiled for fp16 and for bf16, opd_elem.h); the launchers above are dispatchers on
// the launch's dtype (opd_dispatch.cpp) -----------------------------------------------------------------------------------------------
#define OPD_DECL_ELEM(name, ...)          \
    hipError_t name##_f16(__VA_ARGS__);   \
    hipError_t name##_bf16(__VA_ARGS__);
OPD_DECL_ELEM(opd_launch_conv_gemm, const ConvGemmParams& p, hipStream_t stream)
OPD_DECL_ELEM(opd_launch_conv_w8, const ConvGemmParams& p, hipStream_t stream)
OPD_DECL_ELEM(opd_launch_stem_pool_u8, const uint8_t* frames, const int32_t* valid_hw, const f16_t* w, const float* bias, f16_t* out, int B, int H, int W, int OH,
              int OW, int PH, int PW, hipStream_t stream)
OPD_DECL_ELEM(opd_launch_stem_pool, const f16_t* x4p, const f16_t* w, const float* bias, f16_t* out, int B, int Hp, int Wp, int OH, int OW, int PH, int PW,
              hipStream_t stream)
OPD_DECL_ELEM(opd_launch_btail, const BtailParams& p, hipStream_t stream)
OPD_DECL_ELEM(opd_launch_btail256, const BtailParams& p, hipStream_t stream)
OPD_DECL_ELEM(opd_launch_gemm_ln, const GemmLnParams& p, hipStream_t stream)
OPD_DECL_ELEM(opd_launch_enc_ffn, const EncFfnParams& p, hipStream_t stream)
OPD_DECL_ELEM(opd_launch_gemm_k256, const GemmK256Params& p, hipStream_t stream)
OPD_DECL_ELEM(opd_launch_attention, const AttnParams& p, hipStream_t stream)
OPD_DECL_ELEM(opd_launch_attention_map, const f16_t* q, int ldq, const f16_t* k, int ldk, const int32_t* sel, int nsel, int heads, int Lk, float scale,
              const int32_t* key_valid2, int key_row, void* stat, float* out, hipStream_t stream)
OPD_DECL_ELEM(opd_launch_preprocess_u8, const uint8_t* frames, f16_t* out, int B, int H, int W, int Hp, int Wp, const int32_t* valid_hw, hipStream_t stream)
OPD_DECL_ELEM(opd_launch_preprocess_f32, const float* pv, f16_t* out, int B, int H, int W, int Hp, int Wp, const int32_t* valid_hw, hipStream_t stream)
OPD_DECL_ELEM(opd_launch_maxpool, const f16_t* x, f16_t* out, int B, int H, int W, int C, int OH, int OW, hipStream_t stream)
OPD_DECL_ELEM(opd_launch_layernorm, const float* x, const float* gamma, const float* beta, float* y, f16_t* y16, int rows, hipStream_t stream)
OPD_DECL_ELEM(opd_launch_broadcast_rows, const float* c, float* y, f16_t* y16, int rows, hipStream_t stream)
OPD_DECL_ELEM(opd_launch_reduce_ln, const float* partials, int nsplit, size_t slab_stride, const float* residual, const float* gamma, const float* beta, float* y,
              f16_t* y16, int rows, hipStream_t stream)
OPD_DECL_ELEM(opd_launch_reduce_ln_pos, const float* partials, int nsplit, size_t slab_stride, const float* residual, const float* gamma, const float* beta,
              float* y, f16_t* y16, int rows, const float* pos, const float* const* pos_ptrs, int period, f16_t* yp16, hipStream_t stream)
OPD_DECL_ELEM(opd_launch_cast_f16, const float* x, f16_t* y, size_t n, hipStream_t stream)
OPD_DECL_ELEM(opd_launch_reduce_act16, const float* partials, int nsplit, size_t slab_stride, f16_t* out, size_t n, int relu, hipStream_t stream)
