// opd_test_api.cpp — kernel-level test hooks (host buffers in, host buffers out).  NOT part of the drop-in boundary
// (include/opd_detr.h) and NOT part of the product library: this file is linked only into libopd_hip_test.so (csrc/build.py), which
// tests/ and tools/ load instead of libopd_hip.so, so that tests/test_kernels_gpu.py can check each hand-written kernel against the
// oracle on identical inputs.  Every kernel hook allocates its own device buffers, runs ONE kernel on the null stream and frees.
#include <string.h>

#include <string>
#include <cstring>
#include <vector>

#include "opd_model.h"

namespace {
int tfail(int code, const std::string& msg) {
    opd::g_err = msg;
    return code;
}
struct DevMem {
    std::vector<void*> ptrs;
    ~DevMem() {
        for (void* p : ptrs) (void)hipFree(p);
    }
    template <typename T>
    T* up(const T* host, size_t count) {
        void* d = nullptr;
        if (hipMalloc(&d, count * sizeof(T) + 16) != hipSuccess) return nullptr;
        ptrs.push_back(d);
        if (host && hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        return reinterpret_cast<T*>(d);
    }
};
#define TCHK(expr)                                                                                          \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) return tfail(OPD_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));    \
    } while (0)
}  // namespace

#pragma GCC visibility push(default)   // (the library is built with -fvisibility=hidden; these are the test build's extra exports)
extern "C" {

// Launch options of the hooks below (test infrastructure only; the kernel tests are single-threaded): bits 8-10 = forced tile height
// (4 / 5 / 6 x 32 rows), bit 5 = flat-address tile staging (the path tensors beyond 2 GiB take), bit 0 of the second word = k-loop
// gemm + LayerNorm kernel also for K == 256.
static int g_conv_flags = 0, g_gemm_ln_kloop = 0, g_test_dtype = 0;
// the 16-bit operand type the kernel hooks below launch with: their uint16 buffers then hold bfloat16 bit patterns (OPD_DT_BF16)
int opd_test_set_elem_bf16(int on) { g_test_dtype = on ? OPD_DT_BF16 : OPD_DT_F16; return OPD_OK; }
int opd_test_set_conv_flags(int flags) { g_conv_flags = flags; return OPD_OK; }
int opd_test_set_gemm_ln_kloop(int on) { g_gemm_ln_kloop = on ? 1 : 0; return OPD_OK; }
static void apply_conv_flags(ConvGemmParams& p, int flags) {
    p.force_mt = (flags >> 8) & 7;
    p.flat_staging = (flags >> 5) & 1;
}

// x: NHWC fp16 bits [B][H][W][Cin] (stem: NHWC4); w: [N][K] fp16 bits; bias fp32 [N] (or [period][N]);
// res16/res32 optional; out fp16 bits or fp32 [M][N].
int opd_test_conv_gemm(const uint16_t* x, const uint16_t* w, const float* bias, const uint16_t* res16, const float* res32,
                       void* out, int B, int H, int W, int Cin, int OH, int OW, int N, int KH, int KW, int stride, int pad,
                       int relu, int bias_period, int out_f32, int stem) {
    DevMem dm;
    const size_t M = (size_t)B * OH * OW;
    const int K = stem ? 256 : KH * KW * Cin;
    const size_t xin = (size_t)B * H * W * (stem ? 4 : Cin);
    ConvGemmParams p{}; p.dtype = g_test_dtype;
    p.x = dm.up(x, xin);
    p.w = dm.up(w, (size_t)N * K);
    p.bias = dm.up(bias, (size_t)N * (bias_period > 0 ? bias_period : 1));
    p.res16 = res16 ? dm.up(res16, M * N) : nullptr;
    p.res32 = res32 ? dm.up(res32, M * N) : nullptr;
    const size_t obytes = M * N * (out_f32 ? 4 : 2);
    p.out = dm.up<unsigned char>(nullptr, obytes);
    {
        const uint32_t zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        p.zero16 = dm.up(zeros, 8);
    }
    if (!p.zero16 || !p.x || !p.w || !p.bias || !p.out || (res16 && !p.res16) || (res32 && !p.res32)) return tfail(OPD_ENOMEM, "test alloc failed");
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.N = N; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
    p.M = (int)M; p.K = K; p.relu = relu; p.bias_period = bias_period; p.out_f32 = out_f32; p.stem = stem;
    apply_conv_flags(p, g_conv_flags);
    if (g_conv_flags & (1 << 12)) {   // the eight-wave kernel (kernels_w8.hip)
        if (!opd_conv_w8_supported(p)) return tfail(OPD_EINVAL, "conv_w8: shape outside the kernel's contract");
        TCHK(opd_launch_conv_w8(p, nullptr));
    } else {
        TCHK(opd_launch_conv_gemm(p, nullptr));
    }
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(out, p.out, obytes, hipMemcpyDeviceToHost));
    return OPD_OK;
}

// dual-source GEMM: out = relu( conv(x, w1; KH x KH, stride, pad) + conv1x1(x2, w2; stride2) + bias ): w1 [N][KH*KH*Cin], w2 [N][Cin2]
// (the hook concatenates them along K), out fp16 [M][N]
int opd_test_conv_dual(const uint16_t* x, const uint16_t* w1, const uint16_t* x2, const uint16_t* w2, const float* bias, uint16_t* out,
                       int B, int H, int W, int Cin, int KH, int stride, int pad, int N, int H2, int W2, int Cin2, int stride2, int relu) {
    DevMem dm;
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KH) / stride + 1;
    const size_t M = (size_t)B * OH * OW;
    const int K1 = KH * KH * Cin, K = K1 + Cin2;
    std::vector<uint16_t> wc((size_t)N * K);
    for (int n = 0; n < N; ++n) {
        memcpy(&wc[(size_t)n * K], w1 + (size_t)n * K1, (size_t)K1 * 2);
        memcpy(&wc[(size_t)n * K + K1], w2 + (size_t)n * Cin2, (size_t)Cin2 * 2);
    }
    ConvGemmParams p{}; p.dtype = g_test_dtype;
    p.x = dm.up(x, (size_t)B * H * W * Cin);
    p.x2 = dm.up(x2, (size_t)B * H2 * W2 * Cin2);
    p.w = dm.up(wc.data(), wc.size());
    p.bias = dm.up(bias, (size_t)N);
    p.out = dm.up<unsigned char>(nullptr, M * N * 2);
    const uint32_t zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    p.zero16 = dm.up(zeros, 8);
    if (!p.x || !p.x2 || !p.w || !p.bias || !p.out || !p.zero16) return tfail(OPD_ENOMEM, "test alloc failed");
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.N = N; p.KH = KH; p.KW = KH; p.stride = stride; p.pad = pad;
    p.M = (int)M; p.K = K; p.K1 = K1; p.relu = relu; p.H2 = H2; p.W2 = W2; p.Cin2 = Cin2; p.stride2 = stride2;
    TCHK(opd_launch_conv_gemm(p, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(out, p.out, M * N * 2, hipMemcpyDeviceToHost));
    return OPD_OK;
}

// split-K linear + fused reduce / residual / LayerNorm: y = LN(x.W^T + bias + res) (gamma == null: no LN), N == 256
int opd_test_gemm_splitk_ln(const uint16_t* x, const uint16_t* w, const float* bias, const float* res32, const float* gamma,
                            const float* beta, float* y, uint16_t* y16, int M, int K, int splits) {
    DevMem dm;
    const int N = 256;
    ConvGemmParams p{}; p.dtype = g_test_dtype;
    p.x = dm.up(x, (size_t)M * K);
    p.w = dm.up(w, (size_t)N * K);
    p.bias = dm.up(bias, N);
    std::vector<float> zeros(N, 0.f);
    p.zero16 = dm.up(zeros.data(), N);
    float* slab = dm.up<float>(nullptr, (size_t)splits * M * N);
    const float* dres = res32 ? dm.up(res32, (size_t)M * N) : nullptr;
    const float* dg = gamma ? dm.up(gamma, N) : nullptr;
    const float* db = beta ? dm.up(beta, N) : nullptr;
    float* dy = dm.up<float>(nullptr, (size_t)M * N);
    uint16_t* dy16 = dm.up<uint16_t>(nullptr, (size_t)M * N);
    if (!p.x || !p.w || !p.bias || !p.zero16 || !slab || !dy || !dy16) return tfail(OPD_ENOMEM, "test alloc failed");
    p.out = slab;
    p.B = M; p.H = 1; p.W = 1; p.Cin = K; p.OH = 1; p.OW = 1; p.N = N; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0;
    p.M = M; p.K = K; p.out_f32 = 1; p.split_k = splits;
    TCHK(opd_launch_conv_gemm(p, nullptr));
    TCHK(opd_launch_reduce_ln(slab, splits, (size_t)M * N, dres, dg, db, dy, dy16, M, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(y, dy, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    TCHK(hipMemcpy(y16, dy16, (size_t)M * N * 2, hipMemcpyDeviceToHost));
    return OPD_OK;
}

// fused Linear(K->256) + bias + residual + LayerNorm (kernels_rowln.hip)
int opd_test_gemm_ln(const uint16_t* x, const uint16_t* w, const float* bias, const float* res32, const float* gamma,
                     const float* beta, float* y, uint16_t* y16, int M, int K) {
    DevMem dm;
    GemmLnParams p{}; p.dtype = g_test_dtype;
    p.x = dm.up(x, (size_t)M * K);
    p.w = dm.up(w, (size_t)256 * K);
    p.bias = dm.up(bias, 256);
    p.res32 = res32 ? dm.up(res32, (size_t)M * 256) : nullptr;
    p.gamma = dm.up(gamma, 256);
    p.beta = dm.up(beta, 256);
    p.y32 = dm.up<float>(nullptr, (size_t)M * 256);
    p.y16 = dm.up<uint16_t>(nullptr, (size_t)M * 256);
    if (!p.x || !p.w || !p.bias || !p.gamma || !p.beta || !p.y32 || !p.y16 || (res32 && !p.res32)) return tfail(OPD_ENOMEM, "test alloc failed");
    p.M = M; p.K = K; p.kloop = g_gemm_ln_kloop;
    TCHK(opd_launch_gemm_ln(p, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(y, p.y32, (size_t)M * 256 * 4, hipMemcpyDeviceToHost));
    TCHK(hipMemcpy(y16, p.y16, (size_t)M * 256 * 2, hipMemcpyDeviceToHost));
    return OPD_OK;
}

// deep-K row-owner form (gemm_ln256_ring_kernel) with the optional position shadow: pos [period][256] fp32, yp16 = fp16(y + pos[row % period])
int opd_test_gemm_ln_deep(const uint16_t* x, const uint16_t* w, const float* bias, const float* res32, const float* gamma, const float* beta,
                          const float* pos, int period, float* y, uint16_t* y16, uint16_t* yp16, int M, int K, int in_place) {
    DevMem dm;
    GemmLnParams p{}; p.dtype = g_test_dtype;
    p.x = dm.up(x, (size_t)M * K);
    p.w = dm.up(w, (size_t)256 * K);
    p.bias = dm.up(bias, 256);
    float* res = res32 ? dm.up(res32, (size_t)M * 256) : nullptr;
    p.res32 = res;
    p.gamma = gamma ? dm.up(gamma, 256) : nullptr;   // null: no LayerNorm (the input projection)
    p.beta = beta ? dm.up(beta, 256) : nullptr;
    p.y32 = (in_place && res) ? res : dm.up<float>(nullptr, (size_t)M * 256);   // the model writes the residual stream in place
    p.y16 = dm.up<uint16_t>(nullptr, (size_t)M * 256);
    p.pos = pos ? dm.up(pos, (size_t)period * 256) : nullptr;
    p.pos_period = period;
    p.yp16 = pos ? dm.up<uint16_t>(nullptr, (size_t)M * 256) : nullptr;
    if (!p.x || !p.w || !p.bias || (gamma && (!p.gamma || !p.beta)) || !p.y32 || !p.y16 || (res32 && !p.res32) || (pos && (!p.pos || !p.yp16)))
        return tfail(OPD_ENOMEM, "test alloc failed");
    p.M = M; p.K = K; p.deep_k = 1;
    TCHK(opd_launch_gemm_ln(p, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(y, p.y32, (size_t)M * 256 * 4, hipMemcpyDeviceToHost));
    TCHK(hipMemcpy(y16, p.y16, (size_t)M * 256 * 2, hipMemcpyDeviceToHost));
    if (pos) TCHK(hipMemcpy(yp16, p.yp16, (size_t)M * 256 * 2, hipMemcpyDeviceToHost));
    return OPD_OK;
}

// Times `iters` launches of Linear(K -> 256) + residual + LayerNorm (deep != 0: the row-owner ring kernel) on M rows of arbitrary data.
int opd_test_bench_gemm_ln(int M, int K, int deep, int iters, float* us_out) {
    DevMem dm;
    GemmLnParams p{}; p.dtype = g_test_dtype;
    uint16_t* x = dm.up<uint16_t>(nullptr, (size_t)M * K);
    uint16_t* w = dm.up<uint16_t>(nullptr, (size_t)256 * K);
    float* f = dm.up<float>(nullptr, 1024);
    float* res = dm.up<float>(nullptr, (size_t)M * 256);
    uint16_t* y16 = dm.up<uint16_t>(nullptr, (size_t)M * 256);
    if (!x || !w || !f || !res || !y16) return tfail(OPD_ENOMEM, "bench alloc failed");
    TCHK(hipMemset(x, 0x2c, (size_t)M * K * 2));
    TCHK(hipMemset(w, 0x1c, (size_t)256 * K * 2));
    TCHK(hipMemset(f, 0, 4096));
    TCHK(hipMemset(res, 0, (size_t)M * 256 * 4));
    p.x = x; p.w = w; p.bias = f; p.gamma = f + 256; p.beta = f + 512; p.res32 = res; p.y32 = res; p.y16 = y16; p.M = M; p.K = K; p.deep_k = deep;
    hipEvent_t a, b;
    TCHK(hipEventCreate(&a)); TCHK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) TCHK(opd_launch_gemm_ln(p, nullptr));
    TCHK(hipEventRecord(a, nullptr));
    for (int i = 0; i < iters; ++i) TCHK(opd_launch_gemm_ln(p, nullptr));
    TCHK(hipEventRecord(b, nullptr));
    TCHK(hipEventSynchronize(b));
    float ms = 0.f;
    TCHK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *us_out = ms * 1000.f / iters;
    return OPD_OK;
}

// the encoder's FFN block in one launch (enc_ffn_kernel): x [M][256], w1 [F][256], w2 [256][F] as 16-bit elements of the current test element
// type, b1 [F], b2 / gamma / beta [256], res32 [M][256]; y = LayerNorm(res32 + relu(x . w1^T + b1) . w2^T + b2); optional position shadow as in
// opd_test_gemm_ln_deep.  in_place: y32 aliases res32 and y16 aliases x, as in the model.  Optional tail projection: wt [tail * 256][256],
// tail_bias [tail * 256], the first tail_pos passes on y + pos; tail_out [M][tail * 256] (pass t at columns 256 t).  Optional FRONT phase
// (wo != null): x is the ATTENTION output; x' = LayerNorm1(res32 + x . wo^T + bo) * g1 + be1 is computed inside, returned in x1_out [M][256]
// (fp32), and the FFN runs on fp16(x') with the residual x' (always in place on res32 then).
int opd_test_enc_ffn(const uint16_t* x, const uint16_t* w1, const float* b1, const uint16_t* w2, const float* b2, const float* res32, const float* gamma,
                     const float* beta, const float* pos, int period, float* y, uint16_t* y16, uint16_t* yp16, int M, int F, int in_place,
                     const uint16_t* wt, const float* tail_bias, int tail, int tail_pos, uint16_t* tail_out, const uint16_t* wo, const float* bo,
                     const float* g1, const float* be1, int pack_front) {
    if (M <= 0 || F <= 0 || F % 128 || tail < 0 || tail > 16) return tfail(OPD_EINVAL, "enc_ffn: F must be a multiple of 128, tail <= 16");
    if (wo && !pack_front) return tfail(OPD_EINVAL, "enc_ffn: the front phase needs a stream packed with it");
    DevMem dm;
    std::vector<unsigned char> pk(opd_encffn_pack_bytes(F, tail, pack_front));
    std::vector<uint16_t> wo_dummy((size_t)256 * 256, 0);
    opd_encffn_pack(w1, b1, w2, F, wt, tail_bias, tail, pack_front ? (wo ? wo : wo_dummy.data()) : nullptr, pk.data());
    EncFfnParams p{}; p.dtype = g_test_dtype;
    uint16_t* dx = dm.up(x, (size_t)M * 256);
    float* res = dm.up(res32, (size_t)M * 256);
    p.wpack = dm.up(pk.data(), pk.size()); p.b2 = dm.up(b2, 256); p.res32 = res; p.gamma = dm.up(gamma, 256); p.beta = dm.up(beta, 256);
    p.pack_front = pack_front;
    if (wo) {
        p.attn = dx; p.bo = dm.up(bo, 256); p.gamma1 = dm.up(g1, 256); p.beta1 = dm.up(be1, 256);
        if (!p.bo || !p.gamma1 || !p.beta1) return tfail(OPD_ENOMEM, "test alloc failed");
        in_place = 1;
    } else {
        p.x = dx;
    }
    p.y32 = in_place ? res : dm.up<float>(nullptr, (size_t)M * 256);
    p.y16 = (in_place && !wo) ? dx : dm.up<uint16_t>(nullptr, (size_t)M * 256);
    p.pos = pos ? dm.up(pos, (size_t)period * 256) : nullptr;
    p.pos_period = period;
    p.yp16 = pos ? dm.up<uint16_t>(nullptr, (size_t)M * 256) : nullptr;
    if (!dx || !p.wpack || !p.b2 || !p.res32 || !p.gamma || !p.beta || !p.y32 || !p.y16 || (pos && (!p.pos || !p.yp16))) return tfail(OPD_ENOMEM, "test alloc failed");
    p.M = M; p.F = F; p.pack_tail = tail; p.tail = tail; p.tail_pos = tail_pos;
    if (tail) {
        p.tail_ld = tail * 256;
        p.tail_out = dm.up<uint16_t>(nullptr, (size_t)M * p.tail_ld);
        if (!p.tail_out) return tfail(OPD_ENOMEM, "test alloc failed");
        for (int t = 0; t < tail; ++t) p.tail_col[t] = 256 * t;
    }
    TCHK(opd_launch_enc_ffn(p, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(y, p.y32, (size_t)M * 256 * 4, hipMemcpyDeviceToHost));
    TCHK(hipMemcpy(y16, p.y16, (size_t)M * 256 * 2, hipMemcpyDeviceToHost));
    if (pos) TCHK(hipMemcpy(yp16, p.yp16, (size_t)M * 256 * 2, hipMemcpyDeviceToHost));
    if (tail) TCHK(hipMemcpy(tail_out, p.tail_out, (size_t)M * p.tail_ld * 2, hipMemcpyDeviceToHost));
    return OPD_OK;
}

// Times `iters` launches of the fused encoder FFN on M rows of arbitrary data.
int opd_test_bench_enc_ffn(int M, int F, int iters, int dbg, int tail, int front, float* us_out) {
    if (M <= 0 || F <= 0 || F % 128) return tfail(OPD_EINVAL, "bench_enc_ffn: F must be a multiple of 128");
    DevMem dm;
    EncFfnParams p{}; p.dtype = g_test_dtype;
    uint16_t* x = dm.up<uint16_t>(nullptr, (size_t)M * 256);
    unsigned char* wp = dm.up<unsigned char>(nullptr, opd_encffn_pack_bytes(F, tail, 1));
    uint16_t* tout = tail ? dm.up<uint16_t>(nullptr, (size_t)M * tail * 256) : nullptr;
    if (tail && !tout) return tfail(OPD_ENOMEM, "bench alloc failed");
    float* f = dm.up<float>(nullptr, 1024);
    float* res = dm.up<float>(nullptr, (size_t)M * 256);
    uint16_t* y16 = dm.up<uint16_t>(nullptr, (size_t)M * 256);
    if (!x || !wp || !f || !res || !y16) return tfail(OPD_ENOMEM, "bench alloc failed");
    TCHK(hipMemset(x, 0x2c, (size_t)M * 256 * 2));
    TCHK(hipMemset(wp, 0x1c, opd_encffn_pack_bytes(F, tail, 1)));
    TCHK(hipMemset(f, 0, 4096));
    TCHK(hipMemset(res, 0, (size_t)M * 256 * 4));
    p.x = x; p.wpack = wp; p.b2 = f; p.gamma = f + 256; p.beta = f + 512; p.res32 = res; p.y32 = res; p.y16 = y16; p.M = M; p.F = F; p.dbg = dbg;
    p.pack_tail = tail; p.tail = tail; p.tail_pos = 0; p.tail_out = tout; p.tail_ld = tail * 256; p.pack_front = 1;
    if (front) { p.attn = x; p.x = nullptr; p.bo = f; p.gamma1 = f + 256; p.beta1 = f + 512; }
    for (int t = 0; t < tail && t < 16; ++t) p.tail_col[t] = 256 * t;
    hipEvent_t a, b;
    TCHK(hipEventCreate(&a)); TCHK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) TCHK(opd_launch_enc_ffn(p, nullptr));
    TCHK(hipEventRecord(a, nullptr));
    for (int i = 0; i < iters; ++i) TCHK(opd_launch_enc_ffn(p, nullptr));
    TCHK(hipEventRecord(b, nullptr));
    TCHK(hipEventSynchronize(b));
    float ms = 0.f;
    TCHK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *us_out = ms * 1000.f / iters;
    return OPD_OK;
}

int opd_test_gemm_k256(const uint16_t* x, const uint16_t* w, const float* bias, uint16_t* out16, float* out32, int M, int N,
                       int K, int bias_period, int relu) {
    DevMem dm;
    GemmK256Params p{}; p.dtype = g_test_dtype;
    const int slices = K / 256;
    p.x = dm.up(x, (size_t)M * K);
    p.w = dm.up(w, (size_t)N * K);
    p.bias = dm.up(bias, (size_t)N * (bias_period > 0 ? bias_period : 1));
    p.out16 = slices == 1 ? dm.up<uint16_t>(nullptr, (size_t)M * N) : nullptr;
    p.out32 = slices > 1 ? dm.up<float>(nullptr, (size_t)slices * M * N) : nullptr;
    float* sum = slices > 1 ? dm.up<float>(nullptr, (size_t)M * N) : nullptr;
    if (!p.x || !p.w || !p.bias || (!p.out16 && !p.out32)) return tfail(OPD_ENOMEM, "test alloc failed");
    p.M = M; p.N = N; p.ldx = K; p.ldw = K; p.slices = slices; p.bias_period = bias_period; p.relu = relu;
    TCHK(opd_launch_gemm_k256(p, nullptr));
    if (slices > 1) {
        if (N != 256) return tfail(OPD_EINVAL, "sliced test needs N == 256");
        TCHK(opd_launch_reduce_ln(p.out32, slices, (size_t)M * N, nullptr, nullptr, nullptr, sum, nullptr, M, nullptr));
        TCHK(hipDeviceSynchronize());
        TCHK(hipMemcpy(out32, sum, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    } else {
        TCHK(hipDeviceSynchronize());
        TCHK(hipMemcpy(out16, p.out16, (size_t)M * N * 2, hipMemcpyDeviceToHost));
    }
    return OPD_OK;
}

// Times one conv_gemm launch shape on device-resident random data (no host copies): average microseconds over `iters`.
int opd_test_bench_conv(int B, int H, int W, int Cin, int N, int KH, int stride, int with_res, int variant, int dbg, int iters,
                        float* us_out) {
    DevMem dm;
    const int pad = KH / 2, OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KH) / stride + 1;
    const size_t M = (size_t)B * OH * OW, K = (size_t)KH * KH * Cin;
    ConvGemmParams p{}; p.dtype = g_test_dtype;
    uint16_t* x = dm.up<uint16_t>(nullptr, (size_t)B * H * W * Cin);
    uint16_t* w = dm.up<uint16_t>(nullptr, (size_t)N * K);
    float* bias = dm.up<float>(nullptr, N);
    uint16_t* res = with_res ? dm.up<uint16_t>(nullptr, M * N) : nullptr;
    uint16_t* out = dm.up<uint16_t>(nullptr, M * N);
    float* zero = dm.up<float>(nullptr, 4096);
    if (!x || !w || !bias || !out || !zero || (with_res && !res)) return tfail(OPD_ENOMEM, "bench alloc failed");
    TCHK(hipMemset(x, 0x2c, (size_t)B * H * W * Cin * 2));  // fp16 0x2c2c ~ 0.065
    TCHK(hipMemset(w, 0x1c, (size_t)N * K * 2));
    TCHK(hipMemset(bias, 0, (size_t)N * 4));
    TCHK(hipMemset(zero, 0, 4096 * 4));
    if (res) TCHK(hipMemset(res, 0x2c, M * N * 2));
    p.x = x; p.w = w; p.bias = bias; p.res16 = res; p.out = out; p.zero16 = zero;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.N = N; p.KH = KH; p.KW = KH; p.stride = stride; p.pad = pad;
    p.M = (int)M; p.K = (int)K; p.relu = 1; p.dbg = dbg;
    apply_conv_flags(p, variant);   // (`variant`: the flag word of opd_test_set_conv_flags)
    hipEvent_t a, b;
    TCHK(hipEventCreate(&a)); TCHK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) TCHK(opd_launch_conv_gemm(p, nullptr));
    TCHK(hipEventRecord(a, nullptr));
    for (int i = 0; i < iters; ++i) TCHK(opd_launch_conv_gemm(p, nullptr));
    TCHK(hipEventRecord(b, nullptr));
    TCHK(hipEventSynchronize(b));
    float ms = 0.f;
    TCHK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *us_out = ms * 1000.f / iters;
    return OPD_OK;
}

// One traced launch of a layer shape (after `warm` untraced ones): trace_out [max_wgs][8] receives the per-workgroup phase stamps of
// conv_gemm_dma_kernel<..., TRACE>, *wgs_out the grid size.
int opd_test_trace_conv(int B, int H, int W, int Cin, int N, int KH, int stride, int with_res, int split_k, int warm, unsigned long long* trace_out,
                        int max_wgs, int* wgs_out) {
    const int dbg = 0;
    DevMem dm;
    const int pad = KH / 2, OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KH) / stride + 1;
    const size_t M = (size_t)B * OH * OW, K = (size_t)KH * KH * Cin;
    ConvGemmParams p{}; p.dtype = g_test_dtype;
    uint16_t* x = dm.up<uint16_t>(nullptr, (size_t)B * H * W * Cin);
    uint16_t* w = dm.up<uint16_t>(nullptr, (size_t)N * K);
    float* bias = dm.up<float>(nullptr, N);
    uint16_t* res = with_res ? dm.up<uint16_t>(nullptr, M * N) : nullptr;
    uint16_t* out = dm.up<uint16_t>(nullptr, M * N * (split_k > 1 ? 2 * (size_t)split_k : 1));   // split-K: fp32 slabs
    float* zero = dm.up<float>(nullptr, 4096);
    unsigned long long* tr = dm.up<unsigned long long>(nullptr, (size_t)max_wgs * 8);
    if (!x || !w || !bias || !out || !zero || !tr || (with_res && !res)) return tfail(OPD_ENOMEM, "trace alloc failed");
    TCHK(hipMemset(x, 0x2c, (size_t)B * H * W * Cin * 2));
    TCHK(hipMemset(w, 0x1c, (size_t)N * K * 2));
    TCHK(hipMemset(bias, 0, (size_t)N * 4));
    TCHK(hipMemset(zero, 0, 4096 * 4));
    TCHK(hipMemset(tr, 0, (size_t)max_wgs * 64));
    if (res) TCHK(hipMemset(res, 0x2c, M * N * 2));
    p.x = x; p.w = w; p.bias = bias; p.res16 = res; p.out = out; p.zero16 = zero;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.N = N; p.KH = KH; p.KW = KH; p.stride = stride; p.pad = pad;
    p.M = (int)M; p.K = (int)K; p.relu = 1; p.dbg = dbg;
    if (split_k > 1) { p.split_k = split_k; p.out_f32 = 1; p.relu = 0; }
    // `warm` traced launches back to back; the LAST THREE are kept (trace_out [3][max_wgs][8]): the spacing of their wall-clock stamps is
    // the cost of a launch boundary (drain of one kernel, dispatch of the next) on a busy stream
    unsigned long long* tr3 = dm.up<unsigned long long>(nullptr, (size_t)3 * max_wgs * 8);
    if (!tr3) return tfail(OPD_ENOMEM, "trace alloc failed");
    TCHK(hipMemset(tr3, 0, (size_t)3 * max_wgs * 64));
    p.trace = tr;
    for (int i = 0; i < warm; ++i) TCHK(opd_launch_conv_gemm(p, nullptr));
    for (int i = 0; i < 3; ++i) {
        p.trace = tr3 + (size_t)i * max_wgs * 8;
        TCHK(opd_launch_conv_gemm(p, nullptr));
    }
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(trace_out, tr3, (size_t)3 * max_wgs * 64, hipMemcpyDeviceToHost));
    const int bn = (N % 128 == 0) ? 128 : 64;   // (the launcher's choice is not exported: the caller reads stamps until the first all-zero row)
    (void)bn;
    *wgs_out = max_wgs;
    return OPD_OK;
}

// fused bottleneck tail vs. the caller's reference: x1 [B][H][W][C1], w1 [C1][3][3][C1], w2 [4*C1][C1], w3 [C3][4*C1]
// (plain K order: the hook applies opd_permute_k32 where the kernel wants it), res [M][4*C1] or null; outputs y [M][4*C1], z [M][C3] (C3 > 0).
int opd_test_btail(const uint16_t* x1, const uint16_t* w1, const float* b1, const uint16_t* w2, const float* b2,
                   const uint16_t* res, const uint16_t* w3, const float* b3, uint16_t* y, uint16_t* z, int B, int H, int W,
                   int C1, int C3, int stride) {
    if (!opd_btail_supported(C1, C3)) return tfail(OPD_EINVAL, "btail: unsupported (C1, C3)");
    DevMem dm;
    const int C2 = 4 * C1, OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
    const size_t M = (size_t)B * OH * OW;
    std::vector<uint16_t> w2p((size_t)C2 * C1), w3p((size_t)(C3 ? C3 : 1) * C2);
    if (C1 == 256) {   // the stage-3 kernel takes K-permuted 1x1 weights, the stage 1-2 kernel plain ones
        opd_permute_k32(w2, w2p.data(), C2, C1);
        if (C3) opd_permute_k32(w3, w3p.data(), C3, C2);
    } else {
        w2p.assign(w2, w2 + (size_t)C2 * C1);
        if (C3) w3p.assign(w3, w3 + (size_t)C3 * C2);
    }
    BtailParams p{}; p.dtype = g_test_dtype;
    p.x1 = dm.up(x1, (size_t)B * H * W * C1);
    p.w1 = dm.up(w1, (size_t)C1 * 9 * C1);
    p.b1 = dm.up(b1, C1);
    p.w2p = dm.up(w2p.data(), w2p.size());
    p.b2 = dm.up(b2, C2);
    p.res = res ? dm.up(res, M * C2) : nullptr;
    p.y = dm.up<uint16_t>(nullptr, M * C2);
    p.w3p = C3 ? dm.up(w3p.data(), w3p.size()) : nullptr;
    p.b3 = C3 ? dm.up(b3, C3) : nullptr;
    p.z = C3 ? dm.up<uint16_t>(nullptr, M * C3) : nullptr;
    if (!p.x1 || !p.w1 || !p.b1 || !p.w2p || !p.b2 || !p.y || (res && !p.res) || (C3 && (!p.w3p || !p.b3 || !p.z)))
        return tfail(OPD_ENOMEM, "test alloc failed");
    p.B = B; p.H = H; p.W = W; p.OH = OH; p.OW = OW; p.stride = stride; p.M = (int)M; p.C1 = C1; p.C3 = C3;
    TCHK(opd_launch_btail(p, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(y, p.y, M * C2 * 2, hipMemcpyDeviceToHost));
    if (C3) TCHK(hipMemcpy(z, p.z, M * C3 * 2, hipMemcpyDeviceToHost));
    return OPD_OK;
}

// Race screen for the fused tails (the stage-3 kernel reads LDS-DMA data by counted waits and raw barriers: a misplaced wait shows as a
// rare wrong tile that comes and goes with timing): `reps` launches on the same device-resident operands, position-weighted checksums of
// y and z after each, *n_diff = number of launches whose checksums differ from the first launch's.  A second stream keeps the memory system
// busy meanwhile (a 256-MiB device-to-device copy per launch) so that DMA latencies vary between launches.
int opd_test_btail_repeat(const uint16_t* x1, const uint16_t* w1, const float* b1, const uint16_t* w2, const float* b2, const uint16_t* res,
                          const uint16_t* w3, const float* b3, int B, int H, int W, int C1, int C3, int reps, int* n_diff) {
    if (!opd_btail_supported(C1, C3) || !C3 || reps < 2 || !n_diff) return tfail(OPD_EINVAL, "btail_repeat: bad arguments");
    DevMem dm;
    const int C2 = 4 * C1;
    const size_t M = (size_t)B * H * W;
    std::vector<uint16_t> w2p((size_t)C2 * C1), w3p((size_t)C3 * C2);
    if (C1 == 256) {
        opd_permute_k32(w2, w2p.data(), C2, C1);
        opd_permute_k32(w3, w3p.data(), C3, C2);
    } else {
        w2p.assign(w2, w2 + (size_t)C2 * C1);
        w3p.assign(w3, w3 + (size_t)C3 * C2);
    }
    BtailParams p{}; p.dtype = g_test_dtype;
    p.x1 = dm.up(x1, M * C1); p.w1 = dm.up(w1, (size_t)C1 * 9 * C1); p.b1 = dm.up(b1, C1); p.w2p = dm.up(w2p.data(), w2p.size());
    p.b2 = dm.up(b2, C2); p.res = dm.up(res, M * C2); p.y = dm.up<uint16_t>(nullptr, M * C2); p.w3p = dm.up(w3p.data(), w3p.size());
    p.b3 = dm.up(b3, C3); p.z = dm.up<uint16_t>(nullptr, M * C3);
    const size_t noise_bytes = (size_t)256 << 20;
    unsigned char* noise = dm.up<unsigned char>(nullptr, 2 * noise_bytes);
    unsigned long long* sums = dm.up<unsigned long long>(nullptr, (size_t)reps * 2 * OPD_TAP_BLOCKS);
    if (!p.x1 || !p.w1 || !p.b1 || !p.w2p || !p.b2 || !p.res || !p.y || !p.w3p || !p.b3 || !p.z || !noise || !sums) return tfail(OPD_ENOMEM, "test alloc failed");
    p.B = B; p.H = H; p.W = W; p.OH = H; p.OW = W; p.stride = 1; p.M = (int)M; p.C1 = C1; p.C3 = C3;
    hipStream_t side = nullptr;
    TCHK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    for (int r = 0; r < reps; ++r) {
        p.rev = r & 1;
        (void)hipMemcpyAsync(noise + ((r & 1) ? noise_bytes : 0), noise + ((r & 1) ? 0 : noise_bytes), noise_bytes, hipMemcpyDeviceToDevice, side);
        TCHK(hipMemsetAsync(p.y, 0xff, M * C2 * 2, nullptr));
        TCHK(hipMemsetAsync(p.z, 0xff, M * C3 * 2, nullptr));
        TCHK(opd_launch_btail(p, nullptr));
        TCHK(opd_launch_checksum(p.y, M * C2 * 2, sums + (size_t)(2 * r) * OPD_TAP_BLOCKS, nullptr));
        TCHK(opd_launch_checksum(p.z, M * C3 * 2, sums + (size_t)(2 * r + 1) * OPD_TAP_BLOCKS, nullptr));
    }
    TCHK(hipDeviceSynchronize());
    (void)hipStreamDestroy(side);
    std::vector<unsigned long long> h((size_t)reps * 2 * OPD_TAP_BLOCKS);
    TCHK(hipMemcpy(h.data(), sums, h.size() * 8, hipMemcpyDeviceToHost));
    int diff = 0;
    for (int r = 1; r < reps; ++r)
        if (memcmp(h.data() + (size_t)(2 * r) * OPD_TAP_BLOCKS, h.data(), 2 * OPD_TAP_BLOCKS * 8) != 0) ++diff;
    *n_diff = diff;
    return OPD_OK;
}

// fused tail WITH the block's shortcut convolution inside (first block of stage 1): xs [M][64] = the shortcut's input at the output
// resolution, wsc [256][64]; b2sc = b2 + the shortcut's bias.  C1 = 64, C3 = 64, stride 1.
int opd_test_btail_sc(const uint16_t* x1, const uint16_t* w1, const float* b1, const uint16_t* w2, const float* b2sc, const uint16_t* xs,
                      const uint16_t* wsc, const uint16_t* w3, const float* b3, uint16_t* y, uint16_t* z, int B, int H, int W) {
    DevMem dm;
    const int C1 = 64, C2 = 256, C3 = 64;
    const size_t M = (size_t)B * H * W;
    std::vector<uint16_t> w2p((size_t)C2 * C1), w3p((size_t)C3 * C2);
    if (C1 == 256) {
        opd_permute_k32(w2, w2p.data(), C2, C1);
        opd_permute_k32(w3, w3p.data(), C3, C2);
    } else {
        w2p.assign(w2, w2 + (size_t)C2 * C1);
        w3p.assign(w3, w3 + (size_t)C3 * C2);
    }
    BtailParams p{}; p.dtype = g_test_dtype;
    p.x1 = dm.up(x1, M * C1);
    p.w1 = dm.up(w1, (size_t)C1 * 9 * C1);
    p.b1 = dm.up(b1, C1);
    p.w2p = dm.up(w2p.data(), w2p.size());
    p.b2 = dm.up(b2sc, C2);
    p.xs = dm.up(xs, M * 64);
    p.wsc = dm.up(wsc, (size_t)C2 * 64);
    p.y = dm.up<uint16_t>(nullptr, M * C2);
    p.w3p = dm.up(w3p.data(), w3p.size());
    p.b3 = dm.up(b3, C3);
    p.z = dm.up<uint16_t>(nullptr, M * C3);
    if (!p.x1 || !p.w1 || !p.b1 || !p.w2p || !p.b2 || !p.xs || !p.wsc || !p.y || !p.w3p || !p.b3 || !p.z) return tfail(OPD_ENOMEM, "test alloc failed");
    p.B = B; p.H = H; p.W = W; p.OH = H; p.OW = W; p.stride = 1; p.M = (int)M; p.C1 = C1; p.C3 = C3;
    TCHK(opd_launch_btail(p, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(y, p.y, M * C2 * 2, hipMemcpyDeviceToHost));
    TCHK(hipMemcpy(z, p.z, M * C3 * 2, hipMemcpyDeviceToHost));
    return OPD_OK;
}

// Stage 1's first two tails and its last one, both ways (kernels_btail.hip, round 5).  Blocks a (shortcut inside), b, c on x1 [B][H][W][64] /
// xs [M][64]; weights as in opd_test_btail_sc / opd_test_btail (w3a, w3b: [64][256], w3c: [128][256]).
//   old: tail a stores y_a -> tail b reads it back as its residual -> tail c stores all of y_c;
//   new: tail a stores a1 only -> tail b REBUILDS y_a (rc = 1) -> tail c stores y_c at even (oh, ow) only (y_stride2; the buffer is pre-filled
//        with `fill` so the caller sees what was not written).
//   new2: as new, but tail b stores its a1 too (no y_b at all) and tail c rebuilds BOTH previous outputs (rc = 2, btail_rc2_kernel).
// Outputs: yb / zb [M][256] / [M][64], yc / zc [M][256] / [M][128], once per route (index 0 old, 1 new, 2 new2; yb[2] is left untouched).
int opd_test_btail_chain(const uint16_t* x1, const uint16_t* xs, const uint16_t* const* w1, const float* const* b1, const uint16_t* const* w2,
                         const float* const* b2, const uint16_t* wsc, const uint16_t* const* w3, const float* const* b3, uint16_t* const* yb,
                         uint16_t* const* zb, uint16_t* const* yc, uint16_t* const* zc, int B, int H, int W, int fill) {
    DevMem dm;
    const size_t M = (size_t)B * H * W;
    const uint16_t* d_x1 = dm.up(x1, M * 64);
    const uint16_t* d_xs = dm.up(xs, M * 64);
    const uint16_t* d_wsc = dm.up(wsc, (size_t)256 * 64);
    const uint16_t *d_w1[3], *d_w2[3], *d_w3[3];
    const float *d_b1[3], *d_b2[3], *d_b3[3];
    for (int i = 0; i < 3; ++i) {
        const int c3 = i == 2 ? 128 : 64;
        d_w1[i] = dm.up(w1[i], (size_t)64 * 9 * 64); d_b1[i] = dm.up(b1[i], 64);
        d_w2[i] = dm.up(w2[i], (size_t)256 * 64); d_b2[i] = dm.up(b2[i], 256);
        d_w3[i] = dm.up(w3[i], (size_t)c3 * 256); d_b3[i] = dm.up(b3[i], c3);
        if (!d_w1[i] || !d_b1[i] || !d_w2[i] || !d_b2[i] || !d_w3[i] || !d_b3[i]) return tfail(OPD_ENOMEM, "test alloc failed");
    }
    uint16_t* ya = dm.up<uint16_t>(nullptr, M * 256);
    uint16_t* za = dm.up<uint16_t>(nullptr, M * 64);
    uint16_t* a1a = dm.up<uint16_t>(nullptr, M * 64);
    uint16_t* a1b = dm.up<uint16_t>(nullptr, M * 64);
    uint16_t* d_yb = dm.up<uint16_t>(nullptr, M * 256);
    uint16_t* d_zb = dm.up<uint16_t>(nullptr, M * 64);
    uint16_t* d_yc = dm.up<uint16_t>(nullptr, M * 256);
    uint16_t* d_zc = dm.up<uint16_t>(nullptr, M * 128);
    if (!d_x1 || !d_xs || !d_wsc || !ya || !za || !a1a || !a1b || !d_yb || !d_zb || !d_yc || !d_zc) return tfail(OPD_ENOMEM, "test alloc failed");
    for (int route = 0; route < 3; ++route) {
        TCHK(hipMemset(ya, 0xEE, M * 256 * 2));
        TCHK(hipMemset(d_yc, fill, M * 256 * 2));
        auto base = [&](int i, const uint16_t* in, uint16_t* y, uint16_t* z, int c3) {
            BtailParams p{}; p.dtype = g_test_dtype;
            p.x1 = in; p.w1 = d_w1[i]; p.b1 = d_b1[i]; p.w2p = d_w2[i]; p.b2 = d_b2[i]; p.y = y; p.w3p = d_w3[i]; p.b3 = d_b3[i]; p.z = z;
            p.B = B; p.H = H; p.W = W; p.OH = H; p.OW = W; p.stride = 1; p.M = (int)M; p.C1 = 64; p.C3 = c3;
            return p;
        };
        BtailParams pa = base(0, d_x1, ya, za, 64);
        pa.xs = d_xs; pa.wsc = d_wsc;
        if (route) { pa.y = nullptr; pa.a1_out = a1a; }
        TCHK(opd_launch_btail(pa, nullptr));
        BtailParams pb = base(1, za, d_yb, d_zb, 64);
        if (route) { pb.rc = 1; pb.rc_a1[0] = a1a; pb.rc_xs = d_xs; pb.rc_w2[0] = d_w2[0]; pb.rc_wsc = d_wsc; pb.rc_b[0] = d_b2[0]; }
        else pb.res = ya;
        if (route == 2) { pb.y = nullptr; pb.a1_out = a1b; }
        TCHK(opd_launch_btail(pb, nullptr));
        BtailParams pc = base(2, d_zb, d_yc, d_zc, 128);
        pc.y_stride2 = route ? 1 : 0;
        if (route == 2) {
            pc.rc = 2; pc.rc_xs = d_xs; pc.rc_wsc = d_wsc;
            pc.rc_a1[0] = a1b; pc.rc_w2[0] = d_w2[1]; pc.rc_b[0] = d_b2[1];
            pc.rc_a1[1] = a1a; pc.rc_w2[1] = d_w2[0]; pc.rc_b[1] = d_b2[0];
        } else {
            pc.res = d_yb;
        }
        TCHK(opd_launch_btail(pc, nullptr));
        TCHK(hipDeviceSynchronize());
        if (route < 2) TCHK(hipMemcpy(yb[route], d_yb, M * 256 * 2, hipMemcpyDeviceToHost));
        TCHK(hipMemcpy(zb[route], d_zb, M * 64 * 2, hipMemcpyDeviceToHost));
        TCHK(hipMemcpy(yc[route], d_yc, M * 256 * 2, hipMemcpyDeviceToHost));
        TCHK(hipMemcpy(zc[route], d_zc, M * 128 * 2, hipMemcpyDeviceToHost));
    }
    return OPD_OK;
}

// Times the fused tail (us_out[0]) and the three unfused launches it replaces (us_out[1..3]: c1, c2, c0') on
// device-resident data of the given shape.
// Two warm launches, then one traced launch of a fused tail: trace_out [wgs][16] (kernels_btail.hip, TRACE), *wgs_out = grid size
int opd_test_trace_btail(int B, int H, int W, int C1, int C3, int dbg, unsigned long long* trace_out, int max_wgs, int* wgs_out) {
    if (!opd_btail_supported(C1, C3)) return tfail(OPD_EINVAL, "btail: unsupported (C1, C3)");
    DevMem dm;
    const int C2 = 4 * C1;
    const size_t M = (size_t)B * H * W;
    const int wgs = (int)((M + 127) / 128);
    if (wgs > max_wgs) return tfail(OPD_EINVAL, "trace buffer too small");
    uint16_t* x1 = dm.up<uint16_t>(nullptr, M * C1);
    uint16_t* w1 = dm.up<uint16_t>(nullptr, (size_t)C1 * 9 * C1);
    uint16_t* w2 = dm.up<uint16_t>(nullptr, (size_t)C2 * C1);
    uint16_t* w3 = dm.up<uint16_t>(nullptr, (size_t)C3 * C2);
    float* bias = dm.up<float>(nullptr, C2);
    uint16_t* res = dm.up<uint16_t>(nullptr, M * C2);
    uint16_t* y = dm.up<uint16_t>(nullptr, M * C2);
    uint16_t* z = dm.up<uint16_t>(nullptr, M * C3);
    unsigned long long* tr = dm.up<unsigned long long>(nullptr, (size_t)wgs * 16);
    if (!x1 || !w1 || !w2 || !w3 || !bias || !res || !y || !z || !tr) return tfail(OPD_ENOMEM, "trace alloc failed");
    TCHK(hipMemset(x1, 0x2c, M * C1 * 2));
    TCHK(hipMemset(w1, 0x1c, (size_t)C1 * 9 * C1 * 2));
    TCHK(hipMemset(w2, 0x1c, (size_t)C2 * C1 * 2));
    TCHK(hipMemset(w3, 0x1c, (size_t)C3 * C2 * 2));
    TCHK(hipMemset(bias, 0, (size_t)C2 * 4));
    TCHK(hipMemset(res, 0x2c, M * C2 * 2));
    TCHK(hipMemset(tr, 0, (size_t)wgs * 128));
    BtailParams p{}; p.dtype = g_test_dtype;
    p.x1 = x1; p.w1 = w1; p.b1 = bias; p.w2p = w2; p.b2 = bias; p.res = res; p.y = y; p.w3p = w3; p.b3 = bias; p.z = z;
    p.B = B; p.H = H; p.W = W; p.OH = H; p.OW = W; p.stride = 1; p.M = (int)M; p.C1 = C1; p.C3 = C3; p.dbg = dbg;
    for (int i = 0; i < 2; ++i) TCHK(opd_launch_btail(p, nullptr));
    p.trace = tr;
    TCHK(opd_launch_btail(p, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(trace_out, tr, (size_t)wgs * 128, hipMemcpyDeviceToHost));
    *wgs_out = wgs;
    return OPD_OK;
}

int opd_test_bench_btail(int B, int H, int W, int C1, int C3, int stride, int dbg, int iters, float* us_out) {
    if (!opd_btail_supported(C1, C3)) return tfail(OPD_EINVAL, "btail: unsupported (C1, C3)");
    DevMem dm;
    const int C2 = 4 * C1, OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
    const size_t M = (size_t)B * OH * OW;
    uint16_t* x1 = dm.up<uint16_t>(nullptr, (size_t)B * H * W * C1);
    uint16_t* w1 = dm.up<uint16_t>(nullptr, (size_t)C1 * 9 * C1);
    uint16_t* w2 = dm.up<uint16_t>(nullptr, (size_t)C2 * C1);
    uint16_t* w3 = dm.up<uint16_t>(nullptr, (size_t)(C3 ? C3 : 64) * C2);
    float* bias = dm.up<float>(nullptr, C2);
    uint16_t* res = dm.up<uint16_t>(nullptr, M * C2);
    uint16_t* a1 = dm.up<uint16_t>(nullptr, M * C1);
    uint16_t* y = dm.up<uint16_t>(nullptr, M * C2);
    uint16_t* z = dm.up<uint16_t>(nullptr, M * (C3 ? C3 : 64));
    float* zero = dm.up<float>(nullptr, 4096);
    if (!x1 || !w1 || !w2 || !w3 || !bias || !res || !a1 || !y || !z || !zero) return tfail(OPD_ENOMEM, "bench alloc failed");
    TCHK(hipMemset(x1, 0x2c, (size_t)B * H * W * C1 * 2));
    TCHK(hipMemset(w1, 0x1c, (size_t)C1 * 9 * C1 * 2));
    TCHK(hipMemset(w2, 0x1c, (size_t)C2 * C1 * 2));
    TCHK(hipMemset(w3, 0x1c, (size_t)(C3 ? C3 : 64) * C2 * 2));
    TCHK(hipMemset(bias, 0, (size_t)C2 * 4));
    TCHK(hipMemset(res, 0x2c, M * C2 * 2));
    TCHK(hipMemset(zero, 0, 4096 * 4));
    BtailParams p{}; p.dtype = g_test_dtype;
    p.x1 = x1; p.w1 = w1; p.b1 = bias; p.w2p = w2; p.b2 = bias; p.res = res; p.y = y; p.w3p = w3; p.b3 = bias; p.z = z;
    p.B = B; p.H = H; p.W = W; p.OH = OH; p.OW = OW; p.stride = stride; p.M = (int)M; p.C1 = C1; p.C3 = C3; p.dbg = dbg;
    ConvGemmParams c[3] = {};
    c[0].x = x1; c[0].w = w1; c[0].out = a1; c[0].B = B; c[0].H = H; c[0].W = W; c[0].Cin = C1; c[0].OH = OH; c[0].OW = OW; c[0].N = C1;
    c[0].KH = c[0].KW = 3; c[0].stride = stride; c[0].pad = 1; c[0].K = 9 * C1;
    c[1].x = a1; c[1].w = w2; c[1].res16 = res; c[1].out = y; c[1].B = B; c[1].H = OH; c[1].W = OW; c[1].Cin = C1; c[1].OH = OH; c[1].OW = OW;
    c[1].N = C2; c[1].KH = c[1].KW = 1; c[1].stride = 1; c[1].K = C1;
    c[2].x = y; c[2].w = w3; c[2].out = z; c[2].B = B; c[2].H = OH; c[2].W = OW; c[2].Cin = C2; c[2].OH = OH; c[2].OW = OW; c[2].N = C3 ? C3 : 64;
    c[2].KH = c[2].KW = 1; c[2].stride = 1; c[2].K = C2;
    for (auto& q : c) { q.bias = bias; q.zero16 = zero; q.M = (int)M; q.relu = 1; }
    hipEvent_t ev[2];
    TCHK(hipEventCreate(&ev[0])); TCHK(hipEventCreate(&ev[1]));
    for (int k = 0; k < 4; ++k) {
        if (k == 3 && !C3) { us_out[3] = 0.f; break; }
        if (k > 0 && dbg) { us_out[k] = 0.f; continue; }
        for (int i = -2; i < iters; ++i) {
            if (i == 0) TCHK(hipEventRecord(ev[0], nullptr));
            if (k == 0) TCHK(opd_launch_btail(p, nullptr));
            else TCHK(opd_launch_conv_gemm(c[k - 1], nullptr));
        }
        TCHK(hipEventRecord(ev[1], nullptr));
        TCHK(hipEventSynchronize(ev[1]));
        float ms = 0.f;
        TCHK(hipEventElapsedTime(&ms, ev[0], ev[1]));
        us_out[k] = ms * 1000.f / iters;
    }
    (void)hipEventDestroy(ev[0]); (void)hipEventDestroy(ev[1]);
    return OPD_OK;
}

// Times `iters` launches of the attention kernel on caller-supplied operands with leading dimension ld (768 = the fused QKV buffer).
int opd_test_bench_attention(const uint16_t* q, const uint16_t* k, const uint16_t* v, int B, int heads, int Lq, int Lk, int ldq, int ldkv,
                             float scale, int iters, float* us_out) {
    DevMem dm;
    AttnParams p{}; p.dtype = g_test_dtype;
    p.q = dm.up(q, (size_t)B * Lq * ldq);
    p.k = dm.up(k, (size_t)B * Lk * ldkv);
    p.v = dm.up(v, (size_t)B * Lk * ldkv);
    p.o = dm.up<uint16_t>(nullptr, (size_t)B * Lq * heads * 32);
    if (!p.q || !p.k || !p.v || !p.o) return tfail(OPD_ENOMEM, "bench alloc failed");
    p.B = B; p.heads = heads; p.Lq = Lq; p.Lk = Lk; p.ldq = ldq; p.ldk = p.ldv = ldkv; p.ldo = heads * 32; p.scale = scale;
    hipEvent_t a, b;
    TCHK(hipEventCreate(&a)); TCHK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) TCHK(opd_launch_attention(p, nullptr));
    TCHK(hipEventRecord(a, nullptr));
    for (int i = 0; i < iters; ++i) TCHK(opd_launch_attention(p, nullptr));
    TCHK(hipEventRecord(b, nullptr));
    TCHK(hipEventSynchronize(b));
    float ms = 0.f;
    TCHK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *us_out = ms * 1000.f / iters;
    return OPD_OK;
}

// One traced launch (after 2 untraced ones): trace_out [max_wgs][8] = per-workgroup sums of wave 0's cycles in the five phases of a
// key tile (loads issued | S + max + branch | exp + PV | wait for the next tile's loads | LDS stores | barrier), total, tiles.
int opd_test_trace_attention(const uint16_t* q, const uint16_t* k, const uint16_t* v, int B, int heads, int Lq, int Lk, int ldq, int ldkv,
                             float scale, unsigned long long* trace_out, int max_wgs, int* wgs_out) {
    DevMem dm;
    AttnParams p{}; p.dtype = g_test_dtype;
    p.q = dm.up(q, (size_t)B * Lq * ldq);
    p.k = dm.up(k, (size_t)B * Lk * ldkv);
    p.v = dm.up(v, (size_t)B * Lk * ldkv);
    p.o = dm.up<uint16_t>(nullptr, (size_t)B * Lq * heads * 32);
    const int total = ((Lq + 63) / 64) * heads * B, grid = 8 * ((total + 7) / 8);
    unsigned long long* tr = dm.up<unsigned long long>(nullptr, (size_t)grid * 12);
    if (!p.q || !p.k || !p.v || !p.o || !tr) return tfail(OPD_ENOMEM, "trace alloc failed");
    TCHK(hipMemset(tr, 0, (size_t)grid * 96));
    p.B = B; p.heads = heads; p.Lq = Lq; p.Lk = Lk; p.ldq = ldq; p.ldk = p.ldv = ldkv; p.ldo = heads * 32; p.scale = scale;
    for (int i = 0; i < 2; ++i) TCHK(opd_launch_attention(p, nullptr));
    p.trace = tr;
    TCHK(opd_launch_attention(p, nullptr));
    TCHK(hipDeviceSynchronize());
    const int n = grid < max_wgs ? grid : max_wgs;
    TCHK(hipMemcpy(trace_out, tr, (size_t)n * 96, hipMemcpyDeviceToHost));
    *wgs_out = n;
    return OPD_OK;
}

int opd_test_attention(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* o, int B, int heads, int Lq, int Lk,
                       float scale) {
    DevMem dm;
    const int D = heads * 32;
    AttnParams p{}; p.dtype = g_test_dtype;
    p.q = dm.up(q, (size_t)B * Lq * D);
    p.k = dm.up(k, (size_t)B * Lk * D);
    p.v = dm.up(v, (size_t)B * Lk * D);
    p.o = dm.up<uint16_t>(nullptr, (size_t)B * Lq * D);
    if (!p.q || !p.k || !p.v || !p.o) return tfail(OPD_ENOMEM, "test alloc failed");
    p.B = B; p.heads = heads; p.Lq = Lq; p.Lk = Lk; p.ldq = p.ldk = p.ldv = p.ldo = D; p.scale = scale;
    TCHK(opd_launch_attention(p, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(o, p.o, (size_t)B * Lq * D * 2, hipMemcpyDeviceToHost));
    return OPD_OK;
}

// attention with a per-frame key mask: key k = (k / key_row, k % key_row) is valid inside key_valid[b] = (rows, cols)
int opd_test_attention_masked(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* o, int B, int heads, int Lq,
                              int Lk, float scale, const int32_t* key_valid, int key_row) {
    DevMem dm;
    const int D = heads * 32;
    AttnParams p{}; p.dtype = g_test_dtype;
    p.q = dm.up(q, (size_t)B * Lq * D);
    p.k = dm.up(k, (size_t)B * Lk * D);
    p.v = dm.up(v, (size_t)B * Lk * D);
    p.o = dm.up<uint16_t>(nullptr, (size_t)B * Lq * D);
    p.key_valid = dm.up(key_valid, (size_t)B * 2);
    if (!p.q || !p.k || !p.v || !p.o || !p.key_valid) return tfail(OPD_ENOMEM, "test alloc failed");
    p.B = B; p.heads = heads; p.Lq = Lq; p.Lk = Lk; p.ldq = p.ldk = p.ldv = p.ldo = D; p.scale = scale;
    p.key_row = key_row;
    TCHK(opd_launch_attention(p, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(o, p.o, (size_t)B * Lq * D * 2, hipMemcpyDeviceToHost));
    return OPD_OK;
}

int opd_test_layernorm(const float* x, const float* g, const float* b, float* y, uint16_t* y16, int rows) {
    DevMem dm;
    const float* dx = dm.up(x, (size_t)rows * 256);
    const float* dg = dm.up(g, 256);
    const float* db = dm.up(b, 256);
    float* dy = dm.up<float>(nullptr, (size_t)rows * 256);
    uint16_t* dy16 = dm.up<uint16_t>(nullptr, (size_t)rows * 256);
    if (!dx || !dg || !db || !dy || !dy16) return tfail(OPD_ENOMEM, "test alloc failed");
    TCHK(opd_launch_layernorm(dx, dg, db, dy, dy16, rows, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(y, dy, (size_t)rows * 256 * 4, hipMemcpyDeviceToHost));
    TCHK(hipMemcpy(y16, dy16, (size_t)rows * 256 * 2, hipMemcpyDeviceToHost));
    return OPD_OK;
}

int opd_test_maxpool(const uint16_t* x, uint16_t* out, int B, int H, int W, int C, int OH, int OW) {
    DevMem dm;
    const uint16_t* dx = dm.up(x, (size_t)B * H * W * C);
    uint16_t* dout = dm.up<uint16_t>(nullptr, (size_t)B * OH * OW * C);
    if (!dx || !dout) return tfail(OPD_ENOMEM, "test alloc failed");
    TCHK(opd_launch_maxpool(dx, dout, B, H, W, C, OH, OW, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(out, dout, (size_t)B * OH * OW * C * 2, hipMemcpyDeviceToHost));
    return OPD_OK;
}

// valid_hw (nullable): host [B][2] per-frame (h, w) inside the H x W canvas
int opd_test_preprocess_u8(const uint8_t* frames, uint16_t* out, int B, int H, int W, int Hp, int Wp, const int32_t* valid_hw) {
    DevMem dm;
    const uint8_t* din = dm.up(frames, (size_t)B * H * W * 3);
    uint16_t* dout = dm.up<uint16_t>(nullptr, (size_t)B * Hp * Wp * 4);
    const int32_t* dvalid = valid_hw ? dm.up(valid_hw, (size_t)B * 2) : nullptr;
    if (!din || !dout || (valid_hw && !dvalid)) return tfail(OPD_ENOMEM, "test alloc failed");
    TCHK(opd_launch_preprocess_u8(din, dout, B, H, W, Hp, Wp, dvalid, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(out, dout, (size_t)B * Hp * Wp * 8, hipMemcpyDeviceToHost));
    return OPD_OK;
}

// Stem on the zero-bordered NHWC4 image through the LDS-DMA kernel (stem mode 2): x4p [B][Hp][Wp][4], w [64][8][8][4].
int opd_test_stem2(const uint16_t* x4p, const uint16_t* w, const float* bias, uint16_t* out, int B, int Hp, int Wp, int OH, int OW) {
    DevMem dm;
    ConvGemmParams p{}; p.dtype = g_test_dtype;
    const size_t M = (size_t)B * OH * OW;
    p.x = dm.up(x4p, (size_t)B * Hp * Wp * 4);
    p.w = dm.up(w, (size_t)64 * 256);
    p.bias = dm.up(bias, 64);
    std::vector<float> zeros(64, 0.f);
    p.zero16 = dm.up(zeros.data(), 64);
    p.out = dm.up<uint16_t>(nullptr, M * 64);
    if (!p.x || !p.w || !p.bias || !p.zero16 || !p.out) return tfail(OPD_ENOMEM, "test alloc failed");
    p.B = B; p.H = Hp; p.W = Wp; p.Cin = 256; p.OH = OH; p.OW = OW; p.N = 64; p.KH = 1; p.KW = 1; p.stride = 2; p.pad = 0;
    p.M = (int)M; p.K = 256; p.relu = 1; p.stem = 2;
    TCHK(opd_launch_conv_gemm(p, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(out, p.out, M * 64 * 2, hipMemcpyDeviceToHost));
    return OPD_OK;
}

// Fused stem + max-pool on the zero-bordered NHWC4 image: out = pooled [B][PH][PW][64] fp16
int opd_test_stem_pool(const uint16_t* x4p, const uint16_t* w, const float* bias, uint16_t* out, int B, int Hp, int Wp, int OH, int OW,
                       int PH, int PW) {
    DevMem dm;
    const uint16_t* dx = dm.up(x4p, (size_t)B * Hp * Wp * 4);
    const uint16_t* dw = dm.up(w, (size_t)64 * 256);
    const float* db = dm.up(bias, 64);
    uint16_t* dout = dm.up<uint16_t>(nullptr, (size_t)B * PH * PW * 64);
    if (!dx || !dw || !db || !dout) return tfail(OPD_ENOMEM, "test alloc failed");
    TCHK(opd_launch_stem_pool(dx, dw, db, dout, B, Hp, Wp, OH, OW, PH, PW, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(out, dout, (size_t)B * PH * PW * 64 * 2, hipMemcpyDeviceToHost));
    return OPD_OK;
}

// Pre-processing + stem + max-pool in one launch against the two-kernel path on the same uint8 frames: out_fused / out_split =
// pooled [B][PH][PW][64] fp16 (the caller checks bit-equality); valid_hw nullable [B][2]
int opd_test_stem_pool_u8(const uint8_t* frames, const int32_t* valid_hw, const uint16_t* w, const float* bias, uint16_t* out_fused,
                          uint16_t* out_split, int B, int H, int W) {
    DevMem dm;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1, PH = (OH - 1) / 2 + 1, PW = (OW - 1) / 2 + 1, Hp = 2 * OH + 6, Wp = 2 * OW + 6;
    const uint8_t* df = dm.up(frames, (size_t)B * H * W * 3);
    const int32_t* dv = valid_hw ? dm.up(valid_hw, (size_t)2 * B) : nullptr;
    const uint16_t* dw = dm.up(w, (size_t)64 * 256);
    const float* db = dm.up(bias, 64);
    uint16_t* dx = dm.up<uint16_t>(nullptr, (size_t)B * Hp * Wp * 4);
    uint16_t* d1 = dm.up<uint16_t>(nullptr, (size_t)B * PH * PW * 64);
    uint16_t* d2 = dm.up<uint16_t>(nullptr, (size_t)B * PH * PW * 64);
    if (!df || !dw || !db || !dx || !d1 || !d2 || (valid_hw && !dv)) return tfail(OPD_ENOMEM, "test alloc failed");
    TCHK(opd_launch_stem_pool_u8(df, dv, dw, db, d1, B, H, W, OH, OW, PH, PW, nullptr));
    TCHK(opd_launch_preprocess_u8(df, dx, B, H, W, Hp, Wp, dv, nullptr));
    TCHK(opd_launch_stem_pool(dx, dw, db, d2, B, Hp, Wp, OH, OW, PH, PW, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(out_fused, d1, (size_t)B * PH * PW * 64 * 2, hipMemcpyDeviceToHost));
    TCHK(hipMemcpy(out_split, d2, (size_t)B * PH * PW * 64 * 2, hipMemcpyDeviceToHost));
    return OPD_OK;
}

// heads_kernel alone: hs [rows][256] fp32 (+ optional final LayerNorm), weights in the reference's [out][in] layout (the hook
// transposes them as opd_model.cpp does); logits [rows][ncls], boxes [rows][4]
static int g_test_heads2 = 1;   // heads hooks: 1 = heads2_kernel (split fp16 operands), 0 = heads_kernel (fp32 matrix pipe)
int opd_test_set_heads2(int on) { g_test_heads2 = on; return OPD_OK; }
// (the three 256-wide layers as split fp16 pairs in fragment order, class matrix padded to 128 rows)
static bool heads_frags(DevMem& dm, HeadParams& p, const float* wc, const float* w1, const float* w2, int ncls) {
    if (!g_test_heads2 || ncls > 128) return true;
    std::vector<float> wcp((size_t)128 * 256, 0.f);
    std::copy(wc, wc + (size_t)ncls * 256, wcp.begin());
    std::vector<uint16_t> f((size_t)2 * 128 * 256);
    opd_split_f16_frag(wcp.data(), 128, 256, f.data());
    p.wc_f = dm.up(f.data(), f.size());
    std::vector<uint16_t> f2((size_t)2 * 256 * 256);
    opd_split_f16_frag(w1, 256, 256, f2.data());
    p.w1_f = dm.up(f2.data(), f2.size());
    opd_split_f16_frag(w2, 256, 256, f2.data());
    p.w2_f = dm.up(f2.data(), f2.size());
    return p.wc_f && p.w1_f && p.w2_f;
}
int opd_test_heads(const float* hs, const float* ln_g, const float* ln_b, const float* wc, const float* bc, const float* w1, const float* b1,
                   const float* w2, const float* b2, const float* w3, const float* b3, float* logits, float* boxes, int rows, int ncls) {
    DevMem dm;
    auto tr = [&](const float* w, int O, int I) -> const float* {
        std::vector<float> t((size_t)O * I);
        for (int o = 0; o < O; ++o)
            for (int i = 0; i < I; ++i) t[(size_t)i * O + o] = w[(size_t)o * I + i];
        return dm.up(t.data(), t.size());
    };
    HeadParams p{};
    p.hs = dm.up(hs, (size_t)rows * 256);
    p.ln_gamma = ln_g ? dm.up(ln_g, 256) : nullptr;
    p.ln_beta = ln_b ? dm.up(ln_b, 256) : nullptr;
    p.wc = tr(wc, ncls, 256); p.bc = dm.up(bc, ncls);
    p.w1 = tr(w1, 256, 256); p.b1 = dm.up(b1, 256);
    p.w2 = tr(w2, 256, 256); p.b2 = dm.up(b2, 256);
    p.w3 = tr(w3, 4, 256); p.b3 = dm.up(b3, 4);
    p.logits = dm.up<float>(nullptr, (size_t)rows * ncls);
    p.boxes = dm.up<float>(nullptr, (size_t)rows * 4);
    if (!p.hs || !p.wc || !p.bc || !p.w1 || !p.b1 || !p.w2 || !p.b2 || !p.w3 || !p.b3 || !p.logits || !p.boxes) return tfail(OPD_ENOMEM, "test alloc failed");
    p.rows = rows; p.ncls = ncls;
    if (!heads_frags(dm, p, wc, w1, w2, ncls)) return tfail(OPD_ENOMEM, "test alloc failed");
    TCHK(opd_launch_heads(p, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(logits, p.logits, (size_t)rows * ncls * 4, hipMemcpyDeviceToHost));
    TCHK(hipMemcpy(boxes, p.boxes, (size_t)rows * 4 * 4, hipMemcpyDeviceToHost));
    return OPD_OK;
}

// postprocess_kernel alone: logits [B][Q][ncls], boxes [B][Q][4] cxcywh, orig_hw [B][2] -> records [B][Q] (compacted) + counts [B]
int opd_test_postprocess(const float* logits, const float* boxes, const int32_t* orig_hw, int B, int Q, int ncls, float threshold,
                         opd_det* records, int32_t* counts) {
    DevMem dm;
    PostParams p{};
    p.logits = dm.up(logits, (size_t)B * Q * ncls);
    p.boxes = dm.up(boxes, (size_t)B * Q * 4);
    p.orig_hw = dm.up(orig_hw, (size_t)B * 2);
    opd_det* rec = dm.up<opd_det>(nullptr, (size_t)B * Q);
    p.counts = dm.up<int32_t>(nullptr, B);
    if (!p.logits || !p.boxes || !p.orig_hw || !rec || !p.counts) return tfail(OPD_ENOMEM, "test alloc failed");
    TCHK(hipMemset(rec, 0, (size_t)B * Q * sizeof(opd_det)));
    p.records = rec; p.B = B; p.Q = Q; p.ncls = ncls; p.threshold = threshold;
    TCHK(opd_launch_postprocess(p, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(records, rec, (size_t)B * Q * sizeof(opd_det), hipMemcpyDeviceToHost));
    TCHK(hipMemcpy(counts, p.counts, (size_t)B * 4, hipMemcpyDeviceToHost));
    return OPD_OK;
}

// roi_features_kernel alone: enc [h][w][256] fp32, rois [n][4] = (x0, y0, x1, y1) in map cells -> out [n][256]
int opd_test_roi_features(const float* enc, const int32_t* rois, int n, int h, int w, float* out) {
    DevMem dm;
    const float* d_enc = dm.up(enc, (size_t)h * w * 256);
    const int32_t* d_rois = dm.up(rois, (size_t)n * 4);
    float* d_out = dm.up<float>(nullptr, (size_t)n * 256);
    if (!d_enc || !d_rois || !d_out) return tfail(OPD_ENOMEM, "test alloc failed");
    TCHK(opd_launch_roi_features(d_enc, d_rois, d_out, n, h, w, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(out, d_out, (size_t)n * 256 * 4, hipMemcpyDeviceToHost));
    return OPD_OK;
}

// host-only helpers of the loader, exposed so CPU tests can exercise them without a GPU
uint16_t opd_test_f32_to_f16(float f) { return opd::f32_to_f16(f); }
float opd_test_f16_to_f32(uint16_t h) { return opd::f16_to_f32(h); }
int opd_test_normalise_key(const char* in, char* out, int cap) {
    const std::string k = opd::normalise_key(in);
    if ((int)k.size() + 1 > cap) return OPD_EINVAL;
    memcpy(out, k.c_str(), k.size() + 1);
    return OPD_OK;
}
// parse + schema-check a checkpoint on the host (no GPU needed): returns 0 and fills depths[4], enc, dec, queries, ncls
int opd_test_inspect_checkpoint(const char* path, int32_t* info8) {
    opd::StateDict sd;
    std::string err;
    int rc = opd::load_safetensors(path, &sd, &err);
    if (rc) return tfail(rc, err);
    opd::Arch a;
    rc = opd::infer_arch(sd, &a, &err);
    if (rc) return tfail(rc, err);
    for (int i = 0; i < 4; ++i) info8[i] = a.depths[i];
    info8[4] = a.enc_layers; info8[5] = a.dec_layers; info8[6] = a.queries; info8[7] = a.ncls;
    return OPD_OK;
}


// ---- hooks that reach into a model handle (opd_model.h): fusion switches, poison allocation, graph guard, diagnostic taps ----------
// Pillow coefficient tables of the device resize (host only): bounds [out][2], coeffs [out][ksize]; returns ksize
int opd_test_resize_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* coeffs, int coeffs_capacity) {
    std::vector<int32_t> b, k;
    int ksize = 0;
    opd_resize_coeffs(in_size, out_size, &b, &k, &ksize);
    if ((int)k.size() > coeffs_capacity) return fail(OPD_EINVAL, "coefficient buffer too small");
    memcpy(bounds, b.data(), b.size() * 4);
    memcpy(coeffs, k.data(), k.size() * 4);
    return ksize;
}
// host-only pieces of the ragged-batch path, exported for the CPU tests
int opd_test_valid_prefix(int valid, int in, int out) { return valid_prefix(valid, in, out); }
int opd_test_sine_pos_embed(int h, int w, int vh, int vw, int D, float* out) {
    if (!out || h < 1 || w < 1 || vh < 1 || vw < 1 || vh > h || vw > w || D < 2 || (D & 1)) return fail(OPD_EINVAL, "bad sine_pos_embed arguments");
    std::vector<float> pos;
    sine_pos_embed(h, w, vh, vw, D, &pos);
    memcpy(out, pos.data(), pos.size() * sizeof(float));
    return OPD_OK;
}
int opd_test_set_fuse_gemm_ln(opd_detr* m, int on) {
    if (!m) return fail(OPD_EINVAL, "null model handle");
    m->fuse_gemm_ln = on ? 1 : 0;
    m->small_m_gemm = on ? 1 : 0;   // the switch covers the transformer-side specialisations
    m->deep_fc2 = on ? 1 : 0;
    m->fuse_dec0 = on ? 1 : 0;
    m->fused_dec = on ? 1 : 0;   // (the unfused chain is the cross-check of the fused decoder as well)
    if (fill_qc0(m) != OPD_OK) return OPD_EHIP;
    for (auto& g : m->graphs)  // captured graphs hold the old launch sequence
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    m->graphs.clear();
    return OPD_OK;
}
int opd_test_set_fuse_btail(opd_detr* m, int on) {   // bit 0: fused bottleneck tails, bit 1: the shortcut of stage 1 inside its first tail
    if (!m) return fail(OPD_EINVAL, "null model handle");
    m->fuse_btail = (on & 1) ? 1 : 0;
    m->fuse_shortcut = (on & 2) ? 1 : 0;
    for (auto& g : m->graphs)  // captured graphs hold the old launch sequence
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    m->graphs.clear();
    return OPD_OK;
}
int opd_test_set_pos_shadow(opd_detr* m, int on) {   // 0: row-periodic bias tables W.pos + b (round-1 form) instead of the fp16(x + pos) shadow
    if (!m) return fail(OPD_EINVAL, "null model handle");
    m->pos_shadow = on ? 1 : 0;
    for (auto& g : m->graphs)  // captured graphs hold the old launch sequence
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    m->graphs.clear();
    return OPD_OK;
}
int opd_test_set_fuse_stem_pool(opd_detr* m, int on) {
    if (!m) return fail(OPD_EINVAL, "null model handle");
    m->fuse_stem_pool = (on & 1) ? 1 : 0;   // bit 0: stem + pool in one kernel; bit 1: pre-processing inside it as well
    m->fuse_prep = (on & 2) ? 1 : 0;
    for (auto& g : m->graphs)  // captured graphs hold the old launch sequence
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    m->graphs.clear();
    return OPD_OK;
}

int opd_test_set_alloc_poison(int byte) {   // -1: off; 0 .. 255: fill byte for the buffers and red zones of handles created from now on
    g_alloc_poison = byte < 0 ? -1 : (byte & 255);
    return OPD_OK;
}
// Scans the red zones of a poison-mode handle (its own buffers and its weight set's): returns the number of buffers with a damaged
// zone (0 = intact) and describes the first one in opd_last_error().
int opd_test_check_redzones(opd_detr* m) {
    ApiScope api_scope;
    if (!m) return fail(OPD_EINVAL, "null model handle");
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipDeviceSynchronize());
    std::vector<unsigned char> h(OPD_REDZONE);
    int bad = 0;
    std::string first;
    auto scan = [&](const std::vector<RedZoned>& v, const char* what) -> int {
        for (size_t i = 0; i < v.size(); ++i)
            for (int side = 0; side < 2; ++side) {
                const char* z = static_cast<const char*>(v[i].base) + (side ? OPD_REDZONE + v[i].bytes : 0);
                HIPCHK(hipMemcpy(h.data(), z, OPD_REDZONE, hipMemcpyDeviceToHost));
                size_t lo = OPD_REDZONE, hi = 0;
                for (size_t k = 0; k < OPD_REDZONE; ++k)
                    if (h[k] != (unsigned char)v[i].poison) { lo = std::min(lo, k); hi = k; }
                if (lo <= hi) {
                    if (!bad++) first = std::string(what) + " buffer #" + std::to_string(i) + " (" + std::to_string(v[i].bytes) + " bytes): " +
                                        (side ? "zone BEHIND it" : "zone IN FRONT of it") + " overwritten at zone offsets " + std::to_string(lo) + " .. " + std::to_string(hi);
                }
            }
        return OPD_OK;
    };
    RCCHK(scan(m->zoned, "handle"));
    if (m->weights) RCCHK(scan(m->weights->zoned, "weight-set"));
    if (bad) g_err = first;
    return bad;
}
int opd_test_set_graph_guard(int on) {
    g_graph_guard = on ? 1 : 0;
    return OPD_OK;
}
// Diagnostic taps: after every launch of the forward a checksum launch of that launch's output (captured into the graph with it).
int opd_test_set_taps(opd_detr* m, int on) {
    ApiScope api_scope;
    if (!m) return fail(OPD_EINVAL, "null model handle");
    HIPCHK(hipSetDevice(m->device));
    if (on && !m->d_taps) RCCHK(dalloc(m, &m->d_taps, (size_t)OPD_MAX_TAPS * OPD_TAP_BLOCKS, false));
    m->taps = on ? 1 : 0;
    for (auto& g : m->graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    m->graphs.clear();
    return OPD_OK;
}
// sums[i] = checksum of tap i of the last forward, names = '\n'-joined tap names; returns the number of taps
int opd_test_read_taps(opd_detr* m, unsigned long long* sums, int cap, char* names, int names_cap) {
    ApiScope api_scope;
    if (!m || !sums || !m->d_taps) return fail(OPD_EINVAL, "opd_test_read_taps: taps are not enabled");
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipStreamSynchronize(m->stream));
    const int n = std::min(cap, (int)m->tap_names.size());
    std::vector<unsigned long long> h((size_t)n * OPD_TAP_BLOCKS);
    if (n) HIPCHK(hipMemcpy(h.data(), m->d_taps, h.size() * 8, hipMemcpyDeviceToHost));
    std::string all;
    for (int i = 0; i < n; ++i) {
        unsigned long long s = 0;
        for (int j = 0; j < OPD_TAP_BLOCKS; ++j) s += h[(size_t)i * OPD_TAP_BLOCKS + j];
        sums[i] = s;
        all += m->tap_names[i];
        all += '\n';
    }
    if (names && names_cap > 0) { strncpy(names, all.c_str(), (size_t)names_cap - 1); names[names_cap - 1] = 0; }
    return n;
}




// ---- fused decoder kernels (kernels_dec.hip), one launch each; weights arrive as fp32 and are split here like the loader does ----------
static const f16_t* up_frag(DevMem& dm, const float* w, int N, int K) {
    std::vector<f16_t> f((size_t)N * K * 2);
    opd_split_f16_frag(w, N, K, f.data());
    return dm.up(f.data(), f.size());
}
// h_out / q16 [M][256]; k16 / vT [M / Q][8][8][512] in fragment order (host; returned as the device wrote them: padding keys untouched = zero-filled here)
int opd_test_dec_qkv(const float* h_in, const float* partials, int nsplit, const float* b2, const float* ln_g, const float* ln_b, const float* w,
                     const float* bias, int M, int Q, float* h_out, uint16_t* q16, uint16_t* k16, uint16_t* vT) {
    DevMem dm;
    DecQkvParams p{};
    const size_t n = (size_t)M * 256, nv = (size_t)(M / Q) * 8 * 8 * 512;
    if (partials) {
        p.h_in = dm.up(h_in, n); p.partials = dm.up(partials, n * nsplit); p.nsplit = nsplit; p.b2 = dm.up(b2, 256); p.ln_g = dm.up(ln_g, 256); p.ln_b = dm.up(ln_b, 256);
        p.h_out = dm.up<float>(nullptr, n);
        if (!p.h_in || !p.partials || !p.b2 || !p.ln_g || !p.ln_b) return tfail(OPD_ENOMEM, "test alloc failed");
    } else {
        p.h_out = dm.up(const_cast<const float*>(h_in), n);
    }
    p.w = up_frag(dm, w, 768, 256);
    if (!p.w) return tfail(OPD_ENOMEM, "test alloc failed");
    p.bias = dm.up(bias, (size_t)Q * 768);
    p.q16 = dm.up<uint16_t>(nullptr, n); p.k16 = dm.up<uint16_t>(nullptr, nv); p.vT = dm.up<uint16_t>(nullptr, nv);
    if (!p.h_out || !p.bias || !p.q16 || !p.k16 || !p.vT) return tfail(OPD_ENOMEM, "test alloc failed");
    TCHK(hipMemset(p.vT, 0, nv * 2));
    TCHK(hipMemset(p.k16, 0, nv * 2));
    p.M = M; p.Q = Q;
    TCHK(opd_launch_dec_qkv(p, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(h_out, p.h_out, n * 4, hipMemcpyDeviceToHost));
    TCHK(hipMemcpy(q16, p.q16, n * 2, hipMemcpyDeviceToHost));
    TCHK(hipMemcpy(k16, p.k16, nv * 2, hipMemcpyDeviceToHost));
    TCHK(hipMemcpy(vT, p.vT, nv * 2, hipMemcpyDeviceToHost));
    return OPD_OK;
}
// h [B * Q][256] in / out, qc16 [B * Q][256] out
int opd_test_dec_self(const uint16_t* q16, const uint16_t* k16, const uint16_t* vT, float* h, const float* wo, const float* bo, const float* ln_g,
                      const float* ln_b, const float* wq, const float* rbq, int B, int Q, float scale, uint16_t* qc16) {
    DevMem dm;
    DecSelfParams p{};
    const size_t n = (size_t)B * Q * 256, nv = (size_t)B * 8 * 8 * 512;
    p.q16 = dm.up(q16, n); p.k16 = dm.up(k16, nv); p.vT = dm.up(vT, nv); p.h = dm.up(const_cast<const float*>(h), n);
    p.bo = dm.up(bo, 256); p.ln_g = dm.up(ln_g, 256); p.ln_b = dm.up(ln_b, 256); p.rbq = dm.up(rbq, (size_t)Q * 256);
    p.qc16 = dm.up<uint16_t>(nullptr, n);
    p.wo = up_frag(dm, wo, 256, 256); p.wq = up_frag(dm, wq, 256, 256);
    if (!p.wo || !p.wq) return tfail(OPD_ENOMEM, "test alloc failed");
    if (!p.q16 || !p.k16 || !p.vT || !p.h || !p.bo || !p.ln_g || !p.ln_b || !p.rbq || !p.qc16) return tfail(OPD_ENOMEM, "test alloc failed");
    p.B = B; p.Q = Q; p.scale = scale;
    TCHK(opd_launch_dec_self(p, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(h, p.h, n * 4, hipMemcpyDeviceToHost));
    TCHK(hipMemcpy(qc16, p.qc16, n * 2, hipMemcpyDeviceToHost));
    return OPD_OK;
}
// key-split attention partials: part_o [splits][B * Lq][heads * 32], part_ml [splits][B * Lq][heads][2]
int opd_test_attention_split(const uint16_t* q, const uint16_t* k, const uint16_t* v, int B, int heads, int Lq, int Lk, float scale, int splits,
                             const int32_t* key_valid, int key_row, float* part_o, float* part_ml) {
    DevMem dm;
    const int D = heads * 32;
    AttnParams p{}; p.dtype = g_test_dtype;
    p.q = dm.up(q, (size_t)B * Lq * D); p.k = dm.up(k, (size_t)B * Lk * D); p.v = dm.up(v, (size_t)B * Lk * D);
    p.key_valid = key_valid ? dm.up(key_valid, (size_t)B * 2) : nullptr;
    const size_t no = (size_t)splits * B * Lq * D, nm = (size_t)splits * B * Lq * heads * 2;
    p.part_o = dm.up<float>(nullptr, no); p.part_ml = dm.up<float>(nullptr, nm);
    if (!p.q || !p.k || !p.v || !p.part_o || !p.part_ml || (key_valid && !p.key_valid)) return tfail(OPD_ENOMEM, "test alloc failed");
    p.B = B; p.heads = heads; p.Lq = Lq; p.Lk = Lk; p.ldq = p.ldk = p.ldv = p.ldo = D; p.scale = scale; p.key_row = key_row; p.splits = splits;
    TCHK(opd_launch_attention(p, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(part_o, p.part_o, no * 4, hipMemcpyDeviceToHost));
    TCHK(hipMemcpy(part_ml, p.part_ml, nm * 4, hipMemcpyDeviceToHost));
    return OPD_OK;
}
int opd_test_dec_cross_out(const float* part_o, const float* part_ml, int splits, const float* res, int res_period, const float* wo, const float* bo,
                           const float* ln_g, const float* ln_b, int M, float* h) {
    DevMem dm;
    DecCrossOutParams p{};
    const size_t n = (size_t)M * 256;
    p.part_o = dm.up(part_o, n * splits); p.part_ml = dm.up(part_ml, (size_t)splits * M * 16); p.splits = splits;
    p.res = dm.up(res, res_period > 0 ? (size_t)res_period * 256 : n); p.res_period = res_period;
    p.h = dm.up<float>(nullptr, n); p.bo = dm.up(bo, 256); p.ln_g = dm.up(ln_g, 256); p.ln_b = dm.up(ln_b, 256); p.M = M;
    p.wo = up_frag(dm, wo, 256, 256);
    if (!p.wo) return tfail(OPD_ENOMEM, "test alloc failed");
    if (!p.part_o || !p.part_ml || !p.res || !p.h || !p.bo || !p.ln_g || !p.ln_b) return tfail(OPD_ENOMEM, "test alloc failed");
    TCHK(opd_launch_dec_cross_out(p, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(h, p.h, n * 4, hipMemcpyDeviceToHost));
    return OPD_OK;
}
// partials [F / 128][M][256]
int opd_test_dec_ffn(const float* h, const float* w1, const float* b1, const float* w2, int M, int F, float* partials) {
    DevMem dm;
    DecFfnParams p{};
    const size_t n = (size_t)M * 256, np = n * (F / OPD_DEC_FFN_CHUNK);
    p.h = dm.up(h, n); p.b1 = dm.up(b1, (size_t)F); p.partials = dm.up<float>(nullptr, np); p.M = M; p.F = F;
    p.w1 = up_frag(dm, w1, F, 256); p.w2 = up_frag(dm, w2, 256, F);
    if (!p.w1 || !p.w2) return tfail(OPD_ENOMEM, "test alloc failed");
    if (!p.h || !p.b1 || !p.partials) return tfail(OPD_ENOMEM, "test alloc failed");
    TCHK(opd_launch_dec_ffn(p, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(partials, p.partials, np * 4, hipMemcpyDeviceToHost));
    return OPD_OK;
}
// isolated timing of the fused decoder's kernels on zero-filled operands at M = B x Q rows, Lk keys: us per launch of
// [qkv, self, cross-split, cross-out, ffn] (tools/bench_dec.py)
int opd_test_bench_dec(int B, int Q, int Lk, int F, int splits, int iters, float* us5) {
    DevMem dm;
    const int M = B * Q;
    const size_t n = (size_t)M * 256;
    auto z16 = [&](size_t c) { uint16_t* p = dm.up<uint16_t>(nullptr, c); if (p) (void)hipMemset(p, 0, c * 2); return p; };
    auto z32 = [&](size_t c) { float* p = dm.up<float>(nullptr, c); if (p) (void)hipMemset(p, 0, c * 4); return p; };
    DecQkvParams a{}; DecSelfParams b{}; AttnParams c{}; DecCrossOutParams d{}; DecFfnParams e{};
    const int nchunk = F / OPD_DEC_FFN_CHUNK;
    float *h0 = z32(n), *h1 = z32(n), *part = z32(n * nchunk), *vec = z32(4096), *tabs = z32((size_t)Q * 768), *po = z32(n * splits), *pml = z32((size_t)splits * M * 16);
    uint16_t *q16 = z16(n), *k16 = z16((size_t)B * 8 * 8 * 512), *vT = z16((size_t)B * 8 * 8 * 512), *qc = z16(n), *w768 = z16(2 * 768 * 256), *w256 = z16(2 * 65536), *wf = z16((size_t)2 * F * 256),
             *mem = z16((size_t)B * Lk * 512);
    if (!h0 || !h1 || !part || !vec || !tabs || !po || !pml || !q16 || !k16 || !vT || !qc || !w768 || !w256 || !wf || !mem) return tfail(OPD_ENOMEM, "test alloc failed");
    a.h_in = h0; a.partials = part; a.nsplit = nchunk; a.b2 = vec; a.ln_g = vec; a.ln_b = vec; a.h_out = h1; a.w = w768; a.bias = tabs;
    a.q16 = q16; a.k16 = k16; a.vT = vT; a.M = M; a.Q = Q;
    b.q16 = q16; b.k16 = k16; b.vT = vT; b.h = h1; b.wo = w256; b.bo = vec; b.ln_g = vec; b.ln_b = vec; b.wq = w256;
    b.rbq = tabs; b.qc16 = qc; b.B = B; b.Q = Q; b.scale = 0.17677669f;
    c.q = qc; c.k = mem; c.v = mem + 256; c.B = B; c.heads = 8; c.Lq = Q; c.Lk = Lk; c.ldq = 256; c.ldk = c.ldv = 512; c.ldo = 256; c.scale = 0.17677669f;
    c.splits = splits; c.part_o = po; c.part_ml = pml;
    d.part_o = po; d.part_ml = pml; d.splits = splits; d.res = h1; d.h = h1; d.wo = w256; d.bo = vec; d.ln_g = vec; d.ln_b = vec; d.M = M;
    e.h = h1; e.w1 = wf; e.b1 = vec; e.w2 = wf; e.partials = part; e.M = M; e.F = F;
    hipEvent_t e0, e1;
    TCHK(hipEventCreate(&e0)); TCHK(hipEventCreate(&e1));
    for (int k = 0; k < 5; ++k) {
        auto run = [&]() -> hipError_t {
            switch (k) {
                case 0: return opd_launch_dec_qkv(a, nullptr);
                case 1: return opd_launch_dec_self(b, nullptr);
                case 2: return opd_launch_attention(c, nullptr);
                case 3: return opd_launch_dec_cross_out(d, nullptr);
                default: return opd_launch_dec_ffn(e, nullptr);
            }
        };
        for (int i = 0; i < 3; ++i) TCHK(run());
        TCHK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) TCHK(run());
        TCHK(hipEventRecord(e1, nullptr));
        TCHK(hipEventSynchronize(e1));
        float ms = 0.f;
        TCHK(hipEventElapsedTime(&ms, e0, e1));
        us5[k] = ms * 1000.f / iters;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return OPD_OK;
}
// phase stamps of dec_self_kernel (wave 0 of every workgroup, shader clocks): trace_out [ceil(Q / 16) * B][8]
int opd_test_trace_dec_self(int B, int Q, unsigned long long* trace_out) {
    DevMem dm;
    const size_t n = (size_t)B * Q * 256;
    auto z16 = [&](size_t c) { uint16_t* p = dm.up<uint16_t>(nullptr, c); if (p) (void)hipMemset(p, 0, c * 2); return p; };
    auto z32 = [&](size_t c) { float* p = dm.up<float>(nullptr, c); if (p) (void)hipMemset(p, 0, c * 4); return p; };
    DecSelfParams b{};
    const int wgs = ((Q + 15) / 16) * B;
    b.q16 = z16(n); b.k16 = z16((size_t)B * 8 * 8 * 512); b.vT = z16((size_t)B * 8 * 8 * 512); b.h = z32(n); b.wo = z16(2 * 65536); b.wq = z16(2 * 65536);
    float* vec = z32(4096);
    b.bo = vec; b.ln_g = vec; b.ln_b = vec; b.rbq = z32((size_t)Q * 256); b.qc16 = z16(n); b.B = B; b.Q = Q; b.scale = 0.17677669f;
    unsigned long long* tr = dm.up<unsigned long long>(nullptr, (size_t)wgs * 8);
    if (!b.q16 || !b.k16 || !b.vT || !b.h || !b.wo || !b.wq || !vec || !b.rbq || !b.qc16 || !tr) return tfail(OPD_ENOMEM, "test alloc failed");
    for (int i = 0; i < 3; ++i) TCHK(opd_launch_dec_self(b, nullptr));   // warm: code and weights in the caches
    b.trace = tr;
    TCHK(opd_launch_dec_self(b, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(trace_out, tr, (size_t)wgs * 8 * 8, hipMemcpyDeviceToHost));
    return OPD_OK;
}
// heads_kernel with the fused decoder's prologue: rows = LN3(hs + b2f + sum partials), then the final LayerNorm, then the heads
int opd_test_heads_fused(const float* hs, const float* partials, int nsplit, const float* b2f, const float* ln3_g, const float* ln3_b, const float* ln_g,
                         const float* ln_b, const float* wc, const float* bc, const float* w1, const float* b1, const float* w2, const float* b2,
                         const float* w3, const float* b3, int rows, int ncls, float* logits, float* boxes) {
    DevMem dm;
    auto tr = [&](const float* w, int O, int I) -> const float* {
        std::vector<float> t((size_t)O * I);
        for (int o = 0; o < O; ++o)
            for (int i = 0; i < I; ++i) t[(size_t)i * O + o] = w[(size_t)o * I + i];
        return dm.up(t.data(), t.size());
    };
    HeadParams p{};
    p.hs = dm.up(hs, (size_t)rows * 256);
    p.partials = dm.up(partials, (size_t)nsplit * rows * 256); p.nsplit = nsplit; p.ffn_b2 = dm.up(b2f, 256);
    p.ln3_gamma = dm.up(ln3_g, 256); p.ln3_beta = dm.up(ln3_b, 256);
    p.ln_gamma = dm.up(ln_g, 256); p.ln_beta = dm.up(ln_b, 256);
    p.wc = tr(wc, ncls, 256); p.bc = dm.up(bc, ncls);
    p.w1 = tr(w1, 256, 256); p.b1 = dm.up(b1, 256);
    p.w2 = tr(w2, 256, 256); p.b2 = dm.up(b2, 256);
    p.w3 = tr(w3, 4, 256); p.b3 = dm.up(b3, 4);
    p.logits = dm.up<float>(nullptr, (size_t)rows * ncls);
    p.boxes = dm.up<float>(nullptr, (size_t)rows * 4);
    if (!p.hs || !p.partials || !p.ffn_b2 || !p.ln3_gamma || !p.ln3_beta || !p.ln_gamma || !p.ln_beta || !p.wc || !p.bc || !p.w1 || !p.b1 || !p.w2 || !p.b2 ||
        !p.w3 || !p.b3 || !p.logits || !p.boxes)
        return tfail(OPD_ENOMEM, "test alloc failed");
    p.rows = rows; p.ncls = ncls;
    if (!heads_frags(dm, p, wc, w1, w2, ncls)) return tfail(OPD_ENOMEM, "test alloc failed");
    TCHK(opd_launch_heads(p, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(logits, p.logits, (size_t)rows * ncls * 4, hipMemcpyDeviceToHost));
    TCHK(hipMemcpy(boxes, p.boxes, (size_t)rows * 4 * 4, hipMemcpyDeviceToHost));
    return OPD_OK;
}
int opd_test_set_fused_dec(opd_detr* m, int on) {   // 0: the unfused decoder chain (single fp16 operands, nine launches per layer)
    ApiScope api_scope;
    if (!m) return fail(OPD_EINVAL, "null model handle");
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipStreamSynchronize(m->stream));
    m->fused_dec = on ? 1 : 0;
    RCCHK(fill_qc0(m));
    for (auto& g : m->graphs)  // captured graphs hold the old launch sequence
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    m->graphs.clear();
    return OPD_OK;
}
// host-only: the error-diffusion rounding of the weight loader (opd_host.h), in place on [rows][taps][cin]
int opd_test_round_f16_diffused(float* w, int rows, int taps, int cin) {
    if (!w || rows < 0 || taps < 1 || cin < 1) return tfail(OPD_EINVAL, "opd_test_round_f16_diffused: bad arguments");
    opd::round_f16_diffused(w, (size_t)rows, taps, cin);
    return OPD_OK;
}

}  // extern "C"
#pragma GCC visibility pop
