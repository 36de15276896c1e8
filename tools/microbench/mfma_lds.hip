// Micro-benchmark: the k-step body of the implicit-GEMM kernel (swizzled ds_read_b128 fragments + 16x16x32 f16 MFMAs)
// with operands resident in LDS and NO global loads: the compute-side ceiling of the wave structure.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

template <int BARRIER, int NT>
__global__ __launch_bounds__(256, 2) void body(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 65536 / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u ^ (i * 2654435761u & 0x03ff03ffu);
    __syncthreads();
    float4v acc[NT][4];
    for (int a = 0; a < NT; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = float4v{0, 0, 0, 0};
    const int frow = lane & 15, fchk = lane >> 4;
    for (int it = 0; it < iters; ++it) {
        const unsigned char* As = smem + (it & 1) * 32768;
        const unsigned char* Bs = As + 16384;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 xf[4], wf[NT];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) xf[mt] = *reinterpret_cast<const half8*>(As + swz(wm * 64 + mt * 16 + frow, kk * 4 + fchk));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[nt] = *reinterpret_cast<const half8*>(Bs + swz(wn * (NT * 16) + nt * 16 + frow, kk * 4 + fchk));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
        }
        if (BARRIER) __syncthreads();
    }
    float s = 0;
    for (int a = 0; a < NT; ++a) for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][3];
    if (s == 12345.f) out[0] = s;
}

// register-resident operands: pure MFMA issue rate
__global__ __launch_bounds__(256, 2) void mfma_only(float* out, int iters) {
    float4v acc[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = float4v{0, 0, 0, 0};
    half8 x, w;
    for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(0.001f * (threadIdx.x + j)); w[j] = (_Float16)(0.002f * (threadIdx.x * 3 + j)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, acc[nt][mt], 0, 0, 0);
    }
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) s += acc[a][b][0];
    if (s == 12345.f) out[0] = s;
}

template <typename F>
int timeit(const char* name, F launch, double flops) {
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        CHK(hipEventRecord(a)); launch(); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("%-56s %8.1f TFLOP/s  (%.3f ms)\n", name, flops / best / 1e9, best);
    return 0;
}

int main() {
    float* out; CHK(hipMalloc(&out, 64));
    const int iters = 4000;
    CHK(hipFuncSetAttribute((const void*)body<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHK(hipFuncSetAttribute((const void*)body<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHK(hipFuncSetAttribute((const void*)body<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int bpc : {1, 2}) {
        const int blocks = 256 * bpc;
        const double fl4 = (double)blocks * iters * 4 * 32 * 16384.0, fl2 = fl4 / 2;
        char nm[128];
        snprintf(nm, 128, "MFMA only (regs), %d blocks/CU", bpc);
        timeit(nm, [&] { hipLaunchKernelGGL(mfma_only, dim3(blocks), dim3(256), 0, 0, out, iters); }, fl4);
        snprintf(nm, 128, "LDS frags + MFMA 128x128, barrier/k-step, %d blocks/CU", bpc);
        timeit(nm, [&] { hipLaunchKernelGGL((body<1, 4>), dim3(blocks), dim3(256), 65536, 0, out, iters); }, fl4);
        snprintf(nm, 128, "LDS frags + MFMA 128x128, no barrier, %d blocks/CU", bpc);
        timeit(nm, [&] { hipLaunchKernelGGL((body<0, 4>), dim3(blocks), dim3(256), 65536, 0, out, iters); }, fl4);
        snprintf(nm, 128, "LDS frags + MFMA 128x64, barrier/k-step, %d blocks/CU", bpc);
        timeit(nm, [&] { hipLaunchKernelGGL((body<1, 2>), dim3(blocks), dim3(256), 65536, 0, out, iters); }, fl2);
    }
    return 0;
}
