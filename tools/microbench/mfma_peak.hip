// Micro-benchmark: what the matrix pipes of THIS device sustain on random fp16 operands, and at what clock.
// The dense fp16 / bf16 MFMA figure bench.py prices against (2.5 PFLOP/s) is 1024 FLOP / clk / SIMD at 2.4 GHz; under an MFMA-dense load
// the chip holds a lower clock (MI355X_MICROARCH.md "DVFS give-back"), so "fraction of 2.5 PF" has a ceiling well under 1 that no k-loop
// can exceed.  This program measures that ceiling: back-to-back v_mfma_f32_16x16x32_f16 from registers (no LDS, no global memory),
//   (a) operands all zero, (b) small smooth positive operands, (c) full-range random operands (sign and exponent bits random),
// each for ~0.5 s of back-to-back launches, one and two waves per SIMD, with the in-kernel clock from s_memtime / s_memrealtime.
//     hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// 16 independent accumulators (the implicit-GEMM kernels' 4 x 4 wave tile), 4 A and 4 B fragments: 32 MFMAs per iteration
__global__ __launch_bounds__(256) void mfma_regs(const half8* __restrict__ ops, float* out, unsigned long long* clk, int iters) {
    float4v acc[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = float4v{0, 0, 0, 0};
    half8 x[4], w[4];
    for (int j = 0; j < 4; ++j) { x[j] = ops[(j * 256 + threadIdx.x)]; w[j] = ops[((4 + j) * 256 + threadIdx.x)]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)   // (inline asm: hipcc's own schedule of this loop shuffles the accumulators through v_accvgpr moves)
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[nt][mt]) : "v"(w[nt]), "v"(x[mt]));
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // (the last MFMAs' results: inline asm has no hazard tracking)
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; __builtin_memcpy(&u, &h, 2); return u; }

int main() {
    const int iters = 20000;   // 640 k MFMAs per wave and launch
    float* out; CHK(hipMalloc(&out, 64));
    unsigned long long* clk; CHK(hipMalloc(&clk, 2 * 1024 * sizeof(unsigned long long)));
    half8* ops; CHK(hipMalloc(&ops, 8 * 256 * sizeof(half8)));
    std::vector<unsigned short> h(8 * 256 * 8);
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    const char* names[3] = {"zeros", "smooth positive (0.001 .. 2)", "random full range (|x| in [2^-3, 2), random sign)"};
    for (int kind = 0; kind < 3; ++kind) {
        srand(1234);
        for (size_t i = 0; i < h.size(); ++i) {
            if (kind == 0) h[i] = 0;
            else if (kind == 1) h[i] = f2h(0.001f * (float)(i % 2000 + 1));
            else { const float m = 1.0f + (rand() % 1024) / 1024.0f; const int e = rand() % 4 - 3; h[i] = f2h((rand() & 1 ? -1.f : 1.f) * m * (float)(1 << (e + 3)) / 8.0f); }
        }
        CHK(hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        for (int bpc : {1, 2}) {
            const int blocks = 256 * bpc;
            const double flops = (double)blocks * 4 * iters * 32 * (2.0 * 16 * 16 * 32);
            // ~0.5 s of back-to-back launches first (the clock settles), then time 8 launches
            for (int rep = 0; rep < 40; ++rep) mfma_regs<<<blocks, 256>>>(ops, out, clk, iters);
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(a));
            const int reps = 8;
            for (int rep = 0; rep < reps; ++rep) mfma_regs<<<blocks, 256>>>(ops, out, clk, iters);
            CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
            float ms; CHK(hipEventElapsedTime(&ms, a, b));
            std::vector<unsigned long long> c(2 * blocks);
            CHK(hipMemcpy(c.data(), clk, c.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> ghz, cyc;
            for (int i = 0; i < blocks; ++i) { ghz.push_back((double)c[2 * i] / (double)c[2 * i + 1] * 0.1); cyc.push_back((double)c[2 * i] / ((double)iters * 32)); }
            std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
            printf("%-52s %d wave(s)/SIMD: %7.1f TFLOP/s = %.3f of 2500   in-kernel clock %.3f GHz (median; min %.3f max %.3f)   %.2f clk per MFMA per wave\n",
                   names[kind], bpc, flops * reps / ms / 1e9, flops * reps / ms / 1e9 / 2500.0, ghz[blocks / 2], ghz.front(), ghz.back(), cyc[blocks / 2]);
        }
    }
    return 0;
}
