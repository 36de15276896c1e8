// Micro-benchmark: epilogue access patterns against HBM.  A [M][C] fp16 matrix (C = 256) is written / read / read+written
// by waves that own 32 rows each, 64 channels (128 B) per "chunk step", in two lane layouts:
//   paired : the register epilogue's layout — lane (g, li) -> row (g&1)*16 + li, 16 B at channel (g>>1)*8 + nt*16
//            (one instruction touches 32 rows x 32 B)
//   rows   : lane -> row l>>3, 16 B at channel (l&7)*8 (one instruction touches 8 rows x 128 B, i.e. whole lines)
// Answers: is the 32-byte pattern request-rate bound, and what does a streaming read+write mix reach at all?
// Build: hipcc --offload-arch=gfx950 -O3 stpath.hip -o stpath
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int C = 256;

// MODE bit0: read src, bit1: write dst.  LAYOUT 0 paired, 1 rows
template <int MODE, int LAYOUT>
__global__ __launch_bounds__(256) void stream(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst, int M, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int m0 = (blockIdx.x * 4 + wave) * 32;
    if (m0 >= M) return;
    unsigned acc = 0;
#pragma unroll
    for (int j = 0; j < C / 64; ++j) {
        uint4 v[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            size_t o;
            if (LAYOUT == 0) o = (size_t)(m0 + (g & 1) * 16 + li) * C + j * 64 + nt * 16 + (g >> 1) * 8;
            else o = (size_t)(m0 + nt * 8 + (lane >> 3)) * C + j * 64 + (lane & 7) * 8;
            if (MODE & 1) v[nt] = *reinterpret_cast<const uint4*>(src + o);
            else v[nt] = make_uint4(lane, j, nt, m0);
            if (MODE & 2) {
                uint4 w = v[nt];
                w.x += 1;
                *reinterpret_cast<uint4*>(dst + o) = w;
            } else {
                acc += v[nt].x ^ v[nt].w;
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int LAYOUT>
int run(const char* name, const unsigned short* s, unsigned short* d, int M, unsigned* sink) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CHK(hipEventRecord(a));
        hipLaunchKernelGGL((stream<MODE, LAYOUT>), dim3((M + 127) / 128), dim3(256), 0, 0, s, d, M, sink);
        CHK(hipEventRecord(b));
        CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double bytes = (double)M * C * 2 * (((MODE & 1) ? 1 : 0) + ((MODE & 2) ? 1 : 0));
    printf("%-28s %8.1f us  %7.1f GB/s\n", name, best * 1e3, bytes / best / 1e6);
    return 0;
}

int main() {
    const int M = 534400;
    unsigned short *s, *d; unsigned* sink;
    CHK(hipMalloc(&s, (size_t)M * C * 2)); CHK(hipMalloc(&d, (size_t)M * C * 2)); CHK(hipMalloc(&sink, 64));
    CHK(hipMemset(s, 1, (size_t)M * C * 2));
    if (run<1, 0>("read   paired(32B)", s, d, M, sink)) return 1;
    if (run<1, 1>("read   rows(128B)", s, d, M, sink)) return 1;
    if (run<2, 0>("write  paired(32B)", s, d, M, sink)) return 1;
    if (run<2, 1>("write  rows(128B)", s, d, M, sink)) return 1;
    if (run<3, 0>("copy   paired(32B)", s, d, M, sink)) return 1;
    if (run<3, 1>("copy   rows(128B)", s, d, M, sink)) return 1;
    return 0;
}
