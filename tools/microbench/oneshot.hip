// Micro-benchmark for the decoder's latency-bound kernels: W workgroups of 512 threads each pull KB KiB ONCE (everything requested up
// front, one wait) and exit.  How long does such a launch take, by load path and access shape?
//   mode 0: LDS-DMA, 1-KiB contiguous pieces          mode 1: VGPR dwordx4 loads, 1-KiB contiguous per wave-instruction
//   mode 2: VGPR loads in MFMA A-fragment shape (16 rows x 64 B, row pitch 512 B)     mode 3: empty kernel (launch floor)
//   mode 4: LDS-DMA pieces of 8 rows x 128 B at a 4-KiB row pitch (the tile-GEMM staging shape)
// region: 0 = all workgroups read the SAME KB KiB (weights), 1 = each its own.   Build: hipcc --offload-arch=gfx950 -O3 oneshot.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE, int PIECES>   // PIECES = KiB per wave
__global__ __launch_bounds__(512) void pull(const unsigned char* __restrict__ src, size_t block_stride, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* base = src + (size_t)blockIdx.x * block_stride + (size_t)wave * PIECES * 1024;
    unsigned acc = 0;
    if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < PIECES; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + j * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(smem + ((wave * PIECES + j) % 128) * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc = smem[threadIdx.x * 4];
    } else if (MODE == 4) {   // LDS-DMA, one piece = 8 rows x 128 B gathered at a 4-KiB row pitch (a [N][2048] fp16 matrix, 64-wide k-step)
        const int lrow = lane >> 3, lchunk = lane & 7;
#pragma unroll
        for (int j = 0; j < PIECES; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)blockIdx.x * block_stride + (size_t)((wave * 4 + (j & 3)) * 8 + lrow) * 4096 + (j >> 2) * 128 + lchunk * 16),
                                             (__attribute__((address_space(3))) void*)(smem + ((wave * PIECES + j) % 128) * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc = smem[threadIdx.x * 4];
    } else if (MODE == 1) {
        uint4 v[PIECES];
#pragma unroll
        for (int j = 0; j < PIECES; ++j) v[j] = *reinterpret_cast<const uint4*>(base + j * 1024 + lane * 16);
#pragma unroll
        for (int j = 0; j < PIECES; ++j) acc += v[j].x ^ v[j].w;
    } else if (MODE == 2) {
        uint4 v[PIECES];
        const int g = lane >> 4, li = lane & 15;   // piece j = (tile j / 8, k-step j % 8): rows li of the tile, 64 B at k-step offset
#pragma unroll
        for (int j = 0; j < PIECES; ++j) v[j] = *reinterpret_cast<const uint4*>(base + (size_t)((j / 8) * 16 + li) * 512 + (j % 8) * 64 + g * 16);
#pragma unroll
        for (int j = 0; j < PIECES; ++j) acc += v[j].x ^ v[j].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int PIECES>
int run(const char* name, const unsigned char* d, size_t stride, int blocks, unsigned* sink) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    const int lds = (MODE == 0 || MODE == 4) ? (8 * PIECES > 128 ? 128 : 8 * PIECES) * 1024 : 0;
    if (lds > 65536) CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(pull<MODE, PIECES>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int reps = 200;
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((pull<MODE, PIECES>), dim3(blocks), dim3(512), lds, 0, d, stride, sink);
    CHK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((pull<MODE, PIECES>), dim3(blocks), dim3(512), lds, 0, d, stride, sink);
    CHK(hipEventRecord(b));
    CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1000.0 / reps;
    printf("%-34s wgs %4d  %3d KiB/wg  %s: %6.2f us/launch\n", name, blocks, 8 * PIECES, stride ? "own   " : "shared", us);
    return 0;
}

int main() {
    unsigned char* d; unsigned* sink;
    const size_t total = (size_t)512 << 20;
    CHK(hipMalloc(&d, total)); CHK(hipMalloc(&sink, 64));
    CHK(hipMemset(d, 1, total));
    for (int blocks : {50, 200, 256}) {
        run<3, 1>("empty kernel", d, 0, blocks, sink);
        for (size_t stride : {(size_t)0, (size_t)1 << 20}) {
            run<0, 8>("LDS-DMA contiguous", d, stride, blocks, sink);
            run<0, 16>("LDS-DMA contiguous", d, stride, blocks, sink);
            run<0, 32>("LDS-DMA contiguous (128 KiB LDS ring)", d, stride, blocks, sink);
            run<4, 16>("LDS-DMA 8 rows x 128 B gather", d, stride, blocks, sink);
            run<4, 32>("LDS-DMA 8 rows x 128 B gather", d, stride, blocks, sink);
            run<1, 8>("VGPR contiguous", d, stride, blocks, sink);
            run<1, 16>("VGPR contiguous", d, stride, blocks, sink);
            run<1, 32>("VGPR contiguous", d, stride, blocks, sink);
            run<2, 8>("VGPR A-fragment shape", d, stride, blocks, sink);
            run<2, 16>("VGPR A-fragment shape", d, stride, blocks, sink);
            run<2, 32>("VGPR A-fragment shape", d, stride, blocks, sink);
        }
    }
    return 0;
}
