// Micro-benchmark: how fast can one CU pull tile data (a) through LDS-DMA (global_load_lds_dwordx4) and (b) through
// global_load_dwordx4 into VGPRs, from an L2-resident region and from an HBM-sized stream?  Used to calibrate the
// load-path ceiling that bounds the implicit-GEMM kernel.  Build: hipcc --offload-arch=gfx950 -O3 ldpath.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE, int INFLIGHT>  // MODE 0: LDS-DMA, 1: VGPR loads.  each iteration a wave moves INFLIGHT x 1 KiB
__global__ __launch_bounds__(256) void pull(const unsigned char* __restrict__ src, size_t region_bytes, size_t block_stride,
                                            int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* base = src + (size_t)blockIdx.x * block_stride;
    size_t off = (size_t)wave * INFLIGHT * 1024;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < INFLIGHT; ++j) {
                const size_t o = (off + (size_t)j * 1024) % region_bytes;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + o + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(smem + (wave * INFLIGHT + j) * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            uint4 v[INFLIGHT];
#pragma unroll
            for (int j = 0; j < INFLIGHT; ++j) {
                const size_t o = (off + (size_t)j * 1024) % region_bytes;
                v[j] = *reinterpret_cast<const uint4*>(base + o + lane * 16);
            }
#pragma unroll
            for (int j = 0; j < INFLIGHT; ++j) acc += v[j].x ^ v[j].w;
        }
        off += 4 * INFLIGHT * 1024;
    }
    if (MODE == 0) acc = smem[threadIdx.x];
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int INFLIGHT>
int run(const char* name, const unsigned char* d, size_t region, size_t stride, int blocks, int iters, unsigned* sink) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    const int lds = 4 * INFLIGHT * 1024;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(a));
        hipLaunchKernelGGL((pull<MODE, INFLIGHT>), dim3(blocks), dim3(256), lds, 0, d, region, stride, iters, sink);
        CHK(hipEventRecord(b));
        CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b));
        const double bytes = (double)blocks * iters * 4 * INFLIGHT * 1024;
        if (rep == 2) printf("%-44s blocks %5d inflight/wave %2d KiB: %8.1f GB/s  (%.1f B/clk/CU @2.4GHz)\n", name, blocks, INFLIGHT,
                             bytes / ms / 1e6, bytes / ms / 1e6 / 256 / 2.4);
    }
    return 0;
}

int main() {
    const size_t total = (size_t)4 << 30;
    unsigned char* d; unsigned* sink;
    CHK(hipMalloc(&d, total)); CHK(hipMalloc(&sink, 64));
    CHK(hipMemset(d, 1, total));
    for (int bpc : {1, 2, 4}) {
        const int blocks = 256 * bpc;
        // (a) every block loops over its own 64 KiB (L2 / L1 resident after the first pass)
        run<0, 8>("LDS-DMA, own 64 KiB region (L2 hit)", d, 64 << 10, 1 << 20, blocks, 2000, sink);
        run<1, 8>("VGPR loads, own 64 KiB region (L2 hit)", d, 64 << 10, 1 << 20, blocks, 2000, sink);
        // (b) all blocks share one 2 MiB region (L2 resident per XCD)
        run<0, 8>("LDS-DMA, shared 2 MiB region", d, 2 << 20, 0, blocks, 2000, sink);
        run<1, 8>("VGPR loads, shared 2 MiB region", d, 2 << 20, 0, blocks, 2000, sink);
        // (c) stream: each block walks its own 4 MiB of a 4 GiB buffer once
        run<0, 8>("LDS-DMA, HBM stream", d, (size_t)4 << 20, (size_t)4 << 20, blocks, 128, sink);
        run<1, 8>("VGPR loads, HBM stream", d, (size_t)4 << 20, (size_t)4 << 20, blocks, 128, sink);
        run<0, 16>("LDS-DMA, HBM stream", d, (size_t)4 << 20, (size_t)4 << 20, blocks, 64, sink);
        run<1, 16>("VGPR loads, HBM stream", d, (size_t)4 << 20, (size_t)4 << 20, blocks, 64, sink);
    }
    return 0;
}
