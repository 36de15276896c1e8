// Micro-benchmark: does a tensor written by one kernel come back from cache (L2 4 MiB per XCD, 256 MiB Infinity Cache)
// when the NEXT kernel reads it, and does the order in which the consumer walks it matter?  Kernel W streams a buffer out
// in launch order; kernel R reads it either in the same order (oldest lines first: LRU worst case once the buffer is
// larger than the cache) or in reverse (newest first), keeping every chunk on the XCD that wrote it.
// Build: hipcc --offload-arch=gfx950 -O3 reuse.hip -o reuse
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int CHUNK = 64 * 1024;  // bytes per workgroup

__device__ __forceinline__ size_t chunk_of(int b, int n, int reverse) {
    if (!reverse) return (size_t)b;
    const int x = b & 7, s = b >> 3;
    return (size_t)(((n >> 3) - 1 - s) * 8 + x);   // same XCD (b % 8), opposite end of the buffer
}

__global__ __launch_bounds__(256) void wr(uint4* dst, int n, unsigned seed) {
    uint4* p = dst + chunk_of(blockIdx.x, n, 0) * (CHUNK / 16);
#pragma unroll
    for (int i = 0; i < CHUNK / 16 / 256; ++i) p[i * 256 + threadIdx.x] = make_uint4(seed, i, threadIdx.x, blockIdx.x);
}

__global__ __launch_bounds__(256) void rd(const uint4* src, int n, int reverse, unsigned* sink) {
    const uint4* p = src + chunk_of(blockIdx.x, n, reverse) * (CHUNK / 16);
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < CHUNK / 16 / 256; ++i) {
        const uint4 v = p[i * 256 + threadIdx.x];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

// read + write (a consumer that also produces the next tensor): y = f(x), both the same size
__global__ __launch_bounds__(256) void rw(const uint4* src, uint4* dst, int n, int reverse) {
    const size_t c = chunk_of(blockIdx.x, n, reverse);
    const uint4* p = src + c * (CHUNK / 16);
    uint4* q = dst + c * (CHUNK / 16);
#pragma unroll
    for (int i = 0; i < CHUNK / 16 / 256; ++i) {
        uint4 v = p[i * 256 + threadIdx.x];
        v.x += 1;
        q[i * 256 + threadIdx.x] = v;
    }
}

int main() {
    unsigned* sink;
    CHK(hipMalloc(&sink, 4));
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    const int sizes_mb[] = {32, 64, 128, 192, 256, 273, 384, 546};
    printf("%8s %12s %12s %12s | %14s %14s\n", "MiB", "write GB/s", "read fwd", "read rev", "chain fwd GB/s", "chain zigzag");
    for (int mb : sizes_mb) {
        const int n = (int)(((size_t)mb << 20) / CHUNK) & ~7;
        const size_t bytes = (size_t)n * CHUNK;
        uint4 *x, *y;
        CHK(hipMalloc(&x, bytes)); CHK(hipMalloc(&y, bytes));
        float tw = 0, tf = 0, tr = 0;
        const int reps = 10;
        for (int rev = 0; rev < 2; ++rev) {
            float acc_w = 0, acc_r = 0;
            for (int it = 0; it < reps + 2; ++it) {
                CHK(hipEventRecord(a));
                hipLaunchKernelGGL(wr, dim3(n), dim3(256), 0, 0, x, n, (unsigned)it);
                CHK(hipEventRecord(b));
                CHK(hipEventSynchronize(b));
                float ms; CHK(hipEventElapsedTime(&ms, a, b));
                if (it >= 2) acc_w += ms;
                CHK(hipEventRecord(a));
                hipLaunchKernelGGL(rd, dim3(n), dim3(256), 0, 0, x, n, rev, sink);
                CHK(hipEventRecord(b));
                CHK(hipEventSynchronize(b));
                CHK(hipEventElapsedTime(&ms, a, b));
                if (it >= 2) acc_r += ms;
            }
            tw = acc_w / reps;
            (rev ? tr : tf) = acc_r / reps;
        }
        // chain of read+write kernels x -> y -> x -> ... : every launch in the same direction, or alternating directions
        float tc[2];
        for (int zig = 0; zig < 2; ++zig) {
            const int L = 8;
            for (int it = 0; it < 2; ++it) {
                CHK(hipEventRecord(a));
                for (int l = 0; l < L; ++l)
                    hipLaunchKernelGGL(rw, dim3(n), dim3(256), 0, 0, (l & 1) ? y : x, (l & 1) ? x : y, n, zig ? (l & 1) : 0);
                CHK(hipEventRecord(b));
                CHK(hipEventSynchronize(b));
                float ms; CHK(hipEventElapsedTime(&ms, a, b));
                tc[zig] = ms / L;
            }
        }
        const double gb = bytes / 1e9;
        printf("%8d %12.0f %12.0f %12.0f | %14.0f %14.0f\n", mb, gb / (tw * 1e-3), gb / (tf * 1e-3), gb / (tr * 1e-3),
               2 * gb / (tc[0] * 1e-3), 2 * gb / (tc[1] * 1e-3));
        fflush(stdout);
        CHK(hipFree(x)); CHK(hipFree(y));
    }
    return 0;
}
