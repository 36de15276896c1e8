// Do vector-memory STORES retire (decrement vmcnt) in issue order with respect to older LOADS on gfx950?
// Every counted wait of the kernels ("s_waitcnt vmcnt(N)": the N youngest operations may stay in flight) relies on an answer.
// Each wave, per iteration: one LDS-DMA load from a cold, far-away address (HBM miss) into its private LDS slot, then K stores to a hot line,
// then s_waitcnt vmcnt(K) -- "everything but the K stores has returned" if retirement is in issue order -- and reads the slot.  A slot that
// still holds the previous contents (a marker) means the stores retired BEFORE the older load.
// mode 1: the K younger operations are VGPR loads from a hot line instead of stores; mode 2: LDS-DMA loads of a hot line (the same class
// as the older load: the case the wave-private rings rely on).
// Build: hipcc --offload-arch=gfx950 -O3 vmorder.hip -o vmorder
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE, int K, int W, int SLEEP>   // W: the wait's count (K: the stated claim; 0: control); SLEEP: s_sleep units between wait and read
__global__ __launch_bounds__(256) void probe(const unsigned* __restrict__ cold, size_t cold_words, unsigned* hot, unsigned long long* bad, int iters) {
    __shared__ __attribute__((aligned(1024))) unsigned slot[4][256];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long miss = 0;
    unsigned sink = 0;
    size_t pos = ((size_t)blockIdx.x * 4 + wave) * 104729u * 256u % cold_words;
    for (int it = 0; it < iters; ++it) {
        slot[wave][lane * 4] = 0xdeadbeefu;                  // marker
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pos = (pos + (size_t)7919 * 4099 * 256) % (cold_words - 256);
        pos &= ~(size_t)255;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cold + pos + lane * 4), (__attribute__((address_space(3))) void*)&slot[wave][0], 16, 0, 0);
        if (MODE == 3) {   // the OLDER operation is a cold VGPR load (issued by hand: the compiler adds no wait of its own), the K younger ones hot LDS-DMA loads
            unsigned v = 0xdeadbeefu;
            const unsigned* ptr = cold + pos + lane * 4;
            asm volatile("global_load_dword %0, %1, off" : "+v"(v) : "v"(ptr) : "memory");
#pragma unroll
            for (int k = 0; k < K; ++k)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(hot + ((blockIdx.x * 4 + wave) * 64 + lane) * 16), (__attribute__((address_space(3))) void*)&slot[(wave + 1) & 3][0], 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W) : "memory");
            unsigned got;
            asm volatile("v_mov_b32 %0, %1" : "=v"(got) : "v"(v));
            if (got == 0xdeadbeefu) ++miss;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            sink += got;
            continue;
        }
        if (MODE == 2) {   // the K younger operations are LDS-DMA loads of a HOT line into another slot
#pragma unroll
            for (int k = 0; k < K; ++k)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(hot + ((blockIdx.x * 4 + wave) * 64 + lane) * 16), (__attribute__((address_space(3))) void*)&slot[(wave + 1) & 3][0], 16, 0, 0);
        } else if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) __builtin_nontemporal_store(it + k, hot + ((blockIdx.x * 4 + wave) * 64 + lane) * 16 + k);
        } else {
            unsigned v[K];
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = __builtin_nontemporal_load(hot + ((blockIdx.x * 4 + wave) * 64 + lane) * 16 + k);
            asm volatile("" ::: "memory");
            // (no use of v before the wait: the compiler must not insert its own)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W) : "memory");
            if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
            const unsigned got = *(volatile unsigned*)&slot[wave][lane * 4];   // (volatile: the compiler does not see the DMA write and would forward the marker)
            if (got == 0xdeadbeefu) ++miss;
#pragma unroll
            for (int k = 0; k < K; ++k) sink += v[k];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            continue;
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W) : "memory");
        if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
        const unsigned got = *(volatile unsigned*)&slot[wave][lane * 4];   // (volatile: the compiler does not see the DMA write and would forward the marker)
        if (got == 0xdeadbeefu) ++miss;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (miss) atomicAdd(bad, miss);
    if (sink == 0x12345678u) hot[0] = sink;
}

int main() {
    const size_t cold_words = (size_t)1 << 30;   // 4 GiB of cold data: every DMA load misses every cache
    unsigned *cold, *hot;
    unsigned long long* bad;
    CHK(hipMalloc(&cold, cold_words * 4));
    CHK(hipMemset(cold, 0x11, cold_words * 4));   // (never equal to the marker)
    CHK(hipMalloc(&hot, (size_t)1024 * 4 * 64 * 16 * 4));
    CHK(hipMemset(hot, 0, (size_t)1024 * 4 * 64 * 16 * 4));
    CHK(hipMalloc(&bad, 8));
    auto report = [&](const char* what) -> int {
        CHK(hipDeviceSynchronize());
        unsigned long long h = 0;
        CHK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
        printf("%-86s slot still unwritten in %10llu of %llu lane-iterations\n", what, h, 1024ull * 256 * 2000);
        CHK(hipMemset(bad, 0, 8));
        return 0;
    };
    CHK(hipMemset(bad, 0, 8));
#define RUN(MODE, K, W, SL, TXT) hipLaunchKernelGGL((probe<MODE, K, W, SL>), dim3(1024), dim3(256), 0, 0, cold, cold_words, hot, bad, 2000); if (report(TXT)) return 1;
    RUN(0, 4, 0, 0, "cold LDS-DMA load, 4 younger stores, s_waitcnt vmcnt(0), read at once:");
    RUN(0, 4, 4, 0, "cold LDS-DMA load, 4 younger stores, s_waitcnt vmcnt(4), read at once:");
    RUN(0, 4, 4, 8, "cold LDS-DMA load, 4 younger stores, s_waitcnt vmcnt(4), read 512 clocks later:");
    RUN(2, 4, 4, 0, "cold LDS-DMA load, 4 younger HOT LDS-DMA loads, s_waitcnt vmcnt(4), read at once:");
    RUN(3, 4, 4, 0, "cold VGPR load, 4 younger HOT LDS-DMA loads, s_waitcnt vmcnt(4), register read at once:");
    RUN(3, 4, 0, 0, "cold VGPR load, 4 younger HOT LDS-DMA loads, s_waitcnt vmcnt(0), register read at once:");
    RUN(1, 4, 0, 0, "cold LDS-DMA load, 4 younger VGPR loads, s_waitcnt vmcnt(0), read at once:");
    RUN(1, 4, 4, 0, "cold LDS-DMA load, 4 younger VGPR loads, s_waitcnt vmcnt(4), read at once:");
    RUN(1, 4, 4, 8, "cold LDS-DMA load, 4 younger VGPR loads, s_waitcnt vmcnt(4), read 512 clocks later:");
    return 0;
}
