// Microbenchmark (diagnostic, not part of the library): streaming-read ceiling of LDS-DMA (global_load_lds_dwordx4) for the
// attention kernel's access pattern.  One 512-thread workgroup per CU walks "heads"; per head it pulls 84 KiB into LDS
// (one batch in flight while the previous one is waited for, like attn_heads_kernel), nothing is computed.
//   pattern 0: every 1 KiB piece is contiguous in memory (head-major layout)
//   pattern 1: a piece = 8 rows x 128 B at a 4608-byte row stride (packed qkv of 12 heads, what the kernel reads today)
// Build: hipcc --offload-arch=gfx950 -O3 load_probe.hip -o load_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int P, int DEPTH>
__global__ __launch_bounds__(512) void pull(const char* src, int heads_total, int64_t head_bytes, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x 88 KiB would not fit: DEPTH buffers of 44 KiB, 2 passes per head
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int it = 0;
    for (int hd = blockIdx.x; hd < heads_total; hd += gridDim.x, ++it) {
        // 84 pieces of 1 KiB per head: wave w takes pieces w, w+8, ... (10 or 11 per wave)
        for (int pc = wave; pc < 84; pc += 8) {
            const char* p;
            if (P == 0) {
                p = src + (int64_t)hd * (84 * 1024) + pc * 1024 + lane * 16;
            } else {   // packed qkv [B][197][q | k | v: 3 x 12 heads x 128 B]: 28 pieces of 8 rows for each of q, k, v
                const int b = hd / 12, h = hd - b * 12;
                int row = (pc % 28) * 8 + (lane >> 3);
                row = row < 197 ? row : 196;
                p = src + (int64_t)b * (197 * 4608) + (int64_t)row * 4608 + (pc / 28) * 1536 + h * 128 + (lane & 7) * 16;
            }
            __builtin_amdgcn_global_load_lds((const void*)p, LDS_PTR(smem + ((it % DEPTH) * 84 + pc) * 1024 % (150 * 1024)), 16, 0, 0);
        }
        if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(11)" ::: "memory");   // the previous head's pieces have landed, this head's stay in flight
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && smem[17] == 123) sink[0] = 1;
}

template <int P, int DEPTH>
static float run(const char* src, int heads, int64_t hb, int* sink, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void*)pull<P, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((pull<P, DEPTH>), dim3(256), dim3(512), 160 * 1024, 0, src, heads, hb, sink);
    hipEventRecord(a, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((pull<P, DEPTH>), dim3(256), dim3(512), 160 * 1024, 0, src, heads, hb, sink);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const int heads = 24576;    // 8 x the B=256 layer: 2.0 GB per launch, far beyond L2 + Infinity Cache
    char* src = nullptr;
    int* sink = nullptr;
    const size_t max0 = (size_t)heads * 84 * 1024;                   // pattern 0: head h at h * 84 KiB
    const size_t max1 = (size_t)(heads / 12) * 197 * 4608;           // pattern 1: [B = heads/12][197][4608]
    const size_t total = (max0 > max1 ? max0 : max1) + 4096;
    if (hipMalloc(&src, total) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("allocation failed\n"); return 1; }
    hipMemset(src, 1, total);
    const double bytes0 = (double)heads * 84 * 1024, bytes1 = (double)heads * (3 * 197 * 128);
    float t;
    t = run<0, 1>(src, heads, 0, sink, 5); printf("contiguous 1 KiB pieces, 1 batch in flight : %.1f us  %.0f GB/s\n", t * 1e3, bytes0 / t * 1e-6);
    t = run<0, 2>(src, heads, 0, sink, 5); printf("contiguous 1 KiB pieces, 2 batches in flight: %.1f us  %.0f GB/s\n", t * 1e3, bytes0 / t * 1e-6);
    t = run<1, 1>(src, heads, 0, sink, 5); printf("8 x 128 B rows (packed qkv), 1 batch in flight : %.1f us  %.0f GB/s of distinct bytes\n", t * 1e3, bytes1 / t * 1e-6);
    t = run<1, 2>(src, heads, 0, sink, 5); printf("8 x 128 B rows (packed qkv), 2 batches in flight: %.1f us  %.0f GB/s of distinct bytes\n", t * 1e3, bytes1 / t * 1e-6);
    return 0;
}
