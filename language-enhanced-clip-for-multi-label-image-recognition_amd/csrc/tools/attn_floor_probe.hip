// Microbenchmark (diagnostic, not part of the library), round 5: the access-pattern floor of the ViT-B attention kernel (task 3b).
// A pure streamer that moves exactly what attn_heads_kernel moves for one layer at B = 256 - per (image, head) the 197 q, k and v rows of 128 bytes
// at the packed-qkv row stride of 4 608 bytes by LDS-DMA (8 rows per 1 KiB piece, two heads in flight, as the kernel stages them) and the 197 x 128 B
// of its output at a 1 536-byte row stride with 16-byte stores - and computes nothing: 232.4 MB read + 77.5 MB written per launch.
// The qkv buffer is rewritten by a stand-in producer before every timed launch (the qkv GEMM has just written it in the model: whatever part of it
// the 256 MiB memory-side cache holds then, it holds here), the (image, head) pairs are walked from the last to the first like the kernel's default.
// Build: hipcc --offload-arch=gfx950 -O3 attn_floor_probe.hip -o attn_floor_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef __attribute__((ext_vector_type(4))) int i32x4;

constexpr int T = 197, HEADS = 12, D = 768, ROW = 3 * D * 2, OROW = D * 2;   // bytes
constexpr int PIECES = 75;   // 25 pieces of 8 rows for each of q, k, v (the last piece of each repeats row 196)

template <bool STORE, bool LOAD, int LOOK = 1>
__global__ __launch_bounds__(512) void stream_heads(const char* qkv, char* out, int pairs, int reverse, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // two heads x 75 KiB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int it = 0;
    const i32x4 val = {lane, wave, 3, 4};
    for (int v = blockIdx.x; v < pairs; v += gridDim.x, ++it) {
        const int hd = reverse ? pairs - 1 - v : v;
        const int b = hd / HEADS, h = hd - b * HEADS;
        if (LOAD) {
            for (int pc = wave; pc < PIECES; pc += 8) {
                int row = (pc % 25) * 8 + (lane >> 3);
                row = row < T ? row : T - 1;
                const char* p = qkv + ((int64_t)b * T + row) * ROW + (pc / 25) * (D * 2) + h * 128 + (lane & 7) * 16;
                __builtin_amdgcn_global_load_lds((const void*)p, LDS_PTR(smem + ((it & 1) * PIECES + pc) * 1024), 16, 0, 0);
            }
            if (LOOK) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(10 + (STORE ? 4 : 0)) : "memory");   // the previous head's pieces have landed, this head's (and the last stores) stay in flight
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                     // LOOK = 0: no head in flight across the barrier
        }
        __builtin_amdgcn_s_barrier();
        if (STORE) {
            // the head's 197 output rows, 8 rows per wave instruction: 25 instructions over 8 waves
            for (int pc = wave; pc < 25; pc += 8) {
                const int row = pc * 8 + (lane >> 3);
                if (row < T) *(i32x4*)(out + ((int64_t)b * T + row) * OROW + h * 128 + (lane & 7) * 16) = val;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && smem[17] == 123) sink[0] = 1;
}

__global__ void fill(i32x4* p, size_t n16, int salt) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = i32x4{(int)i, salt, 1, 2};
}

template <class K>
static void run(const char* name, K kern, char* qkv, char* out, int B, int* sink, double bytes) {
    const int pairs = B * HEADS;
    const size_t qbytes = (size_t)B * T * ROW;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    std::vector<float> us;
    for (int r = 0; r < 12; ++r) {
        hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, (i32x4*)qkv, qbytes / 16, r);   // the producer: qkv freshly written
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 150 * 1024, 0, (const char*)qkv, out, pairs, 1, sink);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (r >= 2) us.push_back(ms * 1e3f);
    }
    std::sort(us.begin(), us.end());
    printf("%-44s B=%d: median %.1f us, min %.1f us  (%.2f TB/s at the median)\n", name, B, us[us.size() / 2], us[0], bytes / us[us.size() / 2] * 1e-6);
}

int main() {
    char *qkv = nullptr, *out = nullptr;
    int* sink = nullptr;
    const int Bmax = 256;
    if (hipMalloc(&qkv, (size_t)Bmax * T * ROW + 4096) != hipSuccess || hipMalloc(&out, (size_t)Bmax * T * OROW + 4096) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("allocation failed\n"); return 1; }
    for (int B : {256, 128}) {
        const double rd = (double)B * HEADS * 3 * T * 128, wr = (double)B * HEADS * T * 128;
        run("loads + stores (the kernel's traffic)", stream_heads<true, true>, qkv, out, B, sink, rd + wr);
        run("loads only", stream_heads<false, true>, qkv, out, B, sink, rd);
        run("stores only", stream_heads<true, false>, qkv, out, B, sink, wr);
        run("loads + stores, NO head of lookahead", stream_heads<true, true, 0>, qkv, out, B, sink, rd + wr);
    }
    return 0;
}
