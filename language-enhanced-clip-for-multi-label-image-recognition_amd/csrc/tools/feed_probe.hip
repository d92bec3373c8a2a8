// Microbenchmark (diagnostic, not part of the library): how fast can one CU's eight waves pull a 256x256x64 GEMM K-tile's
// operands (64 KiB) into LDS with LDS-DMA, in the access patterns the GEMM kernel could use, with and without MFMAs issued
// beside the loads?  Same tile walk as gemm_tn_256x256x64_pp (persistent, one 512-thread workgroup per CU, XCD-contiguous
// tile ranges), nothing is read back from LDS, the MFMAs run on register operands.
//   SHAPE 0: piece = 16 rows x 64 B  (K-split slots, what the kernel does today: every 128-B line is requested as two halves,
//            two phases apart)
//   SHAPE 1: piece = 8 rows x 128 B  (whole cache lines)
//   SHAPE 2: 16 rows x 64 B, but the two halves of a line are requested back to back (same wave, consecutive instructions)
//   MF: MFMAs (16x16x32 f16) per wave per K-tile (64 = what the real tile needs; 0 = loads alone)
//   NODMA: MFMAs alone (the loop's floor)
// Build: hipcc --offload-arch=gfx950 -O3 feed_probe.hip -o feed_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float acc4;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

template <int SHAPE, int MF, bool NODMA, int INFLIGHT>
__global__ __launch_bounds__(512, 2) void feed(const char* A, const char* W, int64_t M, int N, int K, int tiles_n, int tiles_total, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nk = K / 64;
    const int64_t ld = (int64_t)K * 2;
    acc4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = acc4{0.f, 0.f, 0.f, 0.f};
    f16x8 fa, fb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(0.001f * (lane + i)); fb[i] = (_Float16)(0.002f * (lane - i)); }
    int kt_global = 0;
    for (int v = blockIdx.x; v < tiles_total; v += gridDim.x) {
        const int tile = xcd_remap(v, tiles_total);
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        const int64_t m0 = (int64_t)tm * 256;
        const int n0 = tn * 256;
        for (int kt = 0; kt < nk; ++kt, ++kt_global) {
            char* stage = smem + (kt_global & 1) * 65536;
            if (!NODMA) {
                if (SHAPE == 0 || SHAPE == 2) {
                    // slot order A.k0 B.k0 A.k1 B.k1 (SHAPE 0) or A.k0 A.k1 B.k0 B.k1 per row group (SHAPE 2)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int kh = SHAPE == 0 ? (q >> 1) : (q & 1);
                        const int isw = SHAPE == 0 ? (q & 1) : (q >> 1);
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int r = (wave + 8 * u) * 16 + (lane >> 2);
                            int64_t row = isw ? (n0 + r) : (m0 + r < M ? m0 + r : M - 1);
                            const char* src = (isw ? W : A) + row * ld + kt * 128 + kh * 64 + (lane & 3) * 16;
                            __builtin_amdgcn_global_load_lds((const void*)src, LDS_PTR(stage + (2 * kh + isw) * 16384 + (wave + 8 * u) * 1024), 16, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int isw = 0; isw < 2; ++isw)
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int r = (wave + 8 * u) * 8 + (lane >> 3);
                            int64_t row = isw ? (n0 + r) : (m0 + r < M ? m0 + r : M - 1);
                            const char* src = (isw ? W : A) + row * ld + kt * 128 + (lane & 7) * 16;
                            __builtin_amdgcn_global_load_lds((const void*)src, LDS_PTR(stage + isw * 32768 + (wave + 8 * u) * 1024), 16, 0, 0);
                        }
                }
            }
            if (MF > 0) {
#pragma unroll
                for (int i = 0; i < MF; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[i & 7], 0, 0, 0);
            }
            if (!NODMA) {
                if (INFLIGHT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // the previous K-tile's pieces have landed
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f || (threadIdx.x == 0 && smem[17] == 123)) sink[0] = s;
}

// Register staging instead of LDS-DMA: global_load_dwordx4 into VGPRs (two sets of 8 x 16 B per wave), ds_write_b128 one K-tile later.
// Same bytes, same piece shape as SHAPE 0 (16 rows x 64 B per wave instruction), same MFMAs, one barrier per K-tile.
typedef __attribute__((ext_vector_type(4))) int i32x4;
template <int MF>
__global__ __launch_bounds__(512, 2) void feed_regs(const char* A, const char* W, int64_t M, int N, int K, int tiles_n, int tiles_total, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nk = K / 64;
    const int64_t ld = (int64_t)K * 2;
    acc4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = acc4{0.f, 0.f, 0.f, 0.f};
    f16x8 fa, fb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(0.001f * (lane + i)); fb[i] = (_Float16)(0.002f * (lane - i)); }
    i32x4 ra[8], rb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { ra[i] = i32x4{0, 0, 0, 0}; rb[i] = i32x4{0, 0, 0, 0}; }
    int kt_global = 0;
    for (int v = blockIdx.x; v < tiles_total; v += gridDim.x) {
        const int tile = xcd_remap(v, tiles_total);
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        const int64_t m0 = (int64_t)tm * 256;
        const int n0 = tn * 256;
        for (int kt = 0; kt < nk; ++kt, ++kt_global) {
            char* stage = smem + (kt_global & 1) * 65536;
            // the set loaded during the previous K-tile goes to LDS now (its loads have had a whole K-tile to land)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                i32x4 val = (kt_global & 1) ? rb[q] : ra[q];
                *(i32x4*)(stage + (q >> 1) * 16384 + (wave + 8 * (q & 1)) * 1024 + lane * 16) = val;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int kh = q >> 2, isw = (q >> 1) & 1, u = q & 1;
                const int r = (wave + 8 * u) * 16 + (lane >> 2);
                int64_t row = isw ? (n0 + r) : (m0 + r < M ? m0 + r : M - 1);
                const i32x4 val = *(const i32x4*)((isw ? W : A) + row * ld + kt * 128 + kh * 64 + (lane & 3) * 16);
                if (kt_global & 1) ra[q] = val; else rb[q] = val;
            }
            if (MF > 0) {
#pragma unroll
                for (int i = 0; i < MF; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[i & 7], 0, 0, 0);
            }
            __builtin_amdgcn_s_barrier();
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + (float)ra[i][0] * 1e-30f + (float)rb[i][1] * 1e-30f;
    if (s == 12345.678f || (threadIdx.x == 0 && smem[17] == 123)) sink[0] = s;
}

template <int MF>
static float run_regs(const char* A, const char* W, int64_t M, int N, int K, float* sink, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    auto k = feed_regs<MF>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int tiles_n = N / 256, tiles_total = (int)((M + 255) / 256) * tiles_n;
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 131072, 0, A, W, M, N, K, tiles_n, tiles_total, sink);
    hipEventRecord(a, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(512), 131072, 0, A, W, M, N, K, tiles_n, tiles_total, sink);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}

template <int SHAPE, int MF, bool NODMA, int INFLIGHT>
static float run(const char* A, const char* W, int64_t M, int N, int K, float* sink, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    auto k = feed<SHAPE, MF, NODMA, INFLIGHT>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int tiles_n = N / 256, tiles_total = (int)((M + 255) / 256) * tiles_n;
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 131072, 0, A, W, M, N, K, tiles_n, tiles_total, sink);
    hipEventRecord(a, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(512), 131072, 0, A, W, M, N, K, tiles_n, tiles_total, sink);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}

int main() {
    const int64_t M = 50432;
    char *A = nullptr, *W = nullptr;
    float* sink = nullptr;
    const size_t a_bytes = (size_t)M * 3072 * 2, w_bytes = (size_t)3072 * 3072 * 2;
    if (hipMalloc(&A, a_bytes) != hipSuccess || hipMalloc(&W, w_bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("allocation failed\n"); return 1; }
    hipMemset(A, 1, a_bytes);
    hipMemset(W, 1, w_bytes);
    struct Shape { const char* name; int N, K; } shapes[] = {{"qkv", 2304, 768}, {"out_proj", 768, 768}, {"c_fc", 3072, 768}, {"c_proj", 768, 3072}};
    printf("%-10s %9s %9s %9s %9s %9s %9s %9s %9s   (us per launch; K-tiles/CU; GB/s per CU for the DMA-only column of shape 0)\n", "shape", "mfma", "dma64", "dma128", "dma64bb", "64+mf", "128+mf", "64bb+mf", "128+mf/w0");
    for (const Shape& s : shapes) {
        const int reps = 5;
        const float t_m = run<0, 64, true, 8>(A, W, M, s.N, s.K, sink, reps);
        const float t0 = run<0, 0, false, 8>(A, W, M, s.N, s.K, sink, reps);
        const float t1 = run<1, 0, false, 8>(A, W, M, s.N, s.K, sink, reps);
        const float t2 = run<2, 0, false, 8>(A, W, M, s.N, s.K, sink, reps);
        const float t0m = run<0, 64, false, 8>(A, W, M, s.N, s.K, sink, reps);
        const float t1m = run<1, 64, false, 8>(A, W, M, s.N, s.K, sink, reps);
        const float t2m = run<2, 64, false, 8>(A, W, M, s.N, s.K, sink, reps);
        const float t1w = run<1, 64, false, 0>(A, W, M, s.N, s.K, sink, reps);
        const float tr0 = run_regs<0>(A, W, M, s.N, s.K, sink, reps), trm = run_regs<64>(A, W, M, s.N, s.K, sink, reps);
        const double tiles = (double)((M + 255) / 256) * (s.N / 256);
        const double kt_per_cu = tiles * (s.K / 64) / 256.0;
        printf("%-10s %9.1f %9.1f %9.1f %9.1f %9.1f %9.1f %9.1f %9.1f   kt/cu %.1f  dma64 %.1f GB/s/CU  dma128 %.1f GB/s/CU   | register staging: loads alone %.1f, + mfma %.1f\n", s.name, t_m, t0, t1, t2, t0m, t1m, t2m, t1w,
               kt_per_cu, kt_per_cu * 65536 / t0 * 1e-3, kt_per_cu * 65536 / t1 * 1e-3, tr0, trm);
    }
    return 0;
}
