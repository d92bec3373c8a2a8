// Microbenchmark (diagnostic, not part of the library): the K loop of the 256x256x64 GEMM tile in two workgroup shapes, on the same
// LDS image (K-split slots, 64-byte rows, XOR swizzle), the same LDS-DMA feed and the same MFMA (v_mfma_f32_16x16x32_f16), no epilogue:
//   PP8: the shipped schedule - 8 waves (2 x 4, 128x64 per wave, 128 accumulator VGPRs), ping-pong: four 16-MFMA phases per K-tile,
//        two barriers per phase, the second M-half one barrier behind (gemm_mfma256.hip PP::ktile<0>)
//   W4 : the candidate of DESIGN.md section 9 (round 2) - 4 waves (2 x 2, 128x128 per wave, 256 accumulator registers: one wave per
//        SIMD, 512-register budget), every wave software-pipelined by itself: the 16 fragment reads of the next half K-tile and the 8
//        LDS-DMA pieces of the half four ahead interleaved with the 64 MFMAs of the current one, one barrier per half K-tile.  A third
//        fewer fragment reads per MFMA than PP8 (128 KiB instead of 192 KiB of LDS reads per K-tile and CU).
// Both walk the tiles of M = 50 432 persistently (one workgroup per CU) and run nk K-tiles per tile back to back; operands are random
// fp16 so that the chip clocks as under the real kernel.  Results are not checked (the accumulators feed a sink).
// Build: hipcc --offload-arch=gfx950 -O3 kloop_probe.hip -o kloop_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define PIN() __builtin_amdgcn_sched_barrier(0)
#define LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)
typedef __attribute__((ext_vector_type(8))) _Float16 v8;
typedef __attribute__((ext_vector_type(4))) float acc4;
typedef __attribute__((ext_vector_type(16))) float acc16;
constexpr int SLOT = 256 * 64, STAGE = 4 * SLOT;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
__device__ __forceinline__ acc4 mfma16(v8 a, v8 b, acc4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// ------------------------------------------------------------------------------------------------ PP8 (shipped schedule)
__global__ __launch_bounds__(512, 2) void kloop_pp8(const _Float16* A, const _Float16* W, int64_t M, int K, int tiles_n, int tiles_total, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fc = lane >> 4;
    const int frd = fr * 64 + ((fc ^ ((4 - ((fr >> 2) & 3)) & 3)) << 4);
    const int a_rd = wm * (128 * 64) + frd, b_rd = wn * (64 * 64) + frd;
    const int dma_off[2] = {wave * 1024, (wave + 8) * 1024};
    const int dma_c = ((lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3)) * 8, dma_r = lane >> 2;
    const int nk = K / 64;
    acc4 acc[2][4][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h][i][j] = acc4{0.f, 0.f, 0.f, 0.f};
    v8 af[4], bf[4];
    const _Float16* a_src[2];
    const _Float16* w_src[2];
    auto stage = [&](bool is_b, int st, int kh, int k_elem) {
        char* slot = smem + st * STAGE + (2 * kh + (is_b ? 1 : 0)) * SLOT;
#pragma unroll
        for (int u = 0; u < 2; ++u) __builtin_amdgcn_global_load_lds((const void*)((is_b ? w_src[u] : a_src[u]) + k_elem), LDS_PTR(slot + dma_off[u]), 16, 0, 0);
    };
    auto read_a = [&](int st, int kh, int rh) {
        const char* p = smem + st * STAGE + (2 * kh) * SLOT + a_rd + rh * (64 * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *(const v8*)(p + i * 1024);
    };
    auto read_b = [&](int st, int kh) {
        const char* p = smem + st * STAGE + (2 * kh + 1) * SLOT + b_rd;
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = *(const v8*)(p + j * 1024);
    };
    auto compute = [&](int rh) {
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PIN();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[rh][i][j] = mfma16(af[i], bf[j], acc[rh][i][j]);
        __builtin_amdgcn_s_setprio(0);
        PIN();
        __builtin_amdgcn_s_barrier();
        PIN();
    };
    for (int v = blockIdx.x; v < tiles_total; v += gridDim.x) {
        const int tile = xcd_remap(v, tiles_total);
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = (wave + 8 * u) * 16 + dma_r;
            int64_t ar = (int64_t)tm * 256 + r;
            ar = ar < M ? ar : M - 1;
            a_src[u] = A + ar * K + dma_c;
            w_src[u] = W + (int64_t)(tn * 256 + r) * K + dma_c;
        }
        stage(true, 0, 0, 0); stage(false, 0, 0, 0); stage(true, 0, 1, 32); stage(false, 0, 1, 32); stage(true, 1, 0, 64); stage(false, 1, 0, 64);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nk; ++t) {      // (the probe restages K-tiles modulo nk: every K-tile runs the steady-state body)
            const int s = t & 1;
            const int k1 = ((t + 1) % nk) * 64, k2 = ((t + 2) % nk) * 64;
            read_a(s, 0, 0); read_b(s, 0); stage(true, 1 - s, 1, k1 + 32); PIN(); compute(0);
            read_a(s, 0, 1); stage(false, 1 - s, 1, k1 + 32); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); PIN(); compute(1);
            read_a(s, 1, 0); read_b(s, 1); stage(true, s, 0, k2); PIN(); compute(0);
            read_a(s, 1, 1); stage(false, s, 0, k2); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); PIN(); compute(1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wm == 0) __builtin_amdgcn_s_barrier();
        __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[h][i][j][0] + acc[h][i][j][1] + acc[h][i][j][2] + acc[h][i][j][3];
    if (s == 12345.678f) sink[0] = s;
}

// ------------------------------------------------------------------------------------------------ PP8 with v_mfma_f32_32x32x16 (same schedule, LDS image and feed; 8 MFMAs of 32 cycles per phase instead of 16 of 16)
__global__ __launch_bounds__(512, 2) void kloop_pp8_32(const _Float16* A, const _Float16* W, int64_t M, int K, int tiles_n, int tiles_total, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    // 32x32x16 operand map: lane l -> row l & 31 of a 32-row block, k 8 (l >> 5) .. + 7 of the 16-wide step: 16-byte chunk 2 ks + (l >> 5) of the slot's 64-byte row
    const int fr = lane & 31, fh = lane >> 5;
    const int sw = (4 - ((fr >> 2) & 3)) & 3;
    const int frd0 = fr * 64 + (((0 + fh) ^ sw) << 4), frd1 = fr * 64 + (((2 + fh) ^ sw) << 4);
    const int a_rd = wm * (128 * 64), b_rd = wn * (64 * 64);
    const int dma_off[2] = {wave * 1024, (wave + 8) * 1024};
    const int dma_c = ((lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3)) * 8, dma_r = lane >> 2;
    const int nk = K / 64;
    acc16 acc[2][2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.f;
    v8 af[4], bf[4];
    const _Float16* a_src[2];
    const _Float16* w_src[2];
    auto stage = [&](bool is_b, int st, int kh, int k_elem) {
        char* slot = smem + st * STAGE + (2 * kh + (is_b ? 1 : 0)) * SLOT;
#pragma unroll
        for (int u = 0; u < 2; ++u) __builtin_amdgcn_global_load_lds((const void*)((is_b ? w_src[u] : a_src[u]) + k_elem), LDS_PTR(slot + dma_off[u]), 16, 0, 0);
    };
    auto read_a = [&](int st, int kh, int rh) {
        const char* p = smem + st * STAGE + (2 * kh) * SLOT + a_rd + rh * (64 * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *(const v8*)(p + (i >> 1) * 2048 + ((i & 1) ? frd1 : frd0));   // i = 2 * row block + k-step
    };
    auto read_b = [&](int st, int kh) {
        const char* p = smem + st * STAGE + (2 * kh + 1) * SLOT + b_rd;
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = *(const v8*)(p + (j >> 1) * 2048 + ((j & 1) ? frd1 : frd0));
    };
    auto compute = [&](int rh) {
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PIN();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[rh][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[2 * i + ks], bf[2 * j + ks], acc[rh][i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        PIN();
        __builtin_amdgcn_s_barrier();
        PIN();
    };
    for (int v = blockIdx.x; v < tiles_total; v += gridDim.x) {
        const int tile = xcd_remap(v, tiles_total);
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = (wave + 8 * u) * 16 + dma_r;
            int64_t ar = (int64_t)tm * 256 + r;
            ar = ar < M ? ar : M - 1;
            a_src[u] = A + ar * K + dma_c;
            w_src[u] = W + (int64_t)(tn * 256 + r) * K + dma_c;
        }
        stage(true, 0, 0, 0); stage(false, 0, 0, 0); stage(true, 0, 1, 32); stage(false, 0, 1, 32); stage(true, 1, 0, 64); stage(false, 1, 0, 64);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nk; ++t) {      // (the probe restages K-tiles modulo nk: every K-tile runs the steady-state body)
            const int s = t & 1;
            const int k1 = ((t + 1) % nk) * 64, k2 = ((t + 2) % nk) * 64;
            read_a(s, 0, 0); read_b(s, 0); stage(true, 1 - s, 1, k1 + 32); PIN(); compute(0);
            read_a(s, 0, 1); stage(false, 1 - s, 1, k1 + 32); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); PIN(); compute(1);
            read_a(s, 1, 0); read_b(s, 1); stage(true, s, 0, k2); PIN(); compute(0);
            read_a(s, 1, 1); stage(false, s, 0, k2); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); PIN(); compute(1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wm == 0) __builtin_amdgcn_s_barrier();
        __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[h][i][j][r];
    if (s == 12345.678f) sink[0] = s;
}

// ------------------------------------------------------------------------------------------------ W4 (candidate)
__global__ __launch_bounds__(256, 1) void kloop_w4(const _Float16* A, const _Float16* W, int64_t M, int K, int tiles_n, int tiles_total, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fc = lane >> 4;
    const int frd = fr * 64 + ((fc ^ ((4 - ((fr >> 2) & 3)) & 3)) << 4);
    const int a_rd = wm * (128 * 64) + frd, b_rd = wn * (128 * 64) + frd;
    const int dma_c = ((lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3)) * 8, dma_r = lane >> 2;
    const int nk = K / 64, nh = 2 * nk;
    acc4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = acc4{0.f, 0.f, 0.f, 0.f};
    v8 fa[2][8], fb[2][8];      // two fragment sets: the reads of half n+1 land while half n is multiplied
    const _Float16* a_src[4];
    const _Float16* w_src[4];
    // one LDS-DMA piece of half h (ring slot pair h & 3): u = 0..3 the A pieces wave + 4u, u = 4..7 the B pieces
    auto dma1 = [&](int h, int u) {
        const int st = (h >> 1) & 1, kh = h & 1;
        const bool is_b = u >= 4;
        const int uu = u & 3;
        char* slot = smem + st * STAGE + (2 * kh + (is_b ? 1 : 0)) * SLOT;
        const int k_elem = ((h >> 1) % nk) * 64 + kh * 32;
        __builtin_amdgcn_global_load_lds((const void*)((is_b ? w_src[uu] : a_src[uu]) + k_elem), LDS_PTR(slot + (wave + 4 * uu) * 1024), 16, 0, 0);
    };
    auto read1 = [&](int h, int set, int q) {      // q = 0..7 A row blocks, 8..15 B column blocks
        const int st = (h >> 1) & 1, kh = h & 1;
        if (q < 8) fa[set][q] = *(const v8*)(smem + st * STAGE + (2 * kh) * SLOT + a_rd + q * 1024);
        else fb[set][q - 8] = *(const v8*)(smem + st * STAGE + (2 * kh + 1) * SLOT + b_rd + (q - 8) * 1024);
    };
    for (int v = blockIdx.x; v < tiles_total; v += gridDim.x) {
        const int tile = xcd_remap(v, tiles_total);
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = (wave + 4 * u) * 16 + dma_r;
            int64_t ar = (int64_t)tm * 256 + r;
            ar = ar < M ? ar : M - 1;
            a_src[u] = A + ar * K + dma_c;
            w_src[u] = W + (int64_t)(tn * 256 + r) * K + dma_c;
        }
        // halves 0..3 in flight (32 pieces per wave), half 0 landed, its fragments in set 0
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int u = 0; u < 8; ++u) dma1(h, u);
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int q = 0; q < 16; ++q) read1(0, 0, q);
        LGKM0();
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");     // half 1 landed (this wave's pieces)
        for (int n = 0; n < nh; n += 2) {
#pragma unroll
            for (int par = 0; par < 2; ++par) {             // half n + par multiplies fragment set par, loads set 1 - par
                const int h = n + par;
                PIN();
                __builtin_amdgcn_s_barrier();               // every wave has retired its reads of half h - 1... and its pieces of half h + 1 have landed
                PIN();
#pragma unroll
                for (int m = 0; m < 64; ++m) {
                    // 16 fragment reads of half h + 1 beside the first 32 MFMAs, 8 LDS-DMA pieces of half h + 4 (into half h's slots:
                    // their last read retired before the barrier one half ago... the probe restages modulo nk) beside the last 32
                    if (m < 32 && !(m & 1)) read1(h + 1, 1 - par, m >> 1);
                    if (m >= 32 && !(m & 3)) dma1(h + 4, (m - 32) >> 2);
                    PIN();
                    const int i = m >> 3, j = m & 7;
                    acc[i][j] = mfma16(fa[par][i], fb[par][j], acc[i][j]);
                    PIN();
                }
                LGKM0();
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // half h + 2 landed; h + 3 and h + 4 (16 pieces) stay in flight
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f) sink[0] = s;
}

template <class Kern>
static float run(Kern k, int threads, const _Float16* A, const _Float16* W, int64_t M, int N, int K, float* sink, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int tiles_n = N / 256, tiles_total = (int)((M + 255) / 256) * tiles_n;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(threads), 2 * STAGE, 0, A, W, M, K, tiles_n, tiles_total, sink);
    hipEventRecord(a, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(threads), 2 * STAGE, 0, A, W, M, K, tiles_n, tiles_total, sink);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}

int main() {
    const int64_t M = 50432;
    _Float16 *A = nullptr, *W = nullptr;
    float* sink = nullptr;
    const size_t a_n = (size_t)M * 3072, w_n = (size_t)3072 * 3072;
    if (hipMalloc(&A, a_n * 2) != hipSuccess || hipMalloc(&W, w_n * 2) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("allocation failed\n"); return 1; }
    {
        std::vector<_Float16> h(a_n > w_n ? a_n : w_n);
        unsigned x = 12345u;
        for (size_t i = 0; i < h.size(); ++i) { x = x * 1664525u + 1013904223u; h[i] = (_Float16)(((int)(x >> 9) % 4096 - 2048) * (1.0f / 1024.0f)); }
        hipMemcpy(A, h.data(), a_n * 2, hipMemcpyHostToDevice);
        hipMemcpy(W, h.data(), w_n * 2, hipMemcpyHostToDevice);
    }
    struct Shape { const char* name; int N, K; } shapes[] = {{"qkv", 2304, 768}, {"out_proj", 768, 768}, {"c_fc", 3072, 768}, {"c_proj", 768, 3072}};
    printf("K loop only, us per launch (M = 50432, fp16, random operands), three rounds\n%-10s %10s %10s %10s\n", "shape", "PP8", "PP8-32x32", "W4");
    for (int round = 0; round < 3; ++round)
        for (const Shape& s : shapes) {
            const float t8 = run(kloop_pp8, 512, A, W, M, s.N, s.K, sink, 10);
            const float t32 = run(kloop_pp8_32, 512, A, W, M, s.N, s.K, sink, 10);
            const float t4 = getenv("KLOOP_W4") ? run(kloop_w4, 256, A, W, M, s.N, s.K, sink, 10) : 0.f;
            const double fl = 2.0 * M * s.N * s.K;
            printf("%-10s %10.1f %10.1f %10.1f   (%.0f / %.0f TFLOP/s)\n", s.name, t8, t32, t4, fl / t8 * 1e-6, fl / t32 * 1e-6);
        }
    return 0;
}
