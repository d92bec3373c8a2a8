// Microbenchmark (diagnostic, not part of the library), round 5: K loops of WIDER output tiles beside the shipped 256x256 one.
//
// Why: every 256x256 kernel measured on this chip (this build's ping-pong loop, the 4-wave / AGPR candidate, the vendor's MT256x256x64) ends at
// 0.45 - 0.53 of the MFMA peak in its K loop.  A 256x256x64 K-tile needs 64 KiB of operands per 2 048 MFMA cycles of a SIMD = 32 B per cycle and
// CU at 100 % duty x 2 (four SIMDs work on the same tile: 64 KiB per 2 048 cycles of wall time) ... the loop draws ~35 B / cycle / CU at its 56 %
// duty, which is what a CU has been measured to take in from L2 (MI355X_MICROARCH.md, gather into LDS: 66 - 73 GB/s per CU).  If the feed is the
// limit, the lever is fewer operand bytes per FLOP, i.e. a larger tile: bytes per MAC ~ 1/TM + 1/TN.
//   PP8 : shipped schedule, 256x256, 8 waves as 2 x 4, 128x64 per wave (128 accumulator VGPRs)                      1/TM + 1/TN = 1/128
//   PPX<256,384>: 8 waves as 2 x 4, 128x96 per wave (192 accumulator VGPRs), 24-MFMA phases, ring of R 32-deep steps   = 1/153.6 (-17 %)
//   PPX<128,768>: 8 waves as 1 x 8, 128x96 per wave - the FULL-ROW tile of the round-4 verdict (LayerNorm in-kernel)   = 1/109.7 (+17 %)
// All: v_mfma_f32_16x16x32_f16, ascending K, 64-byte-row K-split LDS image with the XOR swizzle, LDS-DMA feed, ping-pong (waves w and w + 4 share a
// SIMD and run one barrier apart), no epilogue, random operands, results not checked (the accumulators feed a sink).
// Build: hipcc --offload-arch=gfx950 -O3 kloop_probe2.hip -o kloop_probe2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define PIN() __builtin_amdgcn_sched_barrier(0)
typedef __attribute__((ext_vector_type(8))) _Float16 v8;
typedef __attribute__((ext_vector_type(4))) float acc4;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
__device__ __forceinline__ acc4 mfma16(v8 a, v8 b, acc4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// ------------------------------------------------------------------------------------------------ PP8 (shipped schedule, as kloop_probe.hip)
constexpr int SLOT = 256 * 64, STAGE = 4 * SLOT;
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

// ------------------------------------------------------------------------------------------------ PPX: TM x TN tile, 128 x 96 per wave
// One "step" = 32 of K: an A slot (TM rows x 64 B) and a B slot (TN rows x 64 B), a ring of R step slots.  A wave's step is two phases of 24 MFMAs:
//   phase 0: read A rows 0..63 of its 128 (4 fragments) + its 6 B fragments          phase 1: read A rows 64..127 (4 fragments)
// Step s + R - 1 is staged into the slot step s - 1 used (B in phase 0, A in phase 1): B of step s - 1 was last read in phase (s-1, 0), A in phase
// (s-1, 1) - by the other wave group at most one barrier interval later than by this one, and retired (lgkmcnt(0)) one interval after that; a whole
// step lies in between.  One counted wait per step (phase 1): everything up to step s + 1 has landed, R - 2 steps stay in flight.
// MODE 0: no sub-options.  All LDS-DMA through buffer descriptors: one 32-bit per-lane offset per operand, the piece / tile / k part in the scalar offset.
template <int TM, int TN, int R>
__global__ __launch_bounds__(512, 2) void kloop_ppx(const _Float16* A, const _Float16* W, int64_t M, int N, int K, int tiles_total, float* sink) {
    constexpr int WN = TN / 96;
    static_assert((TM / 128) * WN == 8, "8 waves of 128 x 96");
    constexpr int A_SLOT = TM * 64, B_SLOT = TN * 64, PAIR = A_SLOT + B_SLOT;
    constexpr int APW = TM / 128, BPW = TN / 128;   // LDS-DMA pieces (16 rows x 64 B) per wave and step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int fr = lane & 15, fc = lane >> 4;
    const int frd = fr * 64 + ((fc ^ ((4 - ((fr >> 2) & 3)) & 3)) << 4);
    const int a_rd = wm * (128 * 64) + frd, b_rd = A_SLOT + wn * (96 * 64) + frd;
    const int dma_c = ((lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3)) * 16, dma_r = lane >> 2;
    const int voff = dma_r * K * 2 + dma_c;                      // row of the piece, 16-byte chunk (swizzled on the source side)
    const int nsteps = K / 32;                                   // multiple of R (host)
    const int tiles_n = N / TN;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, N * K * 2, 0x00020000);
    acc4 acc[8][6];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = acc4{0.f, 0.f, 0.f, 0.f};
    v8 af[4], bf[6];
    int a_base = 0, w_base = 0;     // byte offsets of this wave's first piece rows of the tile (scalar)
    auto stage_a = [&](int slot, int step) {
#pragma unroll
        for (int u = 0; u < APW; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(smem + slot * PAIR + (wave + 8 * u) * 1024), 16, voff, a_base + (u * 128 * K + step * 32) * 2, 0, 0);
    };
    auto stage_b = [&](int slot, int step) {
#pragma unroll
        for (int u = 0; u < BPW; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(smem + slot * PAIR + A_SLOT + (wave + 8 * u) * 1024), 16, voff, w_base + (u * 128 * K + step * 32) * 2, 0, 0);
    };
    auto read_a = [&](int slot, int h) {
        const char* p = smem + slot * PAIR + a_rd + h * (64 * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *(const v8*)(p + i * 1024);
    };
    auto read_b = [&](int slot) {
        const char* p = smem + slot * PAIR + b_rd;
#pragma unroll
        for (int j = 0; j < 6; ++j) bf[j] = *(const v8*)(p + j * 1024);
    };
    auto compute = [&](int h) {
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PIN();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[4 * h + i][j] = mfma16(af[i], bf[j], acc[4 * h + i][j]);
        __builtin_amdgcn_s_setprio(0);
        PIN();
        __builtin_amdgcn_s_barrier();
        PIN();
    };
    for (int v = blockIdx.x; v < tiles_total; v += gridDim.x) {
        const int tile = xcd_remap(v, tiles_total);
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        a_base = __builtin_amdgcn_readfirstlane((tm * TM + wave * 16) * K * 2);
        w_base = __builtin_amdgcn_readfirstlane((tn * TN + wave * 16) * K * 2);
#pragma unroll
        for (int s = 0; s < R - 1; ++s) { stage_b(s, s); stage_a(s, s); }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * (APW + BPW)) : "memory");
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();
        for (int s0 = 0; s0 < nsteps; s0 += R) {
#pragma unroll
            for (int q = 0; q < R; ++q) {                        // step s0 + q lives in slot q
                const int nxt = (s0 + q + R - 1) % nsteps;       // (the probe restages steps modulo nsteps: every step runs the steady-state body)
                constexpr int dummy = 0; (void)dummy;
                const int ns = (q + R - 1) % R;
                read_a(q, 0); read_b(q); stage_b(ns, nxt); PIN(); compute(0);
                read_a(q, 1); stage_a(ns, nxt);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * (APW + BPW)) : "memory");
                PIN(); compute(1);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (grp == 0) __builtin_amdgcn_s_barrier();
        __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f) sink[0] = s;
}

// ------------------------------------------------------------------------------------------------ FR: 128 x 768 with split rings
// The full-row tile cannot hold two 56 KiB steps plus an epilogue region and a third does not fit at all; but in the 1 x 8 wave layout a wave's 96 B
// columns are PRIVATE to it (6 pieces = its own 6 fragments per step), so only A needs the workgroup's barriers:
//   A ring: 4 steps x 8 KiB (shared; step s + 3 staged in phase (s, 1) into the slot of step s - 1)
//   B ring: per wave 2 steps x 6 KiB (private; step s + 2 staged in phase (s, 1) into the slot step s itself used - its fragments were retired by this
//           wave's own lgkmcnt(0) in front of phase (s, 0)'s MFMAs)
// = 128 KiB, leaving 32 KiB for an epilogue as today.  Wait per step: vmcnt(8) = {A[s+2], B[s+2] x 6, A[s+3]} stay in flight.
__global__ __launch_bounds__(512, 2) void kloop_fr(const _Float16* A, const _Float16* W, int64_t M, int N, int K, int tiles_total, float* sink) {
    constexpr int A_SLOT = 128 * 64, A_RING = 4 * A_SLOT, BW = 6 * 1024;   // B bytes per wave and step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int fr = lane & 15, fc = lane >> 4;
    const int frd = fr * 64 + ((fc ^ ((4 - ((fr >> 2) & 3)) & 3)) << 4);
    const int dma_c = ((lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3)) * 16, dma_r = lane >> 2;
    const int voff = dma_r * K * 2 + dma_c;
    const int nsteps = K / 32;                                   // multiple of 4 (host)
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, N * K * 2, 0x00020000);
    char* bring = smem + A_RING + wave * (2 * BW);
    acc4 acc[8][6];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = acc4{0.f, 0.f, 0.f, 0.f};
    v8 af[4], bf[6];
    int a_base = 0;
    const int w_base = __builtin_amdgcn_readfirstlane(wave * 96 * K * 2);
    auto stage_a = [&](int slot, int step) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(smem + slot * A_SLOT + wave * 1024), 16, voff, a_base + step * 64, 0, 0);
    };
    auto stage_b = [&](int slot, int step) {
#pragma unroll
        for (int u = 0; u < 6; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(bring + slot * BW + u * 1024), 16, voff, w_base + (u * 16 * K + step * 32) * 2, 0, 0);
    };
    auto read_a = [&](int slot, int h) {
        const char* p = smem + slot * A_SLOT + frd + h * (64 * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *(const v8*)(p + i * 1024);
    };
    auto read_b = [&](int slot) {
        const char* p = bring + slot * BW + frd;
#pragma unroll
        for (int j = 0; j < 6; ++j) bf[j] = *(const v8*)(p + j * 1024);
    };
    auto compute = [&](int h) {
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PIN();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[4 * h + i][j] = mfma16(af[i], bf[j], acc[4 * h + i][j]);
        __builtin_amdgcn_s_setprio(0);
        PIN();
        __builtin_amdgcn_s_barrier();
        PIN();
    };
    for (int v = blockIdx.x; v < tiles_total; v += gridDim.x) {
        const int tile = xcd_remap(v, tiles_total);
        a_base = __builtin_amdgcn_readfirstlane((tile * 128 + wave * 16) * K * 2);
        // issue order as in the steady state: B[0] A[0] | B[1] A[1] A[2]   (then per step: B[s+2] A[s+3])
        stage_b(0, 0); stage_a(0, 0); stage_b(1, 1); stage_a(1, 1); stage_a(2, 2);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();
        for (int s0 = 0; s0 < nsteps; s0 += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = s0 + q;
                read_a(q, 0); read_b(q & 1); PIN(); compute(0);
                read_a(q, 1);
                stage_b(q & 1, (s + 2) % nsteps);
                stage_a((q + 3) & 3, (s + 3) % nsteps);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                PIN(); compute(1);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (grp == 0) __builtin_amdgcn_s_barrier();
        __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f) sink[0] = s;
}

struct Timer {
    hipEvent_t a, b;
    Timer() { hipEventCreate(&a); hipEventCreate(&b); }
    template <class F> float us(F&& launch, int reps) {
        for (int r = 0; r < 3; ++r) launch();
        hipEventRecord(a, 0);
        for (int r = 0; r < reps; ++r) launch();
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        return ms / reps * 1e3f;
    }
};

template <int TM, int TN, int R>
static float run_ppx(Timer& t, const _Float16* A, const _Float16* W, int64_t M, int N, int K, float* sink, int grid_cap) {
    constexpr int lds = R * (TM + TN) * 64;
    auto k = kloop_ppx<TM, TN, R>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int tiles = (int)((M + TM - 1) / TM) * (N / TN);
    const int grid = tiles < grid_cap ? tiles : grid_cap;
    return t.us([&] { hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, A, W, M, N, K, tiles, sink); }, 10);
}

int main(int argc, char** argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 50432;     // 25216 = one stream part
    _Float16 *A = nullptr, *W = nullptr;
    float* sink = nullptr;
    const size_t a_n = (size_t)50432 * 3072, w_n = (size_t)3072 * 3072;
    if (hipMalloc(&A, a_n * 2) != hipSuccess || hipMalloc(&W, w_n * 2) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("allocation failed\n"); return 1; }
    {
        std::vector<_Float16> h(a_n > w_n ? a_n : w_n);
        unsigned x = 12345u;
        for (size_t i = 0; i < h.size(); ++i) { x = x * 1664525u + 1013904223u; h[i] = (_Float16)(((int)(x >> 9) % 4096 - 2048) * (1.0f / 1024.0f)); }
        hipMemcpy(A, h.data(), a_n * 2, hipMemcpyHostToDevice);
        hipMemcpy(W, h.data(), w_n * 2, hipMemcpyHostToDevice);
    }
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    struct Shape { const char* name; int N, K; } shapes[] = {{"qkv", 2304, 768}, {"out_proj", 768, 768}, {"c_fc", 3072, 768}, {"c_proj", 768, 3072}};
    Timer t;
    printf("K loop only, us per launch (M = %lld, fp16, random operands, %d CUs), three rounds; TFLOP/s in brackets\n", (long long)M, cus);
    printf("%-9s %16s %16s %16s %16s\n", "shape", "PP8 256x256", "PPX 256x384 R3", "PPX 256x384 R4", "FR 128x768");
    hipFuncSetAttribute((const void*)kloop_pp8, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)kloop_fr, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int round = 0; round < 3; ++round)
        for (const Shape& s : shapes) {
            const double fl = 2.0 * M * s.N * s.K * 1e-6;
            const int t8n = s.N / 256, t8 = (int)((M + 255) / 256) * t8n;
            const float u8 = t.us([&] { hipLaunchKernelGGL(kloop_pp8, dim3(t8 < cus ? t8 : cus), dim3(512), 2 * STAGE, 0, A, W, M, s.K, t8n, t8, sink); }, 10);
            const float u3 = run_ppx<256, 384, 3>(t, A, W, M, s.N, s.K, sink, cus);
            const float u4 = run_ppx<256, 384, 4>(t, A, W, M, s.N, s.K, sink, cus);
            float uf = 0.f;
            if (s.N == 768) {
                const int tf = (int)((M + 127) / 128);
                uf = t.us([&] { hipLaunchKernelGGL(kloop_fr, dim3(tf < cus ? tf : cus), dim3(512), 128 * 1024, 0, A, W, M, s.N, s.K, tf, sink); }, 10);
            }
            printf("%-9s %8.1f (%5.0f) %8.1f (%5.0f) %8.1f (%5.0f) %8.1f (%5.0f)\n", s.name, u8, fl / u8, u3, fl / u3, u4, fl / u4, uf, uf > 0 ? fl / uf : 0.0);
        }
    return 0;
}
