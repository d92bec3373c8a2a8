// Y = act(A . W^T + bias) + residual: 256x128 output tile, 4 waves, TWO workgroups per CU.
//
// "gemm_tn_256x128x32_x2".  The 256x256 ping-pong kernel (gemm_mfma256.hip) owns a CU with one 8-wave workgroup: the two waves of a
// SIMD alternate MFMA cluster / memory cluster in lockstep, and during a tile's epilogue (and the waits around it: ~20 % of a
// K = 768 tile) the matrix pipe idles.  Here a workgroup is 4 waves - ONE per SIMD - with a 128x64 accumulator block each (the same
// wave tile, MFMA instruction and ascending-K order as the other families: bit-identical results), 72 KiB of LDS, and two such
// workgroups are resident per CU.  The two waves that share a SIMD now belong to DIFFERENT workgroups working on different tiles:
// nothing couples them but the matrix pipe itself, so one workgroup's epilogue, its pipeline fill at a tile boundary and its LDS /
// LDS-DMA latencies run under the other's MFMA clusters.  Tiles are half as large (N = 768 at M = 50 432 is 1 182 tiles on 512
// workgroup slots), at the price of 1.5x the L2 -> LDS operand traffic per FLOP of a 256x256 tile.
//
// K-loop: one LDS stage = one 32-deep k-step: A[256 rows][32 k] (16 KiB) | B[128][32 k] (8 KiB), 64-byte rows, the bank swizzle of the
// 256x256 kernel's K-split slots; a ring of THREE stages filled by LDS-DMA (global_load_lds_dwordx4, 6 pieces per wave per stage) two
// steps ahead.  Per step and wave:   s_waitcnt vmcnt(6)  (my pieces of stage t landed; stage t+1's stay in flight)  ->  s_barrier
// (every wave's pieces landed; every wave retired its reads of stage t-1)  ->  issue the DMA of stage t+2 into the ring slot stage
// t-1 just left  ->  12 ds_read_b128 (8 A + 4 B fragments)  ->  lgkmcnt(0)  ->  32 MFMAs.  One barrier per 32 MFMAs.
// No pipelining across tiles: after the last step the ring is drained, the epilogue stages through the same LDS, the next tile
// starts with a two-stage prologue - the bubble is the other workgroup's to fill.
// STATUS (round 2): correct (bit-identical to the other families: 68 GEMM / invariance tests green with it dispatched) but SLOWER than
// the 256x256 kernel in this first form: -5.5 % end to end when it takes the residual shapes (out-proj, c_proj), -13 % when it takes
// every GEMM (profiles/r02_ab_x2_two_workgroups.txt).  Each wave runs reads -> wait -> 32 MFMAs serially and issues 6 LDS-DMA pieces
// per 32 MFMAs (1.5x the operand traffic of a 256x256 tile), so a SIMD's two waves cannot keep the matrix pipe above ~80 % even when
// they interleave perfectly.  Not part of the product library (make x2 builds a variant library for A/B runs).
#include "../leclip_common.h"

int leclip_cu_count();

namespace {

constexpr int TM = 256, TN = 128, TK = 32;
constexpr int A_BYTES = TM * 64, B_BYTES = TN * 64;     // 64-byte rows
constexpr int STAGE = A_BYTES + B_BYTES;                // 24 KiB
constexpr int NSTAGE = 3;
constexpr int EPI_WAVE_BYTES = 16 * 64 * 4;             // one 16x64 fp32 strip
constexpr int LDS_BYTES = NSTAGE * STAGE;               // 72 KiB (the epilogue's 4 x 8 KiB of strips alias stages 0-1)

typedef __attribute__((ext_vector_type(4))) float acc4;
__device__ __forceinline__ acc4 mfma16(bf16x8 a, bf16x8 b, acc4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ acc4 mfma16(f16x8 a, f16x8 b, acc4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

struct X2Args {
    const void* A;
    const void* W;
    int64_t M;
    int N, K;
    int64_t lda, ldw;
    EpiParams epi;
    int tiles_n, tiles_total;
};

__device__ __forceinline__ int xcd_remap_x2(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

template <bool B> struct BoolX { static constexpr bool value = B; };

// PF / CFG as in gemm_mfma256.hip: PF 0 nothing prefetched, 1 the 16-bit residual, 2 the fused LayerNorm's (mean, rstd);
// CFG bit 0 = QuickGELU, bit 1 = emit LayerNorm block partials.  Output (and residual) in the operand dtype, no row remap.
template <typename T, int PF, int CFG>
__global__ __launch_bounds__(256, 2) void gemm_tn_256x128x32_x2(X2Args g) {
    typedef typename VecOf<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const EpiParams& e = g.epi;

    // fragment reads (v_mfma_f32_16x16x32 operand map): lane l -> row l&15 of a 16-row tile, 16-byte chunk l>>4 of the 64-byte row
    const int fr = lane & 15, fc = lane >> 4;
    const int frd = fr * 64 + ((fc ^ ((4 - ((fr >> 2) & 3)) & 3)) << 4);
    const int a_rd = wm * (128 * 64) + frd;
    const int b_rd = A_BYTES + wn * (64 * 64) + frd;
    // LDS-DMA pieces of 16 rows x 64 B: A pieces wave, wave+4, +8, +12; B pieces wave, wave+4.  Lane l writes row 16*piece + (l>>2),
    // physical chunk l&3, which must hold logical chunk (l&3) ^ f((row>>2)&3)
    const int dma_c = ((lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3)) * 8;
    const int dma_r = lane >> 2;
    const int nsteps = g.K / TK;   // >= 2 (host)

    acc4 acc[2][4][4];
    for (int v = blockIdx.x; v < g.tiles_total; v += gridDim.x) {
        const int tile = xcd_remap_x2(v, g.tiles_total);
        const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
        const int64_t m0 = (int64_t)tm * TM;
        const int n0 = tn * TN;
        const T* a_src[4];
        const T* w_src[2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int64_t ar = m0 + (wave + 4 * u) * 16 + dma_r;
            ar = ar < g.M ? ar : g.M - 1;
            a_src[u] = (const T*)g.A + ar * g.lda + dma_c;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) w_src[u] = (const T*)g.W + (int64_t)(n0 + (wave + 4 * u) * 16 + dma_r) * g.ldw + dma_c;
        auto issue = [&](int slot, int k_elem) {
            char* st = smem + slot * STAGE;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                __builtin_amdgcn_global_load_lds((const void*)(a_src[u] + k_elem), LDS_PTR(st + (wave + 4 * u) * 1024), 16, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; ++u)
                __builtin_amdgcn_global_load_lds((const void*)(w_src[u] + k_elem), LDS_PTR(st + A_BYTES + (wave + 4 * u) * 1024), 16, 0, 0);
        };
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[h][i][j] = acc4{0.f, 0.f, 0.f, 0.f};

        issue(0, 0);
        issue(1, TK);
        int slot = 0;
        for (int t = 0; t < nsteps; ++t) {
            if (t + 1 < nsteps) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < nsteps) issue(slot == 0 ? 2 : slot - 1, (t + 2) * TK);
            const char* pa = smem + slot * STAGE + a_rd;
            const char* pb = smem + slot * STAGE + b_rd;
            v8 af[8], bfr[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = *(const v8*)(pa + i * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *(const v8*)(pb + j * 1024);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i >> 2][i & 3][j] = mfma16(af[i], bfr[j], acc[i >> 2][i & 3][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            slot = slot == 2 ? 0 : slot + 1;
        }
        __syncthreads();   // every wave is done reading the ring (nothing is in flight: vmcnt(0) at the last step): the strips may alias it

        // ---- epilogue: 8 passes of 16 rows through two alternating private 4 KiB fp32 strips (as gemm_mfma256.hip's fp32-staged
        // flavour): row-major read-back, 8 columns per lane, bias / fused LayerNorm / QuickGELU / residual / block partials on
        // 8-wide chunks, 16-byte global accesses of full 128-byte lines.
        constexpr int ACT = CFG & 1, STATS = (CFG >> 1) & 1;
        const int crow0 = lane >> 3, ccol0 = (lane & 7) * 8;
        const int nb = n0 + wn * 64 + ccol0;
        float b8[8], s8[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { b8[c] = 0.f; s8[c] = 0.f; }
        if (e.bias) {
            const f32x4 t0 = *(const f32x4*)(e.bias + nb), t1 = *(const f32x4*)(e.bias + nb + 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) { b8[c] = t0[c]; b8[4 + c] = t1[c]; }
        }
        if constexpr (PF == 2) {
            const f32x4 t0 = *(const f32x4*)(e.ln_colsum + nb), t1 = *(const f32x4*)(e.ln_colsum + nb + 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) { s8[c] = t0[c]; s8[4 + c] = t1[c]; }
        }
        i32x4 rpre[PF == 1 ? 16 : 1];
        f32x2 lnpre[PF == 2 ? 2 : 1];
        if constexpr (PF == 1) {   // the tile's 16-bit residual: chunk qu = (strip qu >> 1, row half qu & 1), all before the first store
#pragma unroll
            for (int qu = 0; qu < 16; ++qu) {
                int64_t m = m0 + wm * 128 + (qu >> 1) * 16 + (qu & 1) * 8 + crow0;
                m = m < g.M ? m : g.M - 1;
                rpre[qu] = *(const i32x4*)((const char*)e.res + (m * e.ldr + nb) * 2);
            }
        }
        if constexpr (PF == 2) {   // lane (crow, c) keeps rows c*16 + u*8 + crow; a pass fetches its pair with ds_bpermute
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                int64_t m = m0 + wm * 128 + (lane & 7) * 16 + u * 8 + crow0;
                m = m < g.M ? m : g.M - 1;
                lnpre[u] = *(const f32x2*)(e.ln_stats + 2 * m);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int c = 0; c < 8; ++c) { asm volatile("" : "+v"(b8[c])); asm volatile("" : "+v"(s8[c])); }
        if constexpr (PF == 1) {
#pragma unroll
            for (int qu = 0; qu < 16; ++qu) asm volatile("" : "+v"(rpre[qu]));
        }
        if constexpr (PF == 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) asm volatile("" : "+v"(lnpre[u]));
        }
        float* st = (float*)(smem + wave * (2 * EPI_WAVE_BYTES));
        const int wsw = ((lane >> 4) & 1) << 4;
        const int wr_off = 4 * (lane >> 4) * 64 + (lane & 15);
        auto park = [&](int q) {
            float* sq = st + (q & 1) * (EPI_WAVE_BYTES / 4) + wr_off;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) sq[r * 64 + ((16 * j) ^ wsw)] = acc[q >> 2][q & 3][j][r];
        };
        auto passes = [&](auto check) {
            park(0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (q + 1 < 8) park(q + 1);
                const int64_t row_base = m0 + wm * 128 + crow0;
                T* optr = (T*)e.out + row_base * e.ldy + nb;
                float* sptr = STATS ? e.stats_out + (row_base * e.stats_slots + (nb >> 6)) * 2 : nullptr;
                const int rd_off[2] = {crow0 * 64 + (ccol0 ^ (((crow0 >> 2) & 1) << 4)), (8 + crow0) * 64 + (ccol0 ^ ((((8 + crow0) >> 2) & 1) << 4))};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float* sp = st + (q & 1) * (EPI_WAVE_BYTES / 4) + rd_off[u];
                    const f32x4 v0 = *(const f32x4*)sp, v1 = *(const f32x4*)(sp + 4);
                    const int roff = q * 16 + u * 8;
                    if (decltype(check)::value && row_base + roff >= g.M) continue;
                    float vv[8];
#pragma unroll
                    for (int c = 0; c < 4; ++c) { vv[c] = v0[c]; vv[4 + c] = v1[c]; }
                    f32x2 ln = lnpre[0];
                    if constexpr (PF == 2) {
                        const int src = ((lane & ~7) | q) << 2;
                        const float mean_l = lnpre[u][0], rstd_l = lnpre[u][1];
                        ln[0] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(mean_l)));
                        ln[1] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(rstd_l)));
                    }
                    epi_fast_chunk<T, PF, ACT, STATS>(vv, b8, s8, rpre[PF == 1 ? q * 2 + u : 0], ln, optr + (int64_t)roff * e.ldy,
                                                      STATS ? sptr + (int64_t)roff * e.stats_slots * 2 : nullptr);
                }
            }
        };
        if (m0 + TM <= g.M) passes(BoolX<false>{});
        else passes(BoolX<true>{});
        // the strips alias ring slots 0-1: nobody restages them before every wave has read its strips back (raw barrier: the
        // output stores stay in flight; the next tile's counted waits retire them in order)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

template <typename T, int PF, int CFG>
int launch_x2(const X2Args& a, hipStream_t s) {
    static bool attr_set[LECLIP_MAX_DEVICES] = {};
    leclip_set_max_lds(gemm_tn_256x128x32_x2<T, PF, CFG>, LDS_BYTES, attr_set);
    const int slots = 2 * leclip_cu_count();
    const int grid = a.tiles_total < slots ? a.tiles_total : slots;
    hipLaunchKernelGGL((gemm_tn_256x128x32_x2<T, PF, CFG>), dim3(grid), dim3(256), LDS_BYTES, s, a);
    return leclip_check_launch("gemm_tn_256x128x32_x2");
}

template <typename T>
int dispatch_x2(const X2Args& a, hipStream_t s) {
    const EpiParams& e = a.epi;
    const bool gelu = e.act == LECLIP_ACT_QUICKGELU, stats = e.stats_out != nullptr;
    if (!e.res && !e.ln_stats && !stats) return gelu ? launch_x2<T, 0, 1>(a, s) : launch_x2<T, 0, 0>(a, s);
    if (e.res && !gelu && !e.ln_stats) return stats ? launch_x2<T, 1, 2>(a, s) : launch_x2<T, 1, 0>(a, s);
    if (e.ln_stats && !stats && !e.res) return gelu ? launch_x2<T, 2, 1>(a, s) : launch_x2<T, 2, 0>(a, s);
    return LECLIP_E_UNSUPPORTED;
}

}  // namespace

// Shapes / epilogues this kernel takes (the caller falls back to the 256x256 kernel otherwise)
bool leclip_gemm_x2_eligible(int64_t M, int N, int K, const EpiParams& e, int ab_dtype) {
    if (N % TN != 0 || K % TK != 0 || K < 2 * TK) return false;
    if (e.out_dt != ab_dtype || e.rowmap_P || (e.res && e.res_dt != ab_dtype) || (e.res && e.ln_stats)) return false;
    const bool gelu = e.act == LECLIP_ACT_QUICKGELU, stats = e.stats_out != nullptr;
    if (e.res && gelu) return false;
    if (!e.res && stats) return false;
    return ((M + TM - 1) / TM) * (N / TN) >= 384;
}

int leclip_gemm_x2_launch(const void* A, const void* W, int64_t M, int N, int K, int64_t lda, int64_t ldw, const EpiParams& epi,
                          int ab_dtype, hipStream_t s) {
    X2Args a;
    a.A = A; a.W = W; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.epi = epi;
    const int64_t tiles_m = (M + TM - 1) / TM;
    a.tiles_n = N / TN;
    if (tiles_m * a.tiles_n > 0x7fffffff) { leclip_set_error("gemm: too many tiles"); return LECLIP_E_UNSUPPORTED; }
    a.tiles_total = (int)(tiles_m * a.tiles_n);
    return ab_dtype == LECLIP_BF16 ? dispatch_x2<bf16_t>(a, s) : dispatch_x2<f16_t>(a, s);
}
