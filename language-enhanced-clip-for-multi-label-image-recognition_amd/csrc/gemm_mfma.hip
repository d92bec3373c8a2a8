// Y = act(A . W^T + bias) + residual on the gfx950 matrix cores (bf16 / fp16 in, fp32 accumulate).
//
// Kernel "gemm_tn_128x128x64": one 256-thread workgroup (4 waves, 2x2) per 128x128 output tile, each wave a
// 64x64 sub-tile = 4x4 v_mfma_f32_16x16x32 accumulators - the SAME instruction, fed the same K sequence (32 k per MFMA,
// ascending), as the 256x256 kernel (gemm_mfma256.hip), so both families produce bit-identical fp32 accumulators for a
// given (row, column): which family a call takes depends on M, and an image's result must not depend on its batch.  A [M,K] and W [N,K] are both K-contiguous, so both
// operand tiles are [128 rows][64 k] = 128-byte rows.  Tiles are staged HBM -> LDS with 16-byte LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip), double buffered, one barrier per K-step.  The LDS image is
// lane-linear per wave-instruction (8 rows x 128 B), so the bank-conflict swizzle is applied on the per-lane
// SOURCE address and again on the ds_read_b128 address: 16-byte chunk q of row r lives at chunk q ^ ((r>>1)&7)
// (two 128-B rows share one 256-B bank row; the XOR makes each ds_read_b128 16-lane group hit 16 distinct slots).
// Workgroup ids are remapped so that each XCD (blocks b, b+8, ...) walks a contiguous run of tiles, N fastest:
// the A row-panel of a tile row is re-read from that XCD's L2, not from HBM.
#include "leclip_common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;      // A + B
constexpr int EPI_LD = 68;                       // floats per staged epilogue row (64 + 4 pad)
constexpr int EPI_WAVE_BYTES = 64 * EPI_LD * 4;  // one wave's 64x64 fp32 tile
constexpr int LDS_BYTES = 4 * EPI_WAVE_BYTES;    // 68 KiB >= 2 * STAGE_BYTES (64 KiB): 2 workgroups per CU

typedef __attribute__((ext_vector_type(4))) float acc4;
__device__ __forceinline__ acc4 mfma16(bf16x8 a, bf16x8 b, acc4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ acc4 mfma16(f16x8 a, f16x8 b, acc4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

struct GemmArgs {
    const void* A;
    const void* W;
    int64_t M;
    int N, K;
    int64_t lda, ldw;
    EpiParams epi;
    int tiles_n, tiles_total;
};

// bijective XCD-aware remap (guide T1): blocks with equal id % 8 share an XCD; give each a contiguous chunk.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

template <typename T>
__device__ __forceinline__ void stage_tile(const T* __restrict__ g, int64_t ld, int64_t row0, int64_t row_max, int k0,
                                           char* lds_tile, int wave, int lane) {
    // 16 wave-instructions of 1 KiB (8 rows x 128 B) per tile; wave w issues row-blocks w, w+4, w+8, w+12.
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rb = i * 4 + wave;
        const int r = rb * 8 + (lane >> 3);
        const int q = (lane & 7) ^ ((r >> 1) & 7);
        int64_t gr = row0 + r;
        gr = gr < row_max ? gr : row_max - 1;   // clamp: out-of-range rows are computed and discarded
        const T* src = g + gr * ld + k0 + q * 8;
        __builtin_amdgcn_global_load_lds((const void*)src, LDS_PTR(lds_tile + rb * 1024), 16, 0, 0);
    }
}

// PF: what the epilogue prefetches into registers before its first store (same meaning as in gemm_mfma256.hip):
// 0 nothing, 1 the 16-bit residual, 2 the fused-LayerNorm row statistics, 3 generic (loads inside the store loop).
template <typename T, int PF>
__global__ __launch_bounds__(256, 2) void gemm_tn_128x128x64(GemmArgs g) {
    typedef typename VecOf<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int tile = xcd_remap(blockIdx.x, g.tiles_total);
    const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;
    const T* A = (const T*)g.A;
    const T* W = (const T*)g.W;

    acc4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    const int nk = g.K / BK;
    stage_tile<T>(A, g.lda, m0, g.M, 0, smem, wave, lane);
    stage_tile<T>(W, g.ldw, n0, g.N, 0, smem + TILE_BYTES, wave, lane);
    __syncthreads();   // with LDS-DMA outstanding this is s_waitcnt vmcnt(0) + s_barrier

    // per-lane fragment addressing (v_mfma_f32_16x16x32 operand map): row lane&15 of a 16-row tile, 16-byte k-chunk
    // 4*kk + (lane>>4) of the 64-deep K-tile
    const int fr = lane & 15, fc = lane >> 4;
    for (int t = 0; t < nk; ++t) {
        char* cur = smem + (t & 1) * STAGE_BYTES;
        if (t + 1 < nk) {
            char* nxt = smem + ((t + 1) & 1) * STAGE_BYTES;
            stage_tile<T>(A, g.lda, m0, g.M, (t + 1) * BK, nxt, wave, lane);
            stage_tile<T>(W, g.ldw, n0, g.N, (t + 1) * BK, nxt + TILE_BYTES, wave, lane);
        }
        const char* sa = cur;
        const char* sb = cur + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            v8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = wr * 64 + i * 16 + fr;
                af[i] = *(const v8*)(sa + r * 128 + (((4 * kk + fc) ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = wc * 64 + j * 16 + fr;
                bfr[j] = *(const v8*)(sb + r * 128 + (((4 * kk + fc) ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(af[i], bfr[j], acc[i][j]);
        }
        __syncthreads();
    }

    // ---- epilogue.  Accumulator (i,j) register r holds row 16i + 4*(lane>>4) + r, column 16j + (lane&15): a
    // row-major store straight from registers would be 2-byte scalars in 32-byte runs.  Instead each wave parks its
    // 64x64 fp32 tile in its own LDS region (the K-loop buffers are dead after the final barrier; rows padded to 68
    // floats so the column-per-lane ds_write_b32 are conflict-free), then re-reads it row-major, 8 columns per lane:
    // bias / QuickGELU / residual are applied on 8-wide chunks, residual and output move as 16-byte accesses, and
    // every wave-instruction covers 8 full 128-byte output lines.  Wave-local: no workgroup barrier.
    const EpiParams& e = g.epi;
    float* st = (float*)(smem + wave * EPI_WAVE_BYTES);
    const int crow = lane >> 3, ccol = (lane & 7) * 8;
    const int n = n0 + wc * 64 + ccol;
    float b8[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) b8[c] = 0.f;
    if (e.bias) {
        const f32x4 t0 = *(const f32x4*)(e.bias + n), t1 = *(const f32x4*)(e.bias + n + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { b8[c] = t0[c]; b8[4 + c] = t1[c]; }
    }
    float s8[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) s8[c] = 0.f;
    if (e.ln_stats) {
        const f32x4 t0 = *(const f32x4*)(e.ln_colsum + n), t1 = *(const f32x4*)(e.ln_colsum + n + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { s8[c] = t0[c]; s8[4 + c] = t1[c]; }
    }
    // Every global load of the epilogue is issued here, ahead of the first store, waited for once and laundered
    // through empty asm so hipcc does not guard each use with vmcnt(0): loads and stores retire in order, a load
    // inside the store loop would wait for all earlier stores of the wave.
    i32x4 rpre[PF == 1 ? 8 : 1];
    f32x2 lnpre[PF == 2 ? 8 : 1];
    if constexpr (PF == 1) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            int64_t m = m0 + wr * 64 + it * 8 + crow;
            m = m < g.M ? m : g.M - 1;
            const int64_t rrow = e.rowmap_P ? m % e.rowmap_P + 1 : m;
            rpre[it] = *(const i32x4*)((const char*)e.res + (rrow * e.ldr + n) * 2);
        }
    }
    if constexpr (PF == 2) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            int64_t m = m0 + wr * 64 + it * 8 + crow;
            m = m < g.M ? m : g.M - 1;
            lnpre[it] = *(const f32x2*)(e.ln_stats + 2 * m);
        }
    }
    // park the accumulators while those loads are in flight
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                st[(i * 16 + 4 * fc + r) * EPI_LD + j * 16 + fr] = acc[i][j][r];
    if constexpr (PF != 3) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int c = 0; c < 8; ++c) { asm volatile("" : "+v"(b8[c])); asm volatile("" : "+v"(s8[c])); }
        if constexpr (PF == 1) {
#pragma unroll
            for (int it = 0; it < 8; ++it) asm volatile("" : "+v"(rpre[it]));
        }
        if constexpr (PF == 2) {
#pragma unroll
            for (int it = 0; it < 8; ++it) asm volatile("" : "+v"(lnpre[it]));
        }
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + crow;
        const int64_t m = m0 + wr * 64 + row;
        const f32x4 v0 = *(const f32x4*)(st + row * EPI_LD + ccol), v1 = *(const f32x4*)(st + row * EPI_LD + ccol + 4);
        if (m >= g.M) continue;
        float v[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) { v[c] = v0[c]; v[4 + c] = v1[c]; }
        epi_chunk8<PF>(e, m, n, v, b8, s8, rpre[PF == 1 ? it : 0], lnpre[PF == 2 ? it : 0]);
    }
}

template <typename T, int PF>
int launch_mfma_pf(const GemmArgs& a, hipStream_t s) {
    static bool attr_set[LECLIP_MAX_DEVICES] = {};
    leclip_set_max_lds(gemm_tn_128x128x64<T, PF>, LDS_BYTES, attr_set);
    hipLaunchKernelGGL((gemm_tn_128x128x64<T, PF>), dim3(a.tiles_total), dim3(256), LDS_BYTES, s, a);
    return leclip_check_launch("gemm_tn_128x128x64");
}

template <typename T>
int launch_mfma(const GemmArgs& a, hipStream_t s) {
    const bool res16 = a.epi.res && a.epi.res_dt != LECLIP_F32;
    if (!a.epi.res && !a.epi.ln_stats) return launch_mfma_pf<T, 0>(a, s);
    if (res16 && !a.epi.ln_stats) return launch_mfma_pf<T, 1>(a, s);
    if (a.epi.ln_stats && !a.epi.res) return launch_mfma_pf<T, 2>(a, s);
    return launch_mfma_pf<T, 3>(a, s);
}

}  // namespace

int leclip_gemm_f32_launch(const void* A, const void* W, int64_t M, int N, int K, int64_t lda, int64_t ldw,
                           const EpiParams& epi, hipStream_t s);
extern "C" int leclip_gemm_ln_partials_fwd(const void* A, const void* W, const float* bias, const float* ln_stats, const float* ln_partials,
                                           int ln_slots, float* ln_stats_ws, float ln_eps, const float* ln_colsum, const void* residual,
                                           void* Y, float* stats_out, int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldr,
                                           int64_t ldy, leclip_act act, leclip_dtype ab_dtype, leclip_dtype res_dtype,
                                           leclip_dtype y_dtype, void* stream);
bool leclip_gemm256_eligible(int64_t M, int N, int K);
int leclip_gemm256_cus();
int launch_128(const void* A, const void* W, int64_t M, int N, int K, int64_t lda, int64_t ldw, const EpiParams& epi,
               int ab_dtype, hipStream_t s);
int leclip_gemm256_launch(const void* A, const void* W, int64_t M, int N, int K, int64_t lda, int64_t ldw,
                          const EpiParams& epi, int ab_dtype, hipStream_t s);
bool leclip_gemm384_eligible(int64_t M, int N, int K, int64_t lda, int64_t ldw, const EpiParams& e, int ab_dtype);
bool leclip_gemm384_shape(int64_t M, int N, int K);
int leclip_gemm384_launch(const void* A, const void* W, int64_t M, int N, int K, int64_t lda, int64_t ldw, const EpiParams& epi, int ab_dtype,
                          hipStream_t s);

int leclip_gemm_dispatch(const void* A, const void* W, int64_t M, int N, int K, int64_t lda, int64_t ldw,
                         const EpiParams& epi, int ab_dtype, hipStream_t s) {
    if (ab_dtype == LECLIP_F32) return leclip_gemm_f32_launch(A, W, M, N, K, lda, ldw, epi, s);
    if (N % BN != 0 || K % BK != 0) {
        leclip_set_error("gemm: N=%d must be a multiple of %d and K=%d a multiple of %d for 16-bit operands", N, BN, K, BK);
        return LECLIP_E_UNSUPPORTED;
    }
    if ((lda % 8) || (ldw % 8) || ((uintptr_t)A & 15) || ((uintptr_t)W & 15)) {
        leclip_set_error("gemm: A/W must be 16-byte aligned with leading dimensions that are multiples of 8 elements");
        return LECLIP_E_INVALID;
    }
    {
        const int oa = epi.out_dt == LECLIP_F32 ? 4 : 8, ra = epi.res_dt == LECLIP_F32 ? 4 : 8;
        if ((epi.ldy % oa) || ((uintptr_t)epi.out & 15) || (epi.res && ((epi.ldr % ra) || ((uintptr_t)epi.res & 15))) ||
            (epi.bias && ((uintptr_t)epi.bias & 15))) {
            leclip_set_error("gemm: Y / residual / bias must be 16-byte aligned with 16-byte-multiple leading dimensions");
            return LECLIP_E_INVALID;
        }
    }
    // One kernel family per call, chosen from (M, N, K) alone and covering every row: an image's result must not depend on
    // where its rows sit in the batch (round 1 sent the trailing tile rows of a large batch to the 128x128 kernel, whose
    // MFMA shape sums K in a different order - sharded logits then differed from unsharded ones in the last bits).
    // (three families since round 5, all on v_mfma_f32_16x16x32 with ascending K and one epilogue arithmetic: the same bits from each)
    if (leclip_gemm384_eligible(M, N, K, lda, ldw, epi, ab_dtype)) {
#ifdef LECLIP_GEMM_TAIL_SPLIT   // A/B builds only (make variant VSRC=gemm_mfma DEFS=-DLECLIP_GEMM_TAIL_SPLIT), measured and NOT adopted, profiles/r05_schedule_experiments.txt (9):
        // the row blocks that fill whole rounds of the persistent launch stay with it, the remaining rows go to a launch of 128 x 128 tiles behind it
        // (c_fc of a 128-image part: 792 tiles = 3.09 rounds).  Same bits (the families are bit-identical); +1.6 % on the GEMMs of a one-part step,
        // -0.6 % on the two-part step, whose other part already fills the last round's idle CUs.
        const int n_cu = leclip_cu_count(), tn = N / 256;
        const int64_t rb = (M + 383) / 384, tiles = rb * tn, full = tiles / n_cu, rem = tiles % n_cu;
        if (leclip_gemm_family() < 0 && !epi.stats_out && full >= 1 && rem > 0 && rem * 4 <= n_cu) {
            const int64_t rb_main = full * n_cu / tn, m_main = rb_main * 384, m_tail = M - m_main;
            if (rb_main >= 1 && m_tail > 0) {
                int rc = leclip_gemm384_launch(A, W, m_main, N, K, lda, ldw, epi, ab_dtype, s);
                if (rc) return rc;
                EpiParams t = epi;
                t.out = (char*)epi.out + m_main * epi.ldy * 2;
                if (epi.res) t.res = (const char*)epi.res + m_main * epi.ldr * 2;
                if (epi.ln_stats) t.ln_stats = epi.ln_stats + 2 * m_main;
                return launch_128((const char*)A + m_main * lda * 2, W, m_tail, N, K, lda, ldw, t, ab_dtype, s);
            }
        }
#endif
        return leclip_gemm384_launch(A, W, M, N, K, lda, ldw, epi, ab_dtype, s);
    }
    if (leclip_gemm256_eligible(M, N, K)) return leclip_gemm256_launch(A, W, M, N, K, lda, ldw, epi, ab_dtype, s);
    return launch_128(A, W, M, N, K, lda, ldw, epi, ab_dtype, s);
}

int launch_128(const void* A, const void* W, int64_t M, int N, int K, int64_t lda, int64_t ldw, const EpiParams& epi,
               int ab_dtype, hipStream_t s) {
    GemmArgs a;
    a.A = A; a.W = W; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.epi = epi;
    const int64_t tiles_m = (M + BM - 1) / BM;
    a.tiles_n = N / BN;
    if (tiles_m * a.tiles_n > 0x7fffffff) { leclip_set_error("gemm: too many tiles"); return LECLIP_E_UNSUPPORTED; }
    a.tiles_total = (int)(tiles_m * a.tiles_n);
    return ab_dtype == LECLIP_BF16 ? launch_mfma<bf16_t>(a, s) : launch_mfma<f16_t>(a, s);
}

extern "C" const char* leclip_gemm_kernel_name(int64_t M, int N, int K, leclip_dtype ab_dtype) {
    (void)M;
    if (ab_dtype == LECLIP_F32) return (N % 64 == 0 && K % 32 == 0) ? "gemm_f32_64x64x32" : "unsupported";
    if (N % BN != 0 || K % BK != 0) return "unsupported";
    // (the 384 x 256 kernel takes the residual / fused-LayerNorm / plain 16-bit epilogues of a shape it is eligible for; fp32 output, row remap and
    // the im2col gather stay with the 256 x 256 kernel)
    if (leclip_gemm384_shape(M, N, K)) return "gemm_tn_384x256x32_pp";
    return leclip_gemm256_eligible(M, N, K) ? "gemm_tn_256x256x64_pp" : "gemm_tn_128x128x64";
}

extern "C" int leclip_gemm_bias_act_res_fwd(const void* A, const void* W, const float* bias, const void* residual,
                                            void* Y, int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldr,
                                            int64_t ldy, leclip_act act, leclip_dtype ab_dtype, leclip_dtype res_dtype,
                                            leclip_dtype y_dtype, void* stream) {
    if (!A || !W || !Y || M <= 0 || N <= 0 || K <= 0 || lda < K || ldw < K || ldy < N || (residual && ldr < N)) {
        leclip_set_error("gemm: null pointer or inconsistent sizes (M=%lld N=%d K=%d)", (long long)M, N, K);
        return LECLIP_E_INVALID;
    }
    if (!dtype_ok(ab_dtype) || !dtype_ok(y_dtype) || (residual && !dtype_ok(res_dtype)) ||
        (act != LECLIP_ACT_NONE && act != LECLIP_ACT_QUICKGELU)) {
        leclip_set_error("gemm: bad dtype / activation enum");
        return LECLIP_E_INVALID;
    }
    EpiParams e;
    e.bias = bias; e.res = residual; e.out = Y; e.ldr = ldr; e.ldy = ldy;
    e.res_dt = res_dtype; e.out_dt = y_dtype; e.act = act; e.rowmap_P = 0;
    e.ln_stats = nullptr; e.ln_colsum = nullptr; e.ln_partials = nullptr; e.ln_slots = 0; e.ln_eps = 0.f; e.stats_out = nullptr; e.stats_slots = 0; e.stats_rows = 0;
    return leclip_gemm_dispatch(A, W, M, N, K, lda, ldw, e, ab_dtype, (hipStream_t)stream);
}

int leclip_ln_stats_finalize_launch(const float* partials, float* stats, int64_t rows, int slots, int dim, float eps, hipStream_t s);
bool leclip_gemm384_merges(int64_t M, int N);

// Residual GEMM that also FINISHES the LayerNorm statistics of its output rows (round 5): Y = A W^T + bias + residual and stats[m] = (mean, rstd)
// of row m of Y.  On the 384 x 256 kernel the row block's last-arriving workgroup merges the block partials in the launch itself; every other
// dispatch writes the partials and this entry point launches the merge kernel behind the GEMM - the same bits either way (ln_merge_partials).
extern "C" int leclip_gemm_res_stats_fwd(const void* A, const void* W, const float* bias, const void* residual, void* Y, float* partials_ws,
                                         float* stats, unsigned* tickets_ws, float ln_eps, int64_t M, int N, int K, int64_t lda, int64_t ldw,
                                         int64_t ldr, int64_t ldy, leclip_dtype ab_dtype, leclip_dtype res_dtype, leclip_dtype y_dtype, void* stream) {
    if (!A || !W || !Y || !residual || !partials_ws || !stats || M <= 0 || N <= 0 || K <= 0 || lda < K || ldw < K || ldy < N || ldr < N || N % 64 != 0) {
        leclip_set_error("gemm_res_stats: null pointer or inconsistent sizes (M=%lld N=%d K=%d)", (long long)M, N, K);
        return LECLIP_E_INVALID;
    }
    if (ab_dtype == LECLIP_F32 || !dtype_ok(ab_dtype) || !dtype_ok(y_dtype) || !dtype_ok(res_dtype)) {
        leclip_set_error("gemm_res_stats: 16-bit operands only");
        return LECLIP_E_UNSUPPORTED;
    }
    if (((uintptr_t)partials_ws & 15) || ((uintptr_t)stats & 7) || ((uintptr_t)tickets_ws & 3)) {
        leclip_set_error("gemm_res_stats: partials_ws 16-byte, stats 8-byte, tickets_ws 4-byte aligned");
        return LECLIP_E_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    EpiParams e;
    e.bias = bias; e.res = residual; e.out = Y; e.ldr = ldr; e.ldy = ldy;
    e.res_dt = res_dtype; e.out_dt = y_dtype; e.act = LECLIP_ACT_NONE; e.rowmap_P = 0;
    e.ln_stats = nullptr; e.ln_colsum = nullptr; e.ln_partials = nullptr; e.ln_slots = 0; e.ln_eps = 0.f;
    e.stats_out = partials_ws; e.stats_slots = N / 64; e.stats_rows = M;
    const bool merged = tickets_ws && leclip_gemm384_merges(M, N) && leclip_gemm384_eligible(M, N, K, lda, ldw, e, ab_dtype);
    if (merged) { e.stats_merged = stats; e.stats_tickets = tickets_ws; e.stats_eps = ln_eps; }
    const int rc = leclip_gemm_dispatch(A, W, M, N, K, lda, ldw, e, ab_dtype, s);
    if (rc || merged) return rc;
    return leclip_ln_stats_finalize_launch(partials_ws, stats, M, N / 64, N, ln_eps, s);
}

extern "C" int leclip_gemm_ln_fused_fwd(const void* A, const void* W, const float* bias, const float* ln_stats,
                                        const float* ln_colsum, const void* residual, void* Y, float* stats_out, int64_t M,
                                        int N, int K, int64_t lda, int64_t ldw, int64_t ldr, int64_t ldy, leclip_act act,
                                        leclip_dtype ab_dtype, leclip_dtype res_dtype, leclip_dtype y_dtype, void* stream) {
    return leclip_gemm_ln_partials_fwd(A, W, bias, ln_stats, nullptr, 0, nullptr, 0.f, ln_colsum, residual, Y, stats_out, M, N, K, lda, ldw,
                                       ldr, ldy, act, ab_dtype, res_dtype, y_dtype, stream);
}

extern "C" int leclip_gemm_ln_partials_fwd(const void* A, const void* W, const float* bias, const float* ln_stats, const float* ln_partials,
                                           int ln_slots, float* ln_stats_ws, float ln_eps, const float* ln_colsum, const void* residual,
                                           void* Y, float* stats_out, int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldr,
                                           int64_t ldy, leclip_act act, leclip_dtype ab_dtype, leclip_dtype res_dtype,
                                           leclip_dtype y_dtype, void* stream) {
    if (!A || !W || !Y || M <= 0 || N <= 0 || K <= 0 || lda < K || ldw < K || ldy < N || (residual && ldr < N)) {
        leclip_set_error("gemm_ln_fused: null pointer or inconsistent sizes (M=%lld N=%d K=%d)", (long long)M, N, K);
        return LECLIP_E_INVALID;
    }
    if (ab_dtype == LECLIP_F32 || !dtype_ok(ab_dtype) || !dtype_ok(y_dtype) || (residual && !dtype_ok(res_dtype)) ||
        (act != LECLIP_ACT_NONE && act != LECLIP_ACT_QUICKGELU)) {
        leclip_set_error("gemm_ln_fused: 16-bit operands only (the fp32 parity path keeps LayerNorm as its own kernel)");
        return LECLIP_E_UNSUPPORTED;
    }
    if (ln_stats && ln_partials) { leclip_set_error("gemm_ln_fused: give ln_stats OR ln_partials, not both"); return LECLIP_E_INVALID; }
    const bool ln = ln_stats != nullptr || ln_partials != nullptr;
    if (ln != (ln_colsum != nullptr) || (ln && !bias) || (ln_colsum && ((uintptr_t)ln_colsum & 15)) || (stats_out && ((uintptr_t)stats_out & 7)) ||
        (ln_partials && (ln_slots <= 0 || K % ln_slots != 0 || ((uintptr_t)ln_partials & 7)))) {
        leclip_set_error("gemm_ln_fused: LayerNorm statistics, ln_colsum and bias go together; ln_colsum 16-byte, stats_out / ln_partials 8-byte aligned");
        return LECLIP_E_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    EpiParams e;
    e.bias = bias; e.res = residual; e.out = Y; e.ldr = ldr; e.ldy = ldy;
    e.res_dt = res_dtype; e.out_dt = y_dtype; e.act = act; e.rowmap_P = 0;
    e.ln_stats = ln_stats; e.ln_colsum = ln_colsum; e.stats_out = stats_out; e.stats_slots = N / 64; e.stats_rows = M;
    e.ln_partials = nullptr; e.ln_slots = 0; e.ln_eps = ln_eps;
    if (ln_partials) {
        // The merge (Chan's parallel-variance update, ln_merge_partials) runs as its own small launch into ln_stats_ws.  Merging
        // inside the consuming 256x256 kernel was built and measured in round 2: the row's 12 partial pairs have to be held in
        // registers across the last K-tile, the specialised LayerNorm epilogue kernels sit at 251 of 256 VGPRs, the spill cost
        // 11 % end to end (profiles/r02_ab_ln_merge_in_kernel.txt) against the 1.5 % the 23 launches cost.
        if (!ln_stats_ws) { leclip_set_error("gemm_ln_fused: ln_partials needs the [M, 2] statistics workspace (ln_stats_ws)"); return LECLIP_E_INVALID; }
        const int rc = leclip_ln_stats_finalize_launch(ln_partials, ln_stats_ws, M, ln_slots, K, ln_eps, s);
        if (rc) return rc;
        e.ln_stats = ln_stats_ws;
    }
    return leclip_gemm_dispatch(A, W, M, N, K, lda, ldw, e, ab_dtype, s);
}
