// Shared device helpers for the gfx950 kernels (wave64; GEMMs on v_mfma_f32_16x16x32, attention on 32x32x16; fp32 accumulate).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/leclip_hip.h"

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// Diagnostic library only (make diag): per-workgroup begin / end times on the 100 MHz real-time counter, appended to a device log
// (leclip_diag_set_wglog) - the occupancy timeline of overlapping launches (profiles/two_part_timeline.py).  The product build
// compiles none of it.
#ifdef LECLIP_DIAG
struct WgLog { unsigned long long* buf; unsigned cap; unsigned seq; };   // buf[0] = entry count, entries of 6 u64 from buf[2]
extern unsigned long long* g_leclip_wglog;
extern unsigned g_leclip_wglog_cap, g_leclip_wglog_seq;
// c0: s_memtime (shader clock) taken next to t0: (c1 - c0) / (t1 - t0) x 100 MHz is the clock the CU held while the workgroup ran
__device__ __forceinline__ void wglog_end(const WgLog& w, unsigned tag, unsigned long long t0, unsigned long long c0) {
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long i = atomicAdd(w.buf, 1ull);
    if (i < w.cap) {
        unsigned long long* e = w.buf + 2 + 6 * i;
        unsigned hw, xcc;   // where the workgroup ran: HW_ID (se / sh / cu fields) and XCC_ID
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        e[0] = ((unsigned long long)tag << 32) | blockIdx.x; e[1] = t0; e[2] = t1;
        e[3] = w.seq | ((unsigned long long)(hw & 0xffff) << 32) | ((unsigned long long)(xcc & 0xf) << 48);
        e[4] = c0; e[5] = c1;
    }
}
#endif
int leclip_walk_order();   // capi.hip: the calling thread's walk-order hint (-1 default, 0 ascending, 1 descending)
int leclip_gemm_family();  // capi.hip: the calling thread's GEMM kernel-family override (-1 = the library's rate heuristic; 128 / 256 / 384)
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <typename T> struct VecOf;
template <> struct VecOf<bf16_t> { typedef bf16x8 v8; typedef bf16x4 v4; };
template <> struct VecOf<f16_t> { typedef f16x8 v8; typedef f16x4 v4; };

__device__ __forceinline__ f32x16 mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_32x32x16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// ds_read_b64_tr_b16: per 16-lane group, a 4-row x 16-column block of 16-bit elements, delivered column-major.
__device__ __forceinline__ bf16x4 lds_read_tr16(const bf16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
}
__device__ __forceinline__ f16x4 lds_read_tr16(const f16_t* p) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    return __builtin_bit_cast(f16x4, r);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// generic scalar load / store by runtime dtype (wave-uniform branch)
__device__ __forceinline__ float load_elem(const void* p, int dt, int64_t i) {
    if (dt == LECLIP_F32) return ((const float*)p)[i];
    if (dt == LECLIP_F16) return (float)((const f16_t*)p)[i];
    return (float)((const bf16_t*)p)[i];
}
__device__ __forceinline__ void store_elem(void* p, int dt, int64_t i, float v) {
    if (dt == LECLIP_F32) ((float*)p)[i] = v;
    else if (dt == LECLIP_F16) ((f16_t*)p)[i] = (f16_t)v;
    else ((bf16_t*)p)[i] = (bf16_t)v;
}

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }

static inline int dtype_size(int dt) { return dt == LECLIP_F32 ? 4 : 2; }
static inline bool dtype_ok(int dt) { return dt == LECLIP_F32 || dt == LECLIP_F16 || dt == LECLIP_BF16; }

// host-side error plumbing (capi.hip)
void leclip_set_error(const char* fmt, ...);
int leclip_check_launch(const char* what);

// Per-device one-time function attributes (max dynamic LDS) and the CU count: hipFuncSetAttribute applies to the current
// device only, so the "already done" state is kept per device ordinal, not per process.
constexpr int LECLIP_MAX_DEVICES = 64;
static inline int leclip_device_ordinal() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev >= 0 && dev < LECLIP_MAX_DEVICES ? dev : 0;
}
template <typename F>
static inline void leclip_set_max_lds(F* kernel, int bytes, bool (&done)[LECLIP_MAX_DEVICES]) {
    const int dev = leclip_device_ordinal();
    if (!done[dev]) {
        (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        done[dev] = true;
    }
}
int leclip_cu_count();   // CUs of the current device (gemm_mfma256.hip)

// Epilogue description shared by the GEMM kernels.
struct EpiParams {
    const float* bias;   // [N] or null
    const void* res;     // residual or null
    void* out;
    int64_t ldr, ldy;
    int res_dt, out_dt, act;
    int rowmap_P;        // patch-embed mode: >0 => out row = m + m/P + 1, residual row = m % P + 1
    // LayerNorm fused on the A side (A holds the un-normalised rows, W has gamma folded in):
    //   v = rstd[m] * (acc - mean[m] * ln_colsum[n]) + bias[n]      with bias[n] = sum_k beta[k] W[n,k] + b[n]
    const float* ln_stats;   // [M][2] = (mean, rstd) per row, or null
    const float* ln_colsum;  // [N]    = sum_k (gamma[k] W[n,k])
    const float* ln_partials;   // (reserved: block partials are merged by the finalize kernel before dispatch, see gemm_mfma.hip)
    int ln_slots;
    float ln_eps;
    // Row statistics of the OUTPUT for the next LayerNorm: (sum, sum of squared deviations from the block mean) of the
    // rounded outputs per (row, 64-column block), written - not accumulated - so the result is deterministic and needs no
    // zeroing; leclip_ln_stats_finalize_fwd merges the blocks (parallel-variance update, no E[x^2] - mean^2 cancellation).
    // Layout SLOT-MAJOR, [stats_slots][stats_rows][2] (round 3): the 8 or 16 rows a wave finishes together are 64 / 128 contiguous
    // bytes of one slot's column, one full-line store per epilogue pass - row-major [M][slots][2] made every pair a scattered 8-byte
    // write at a 96-byte stride, 128 partial-line requests per wave and tile (14 % of the residual GEMMs).
    float* stats_out;        // [stats_slots][stats_rows][2] or null
    int stats_slots;
    int64_t stats_rows;      // rows of the partials tensor (= M of the producing GEMM)
    // Round 5, in-producer merge (gemm_tn_384x256x32_pp only): the workgroup whose ticket on a 384-row block's counter comes last merges the
    // block's partials (ln_merge_partials: the finalize kernel's arithmetic, same bits) into stats_merged [M][2] = (mean, rstd) - no merge
    // launch between producer and consumer.  stats_tickets: one counter per row block, zero on entry, zero again when the launch has ended.
    float* stats_merged = nullptr;
    unsigned* stats_tickets = nullptr;
    float stats_eps = 0.f;
};

typedef __attribute__((ext_vector_type(4))) int i32x4;

// Merge a row's block partials (sum, M2 about the block mean; two blocks per f32x4) into (mean, rstd): Chan et al.'s
// parallel-variance update in a fixed block order.  ONE definition for the finalize kernel and for the GEMM kernels that
// merge in place, so that both produce the same bits (batch invariance across kernel families).
constexpr int LN_MERGE_MAXV = 8;   // <= 16 blocks of 64 columns: dim <= 1024
__device__ __forceinline__ f32x2 ln_merge_partials(const f32x4 (&v)[LN_MERGE_MAXV], int slots, int dim, float eps) {
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MERGE_MAXV; ++i)
        if (2 * i < slots) { s1 += v[i][0]; s1 += v[i][2]; }
    const float mean = s1 / (float)dim;
    const float bn = (float)(dim / slots);
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MERGE_MAXV; ++i)
        if (2 * i < slots) {
            float dlt = v[i][0] / bn - mean;
            m2 += fmaf(bn * dlt, dlt, v[i][1]);
            dlt = v[i][2] / bn - mean;
            m2 += fmaf(bn * dlt, dlt, v[i][3]);
        }
    f32x2 o;
    o[0] = mean;
    o[1] = rsqrtf(m2 / (float)dim + eps);
    return o;
}

// Shared 8-wide epilogue step of the 16-bit GEMM kernels: lane holds columns n..n+7 of output row m (8 lanes per
// row: lanes with equal lane>>3).  b8 = bias chunk, s8 = ln_colsum chunk.
// MODE (compile time): 0 = no residual, no fused LayerNorm; 1 = 16-bit residual prefetched in res_val; 2 = fused
// LayerNorm with (mean, rstd) prefetched in ln_val; 3 = generic, everything decided and loaded at run time.  The
// specialised modes keep every global LOAD out of the store loop (vmcnt retires loads and stores in order).
template <int MODE>
__device__ __forceinline__ void epi_chunk8(const EpiParams& e, int64_t m, int n, float (&v)[8], const float (&b8)[8],
                                           const float (&s8)[8], i32x4 res_val, f32x2 ln_val) {
    constexpr bool have_res = MODE == 1, have_ln = MODE == 2;
    // have_res / have_ln: the caller prefetched the 16-bit residual chunk / the row's (mean, rstd) into registers
    // (passed by value: taking the address of a register array would push it to scratch memory)
    if (MODE == 2 || (MODE == 3 && e.ln_stats)) {
        const f32x2 st = have_ln ? ln_val : *(const f32x2*)(e.ln_stats + 2 * m);
        const float mean = st[0], rstd = st[1];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = fmaf(rstd, fmaf(-mean, s8[c], v[c]), b8[c]);
    } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] += b8[c];
    }
    if (e.act == LECLIP_ACT_QUICKGELU) {
        // x * sigmoid(1.702 x) = x / (1 + 2^(-1.702 log2(e) x)): v_exp_f32 + v_rcp_f32   (clip/model.py:202-204)
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = v[c] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930157f * v[c]));
    }
    int64_t orow = m, rrow = m;
    if (e.rowmap_P) { orow = m + m / e.rowmap_P + 1; rrow = m % e.rowmap_P + 1; }
    // Residual in the OUTPUT's 16-bit type (the residual stream of the 16-bit modes): the branch value is rounded to that type before the
    // residual is added - the reference's own order in half precision (clip/model.py:225-228: F.linear returns a 16-bit tensor, `x + ...`
    // is a second 16-bit operation) and, since round 4, what lets the 256 x 256 kernel stage the branch value through LDS in 16 bits
    // (gemm_mfma256.hip, T16 epilogue) with results that stay bit-identical to this code.  fp32 outputs / residuals: no extra rounding.
    if ((MODE == 1 || (MODE == 3 && e.res)) && e.out_dt != LECLIP_F32 && e.res_dt == e.out_dt) {
        if (e.out_dt == LECLIP_BF16) {
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = (float)(bf16_t)v[c];
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = (float)(f16_t)v[c];
        }
    }
    if (MODE == 1 || (MODE == 3 && e.res)) {
        if (MODE == 3 && e.res_dt == LECLIP_F32) {
            const float* rp = (const float*)e.res + rrow * e.ldr + n;
            const f32x4 r0 = *(const f32x4*)rp, r1 = *(const f32x4*)(rp + 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) { v[c] += r0[c]; v[4 + c] += r1[c]; }
        } else if (e.res_dt == LECLIP_BF16) {
            const bf16x8 r8 = have_res ? __builtin_bit_cast(bf16x8, res_val) : *(const bf16x8*)((const bf16_t*)e.res + rrow * e.ldr + n);
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] += (float)r8[c];
        } else {
            const f16x8 r8 = have_res ? __builtin_bit_cast(f16x8, res_val) : *(const f16x8*)((const f16_t*)e.res + rrow * e.ldr + n);
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] += (float)r8[c];
        }
    }
    float rv[8];   // the values as stored (rounded to the output type): what the next LayerNorm will see
    if (e.out_dt == LECLIP_F32) {
        float* op = (float*)e.out + orow * e.ldy + n;
        f32x4 o0, o1;
#pragma unroll
        for (int c = 0; c < 4; ++c) { o0[c] = v[c]; o1[c] = v[4 + c]; }
        *(f32x4*)op = o0;
        *(f32x4*)(op + 4) = o1;
#pragma unroll
        for (int c = 0; c < 8; ++c) rv[c] = v[c];
    } else if (e.out_dt == LECLIP_BF16) {
        bf16x8 o8;
#pragma unroll
        for (int c = 0; c < 8; ++c) o8[c] = (bf16_t)v[c];
        *(bf16x8*)((bf16_t*)e.out + orow * e.ldy + n) = o8;
#pragma unroll
        for (int c = 0; c < 8; ++c) rv[c] = (float)o8[c];
    } else {
        f16x8 o8;
#pragma unroll
        for (int c = 0; c < 8; ++c) o8[c] = (f16_t)v[c];
        *(f16x8*)((f16_t*)e.out + orow * e.ldy + n) = o8;
#pragma unroll
        for (int c = 0; c < 8; ++c) rv[c] = (float)o8[c];
    }
    if (e.stats_out) {
        // (sum, M2 about the block mean) of the 64-column block; the 8 lanes of a row (consecutive lanes) hold its 64 columns
        // of this wave: butterfly (same additions, in the same order, as the DPP form of gemm_mfma256.hip), lane 0 writes
        float s1 = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) s1 += rv[c];
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) s1 += __shfl_xor(s1, o);
        const float mb = s1 * (1.0f / 64.0f);
        float m2 = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) { const float dlt = rv[c] - mb; m2 = fmaf(dlt, dlt, m2); }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) m2 += __shfl_xor(m2, o);
        if ((threadIdx.x & 7) == 0) {
            f32x2 w;
            w[0] = s1; w[1] = m2;
            *(f32x2*)(e.stats_out + ((int64_t)(n >> 6) * e.stats_rows + orow) * 2) = w;
        }
    }
}

// ---- compile-time-specialised epilogue step shared by the 256x256 and the 256x128 kernels
// sum over the 8 consecutive lanes that hold one output row (DPP: no LDS round trip)
__device__ __forceinline__ float row8_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));   // row_half_mirror
    return x;
}

// (sum, M2 about the block mean) of one output row's 64-column block from the 8 rounded values each of the row's 8 consecutive lanes
// holds - the LayerNorm partials a producing GEMM emits.  M2 about the BLOCK mean, so that a large common offset of the row never enters a
// difference of two large sums (the consumer merges the blocks with the parallel-variance update).  Same additions, in the same order,
// as the butterfly in epi_chunk8: the two GEMM families emit identical partials.  Every one of the 8 lanes returns the pair.
template <typename V8>
__device__ __forceinline__ f32x2 row_block_stats(const V8& o8) {
    float s1 = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) s1 += (float)o8[c];
    s1 = row8_sum(s1);
    const float mb = s1 * (1.0f / 64.0f);
    float m2 = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) { const float dlt = (float)o8[c] - mb; m2 = fmaf(dlt, dlt, m2); }
    m2 = row8_sum(m2);
    f32x2 st;
    st[0] = s1;
    st[1] = m2;
    return st;
}

template <bool PRECISE>
__device__ __forceinline__ float epi_apply(const EpiParams& p, int64_t m, int n, float v) {
    if (p.bias) v += p.bias[n];
    if (p.act == LECLIP_ACT_QUICKGELU) {
        // x * sigmoid(1.702 x)  (clip/model.py:202-204)
        float e = PRECISE ? expf(-1.702f * v) : __expf(-1.702f * v);
        v = v / (1.0f + e);
    }
    int64_t rrow = m;
    if (p.rowmap_P) rrow = m % p.rowmap_P + 1;
    if (p.res) v += load_elem(p.res, p.res_dt, rrow * p.ldr + n);
    return v;
}
__device__ __forceinline__ int64_t epi_out_row(const EpiParams& p, int64_t m) {
    return p.rowmap_P ? m + m / p.rowmap_P + 1 : m;
}
