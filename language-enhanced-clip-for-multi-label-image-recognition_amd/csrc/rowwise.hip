// HBM-bound row-wise kernels of the scoring path: LayerNorm, gather+LayerNorm+projection, L2-normalise + cosine
// logits, token embedding / prompt assembly, patch extraction.  One wave64 per row where a row reduction is needed
// (statistics by __shfl_xor butterflies, fp32), 8- or 16-byte vector accesses, no LDS except for the projection.
#include "leclip_common.h"

namespace {

// ------------------------------------------------------------------------------------------------ LayerNorm
// One wave per row, 4 rows per workgroup; each lane owns NV vectors of 4 consecutive elements (dim = 256 * NV').
template <typename TI>
__device__ __forceinline__ f32x4 load4(const TI* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 load4<bf16_t>(const bf16_t* p) {
    const bf16x4 v = *(const bf16x4*)p;
    f32x4 r; for (int i = 0; i < 4; ++i) r[i] = (float)v[i]; return r;
}
template <> __device__ __forceinline__ f32x4 load4<f16_t>(const f16_t* p) {
    const f16x4 v = *(const f16x4*)p;
    f32x4 r; for (int i = 0; i < 4; ++i) r[i] = (float)v[i]; return r;
}
template <typename TO>
__device__ __forceinline__ void store4(TO* p, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, f32x4 v) {
    bf16x4 r; for (int i = 0; i < 4; ++i) r[i] = (bf16_t)v[i]; *(bf16x4*)p = r;
}
template <> __device__ __forceinline__ void store4<f16_t>(f16_t* p, f32x4 v) {
    f16x4 r; for (int i = 0; i < 4; ++i) r[i] = (f16_t)v[i]; *(f16x4*)p = r;
}

constexpr int LN_MAXV = 16;  // dim <= 4096

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, TO* __restrict__ y,
                                                        int64_t rows, int dim, int64_t ldx, int64_t ldy, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const TI* xr = x + row * ldx;
    const int nv = dim >> 8;            // full 256-element sweeps
    const int tail = dim & 255;         // remaining elements (multiple of 64): lanes < tail/4 take one more vector
    f32x4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        if (i < nv || (i == nv && lane * 4 < tail)) {
            v[i] = load4<TI>(xr + i * 256 + lane * 4);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = wave_sum(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        if (i < nv || (i == nv && lane * 4 < tail)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q = fmaf(d, d, q); }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)dim + eps);
    TO* yr = y + row * ldy;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        if (i < nv || (i == nv && lane * 4 < tail)) {
            const int c = i * 256 + lane * 4;
            const f32x4 g = *(const f32x4*)(gamma + c), b = *(const f32x4*)(beta + c);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
            store4<TO>(yr + c, o);
        }
    }
}

// Patch-embedding finish fused with ln_pre (clip/model.py:262-265): row (b, t) = LayerNorm(t == 0 ? class_emb : conv_out[b, t-1]) + pos[t])
// - the class-token concat, the positional add and ln_pre in ONE pass over the residual stream (the GEMM before it then runs
// the plain 16-bit epilogue instead of the generic one with an fp32 residual and a row remap).  One wave per row.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void embed_ln_pre_kernel(const TI* __restrict__ conv_out, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, TO* __restrict__ y, float* __restrict__ stats_out,
                                                           int64_t rows, int T, int dim, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t b = row / T;
    const int t = (int)(row - b * T);
    const TI* xr = conv_out + (b * (T - 1) + (t > 0 ? t - 1 : 0)) * dim;
    const float* pr = pos + (int64_t)t * dim;
    const int nv = dim >> 8, tail = dim & 255;
    f32x4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        if (i < nv || (i == nv && lane * 4 < tail)) {
            const int c = i * 256 + lane * 4;
            const f32x4 p4 = *(const f32x4*)(pr + c);
            const f32x4 a4 = t > 0 ? load4<TI>(xr + c) : *(const f32x4*)(cls + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][e] = a4[e] + p4[e];
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = wave_sum(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        if (i < nv || (i == nv && lane * 4 < tail)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q = fmaf(d, d, q); }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)dim + eps);
    TO* yr = y + row * dim;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        if (i < nv || (i == nv && lane * 4 < tail)) {
            const int c = i * 256 + lane * 4;
            const f32x4 g = *(const f32x4*)(gamma + c), bt = *(const f32x4*)(beta + c);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + bt[e];
            store4<TO>(yr + c, o);
            if (stats_out) {
                // (sum, M2 about the block mean) of the STORED values per 64-column block (16 lanes x 4), the same quantities a
                // GEMM epilogue emits: the first block's fused LayerNorm merges them in place (no row-statistics pass)
                float r4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) r4[e] = (float)(TO)o[e];
                float s1 = (r4[0] + r4[1]) + (r4[2] + r4[3]);
#pragma unroll
                for (int of = 1; of < 16; of <<= 1) s1 += __shfl_xor(s1, of);
                const float mb = s1 * (1.0f / 64.0f);
                float m2 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float dl = r4[e] - mb; m2 = fmaf(dl, dl, m2); }
#pragma unroll
                for (int of = 1; of < 16; of <<= 1) m2 += __shfl_xor(m2, of);
                if ((lane & 15) == 0) {
                    f32x2 w;
                    w[0] = s1; w[1] = m2;
                    *(f32x2*)(stats_out + ((int64_t)(i * 4 + (lane >> 4)) * rows + row) * 2) = w;     // slot-major [dim/64][rows][2]
                }
            }
        }
    }
}

// The same pass for the hot configuration (round 4): 16-bit conv output and stream of one type T, dim a multiple of 256.  A HALF-wave per row: lane l
// of the half moves the 16-byte chunks l, l + 32, l + 64 .. of the row (the one-wave-per-row form above moves 8 bytes per lane and load: half the bytes per
// memory instruction; 65 us at B = 256 against the 31 us its 154 MB take at 5 TB/s), 8 rows per 256-thread workgroup.  A 64-column block is the 8 consecutive
// lanes of one load index, so the block partials are 8-lane butterflies; the row statistics 32-lane ones.
template <typename T, int NV>
__global__ __launch_bounds__(256) void embed_ln_pre16_kernel(const T* __restrict__ conv_out, const float* __restrict__ cls, const float* __restrict__ pos,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y,
                                                             float* __restrict__ stats_out, int64_t rows, int Tk, float eps) {
    typedef typename VecOf<T>::v8 v8;
    constexpr int dim = NV * 256;
    const int l = threadIdx.x & 31;
    const int64_t row = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (row >= rows) return;                     // (a half-wave leaves together; the butterflies below stay inside a half)
    const int64_t b = row / Tk;
    const int t = (int)(row - b * Tk);
    const T* xr = conv_out + (b * (Tk - 1) + (t > 0 ? t - 1 : 0)) * dim;
    const float* pr = pos + (int64_t)t * dim;
    float v[NV][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = j * 256 + l * 8;
        const f32x4 p0 = *(const f32x4*)(pr + c), p1 = *(const f32x4*)(pr + c + 4);
        if (t > 0) {
            const v8 a8 = *(const v8*)(xr + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[j][e] = (float)a8[e] + p0[e]; v[j][4 + e] = (float)a8[4 + e] + p1[e]; }
        } else {
            const f32x4 c0 = *(const f32x4*)(cls + c), c1 = *(const f32x4*)(cls + c + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[j][e] = c0[e] + p0[e]; v[j][4 + e] = c1[e] + p1[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; e += 2) s += v[j][e] + v[j][e + 1];
    }
#pragma unroll
    for (int of = 1; of < 32; of <<= 1) s += __shfl_xor(s, of);
    const float mean = s * (1.0f / (float)dim);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[j][e] - mean; q = fmaf(d, d, q); }
#pragma unroll
    for (int of = 1; of < 32; of <<= 1) q += __shfl_xor(q, of);
    const float rstd = rsqrtf(q * (1.0f / (float)dim) + eps);
    T* yr = y + row * dim;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = j * 256 + l * 8;
        const f32x4 g0 = *(const f32x4*)(gamma + c), g1 = *(const f32x4*)(gamma + c + 4);
        const f32x4 b0 = *(const f32x4*)(beta + c), b1 = *(const f32x4*)(beta + c + 4);
        v8 o8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o8[e] = (T)((v[j][e] - mean) * rstd * g0[e] + b0[e]);
            o8[4 + e] = (T)((v[j][4 + e] - mean) * rstd * g1[e] + b1[e]);
        }
        *(v8*)(yr + c) = o8;
        if (stats_out) {
            // (sum, M2 about the block mean) of the STORED values of the 64-column block j * 4 + (l >> 3): the quantities a GEMM epilogue emits
            float s1 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s1 += (float)o8[e];
#pragma unroll
            for (int of = 1; of < 8; of <<= 1) s1 += __shfl_xor(s1, of);
            const float mb = s1 * (1.0f / 64.0f);
            float m2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dl = (float)o8[e] - mb; m2 = fmaf(dl, dl, m2); }
#pragma unroll
            for (int of = 1; of < 8; of <<= 1) m2 += __shfl_xor(m2, of);
            if ((l & 7) == 0) {
                f32x2 w;
                w[0] = s1; w[1] = m2;
                *(f32x2*)(stats_out + ((int64_t)(j * 4 + (l >> 3)) * rows + row) * 2) = w;     // slot-major [dim/64][rows][2]
            }
        }
    }
}

template <typename TI>
int ln_dispatch_out(const void* x, const float* g, const float* b, void* y, int64_t rows, int dim, int64_t ldx,
                    int64_t ldy, float eps, int ydt, hipStream_t s) {
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (ydt == LECLIP_F32) hipLaunchKernelGGL((layernorm_kernel<TI, float>), grid, block, 0, s, (const TI*)x, g, b, (float*)y, rows, dim, ldx, ldy, eps);
    else if (ydt == LECLIP_F16) hipLaunchKernelGGL((layernorm_kernel<TI, f16_t>), grid, block, 0, s, (const TI*)x, g, b, (f16_t*)y, rows, dim, ldx, ldy, eps);
    else hipLaunchKernelGGL((layernorm_kernel<TI, bf16_t>), grid, block, 0, s, (const TI*)x, g, b, (bf16_t*)y, rows, dim, ldx, ldy, eps);
    return leclip_check_launch("layernorm_kernel");
}

template <typename TP>
__device__ __forceinline__ float proj_column(const TP* __restrict__ p, const float* __restrict__ xn, int dim, int E) {
    float acc = 0.f;
    int k = 0;
    for (; k + 8 <= dim; k += 8) {
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = (float)p[(int64_t)(k + u) * E];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fmaf(xn[k + u], w[u], acc);
    }
    for (; k < dim; ++k) acc = fmaf(xn[k], (float)p[(int64_t)k * E], acc);
    return acc;
}

// ------------------------------------------------------------------------------- row statistics for fused LayerNorm
// (mean, rstd) per row, two-pass fp32 like layernorm_kernel (one wave per row).
template <typename TI>
__global__ __launch_bounds__(256) void row_stats_kernel(const TI* __restrict__ x, float* __restrict__ stats, int64_t rows, int dim,
                                                        int64_t ldx, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const TI* xr = x + row * ldx;
    const int nv = dim >> 8, tail = dim & 255;
    f32x4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
        if (i < nv || (i == nv && lane * 4 < tail)) { v[i] = load4<TI>(xr + i * 256 + lane * 4); s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
    const float mean = wave_sum(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
        if (i < nv || (i == nv && lane * 4 < tail)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q = fmaf(d, d, q); }
        }
    q = wave_sum(q);
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rsqrtf(q / (float)dim + eps); }
}

// Merge the per-(row, 64-column block) partials (sum, M2 about the block mean) a GEMM epilogue wrote into (mean, rstd):
// mean = sum of sums / dim, M2 = sum_b [ M2_b + 64 (mean_b - mean)^2 ]  (Chan et al.'s parallel-variance update, fixed block
// order): every term is a sum of squares of deviations, so rows whose mean dwarfs their spread lose nothing to cancellation.
// One thread per row; all of a row's partials are requested up front as independent 16-byte loads, then merged in the fixed
// block order by ln_merge_partials (shared with the GEMM kernels that merge in place).
__global__ __launch_bounds__(256) void ln_stats_finalize_kernel(const float* __restrict__ partials, float* __restrict__ stats, int64_t rows,
                                                                int slots, int dim, float eps) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    f32x4 v[LN_MERGE_MAXV];
    const f32x2* p = (const f32x2*)partials + r;       // slot-major [slots][rows][2]: consecutive threads read consecutive pairs of a slot
#pragma unroll
    for (int i = 0; i < LN_MERGE_MAXV; ++i)
        if (2 * i < slots) {
            const f32x2 a = p[(int64_t)(2 * i) * rows], b = p[(int64_t)(2 * i + 1) * rows];
            v[i][0] = a[0]; v[i][1] = a[1]; v[i][2] = b[0]; v[i][3] = b[1];
        }
    *(f32x2*)(stats + 2 * r) = ln_merge_partials(v, slots, dim, eps);
}

// any slot count (odd, > 16): plain loop, same formula
__global__ void ln_stats_finalize_generic_kernel(const float* __restrict__ partials, float* __restrict__ stats, int64_t rows, int slots,
                                                 int dim, float eps) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const f32x2* p = (const f32x2*)partials + r;
    float s1 = 0.f;
    for (int i = 0; i < slots; ++i) s1 += p[(int64_t)i * rows][0];
    const float mean = s1 / (float)dim;
    const float bn = (float)(dim / slots);
    float m2 = 0.f;
    for (int i = 0; i < slots; ++i) { const f32x2 t = p[(int64_t)i * rows]; const float dlt = t[0] / bn - mean; m2 += fmaf(bn * dlt, dlt, t[1]); }
    stats[2 * r] = mean;
    stats[2 * r + 1] = rsqrtf(m2 / (float)dim + eps);
}

// ------------------------------------------------------------------------- gather + LayerNorm + projection
// One workgroup per output row: the gathered row is normalised into LDS (fp32), then thread e accumulates
// out[e] = sum_k xn[k] * proj[k][e] reading proj rows coalesced across threads.
__global__ __launch_bounds__(256) void gather_ln_proj_kernel(const void* __restrict__ x, const int64_t* __restrict__ rows,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const void* __restrict__ proj, float* __restrict__ out,
                                                             int dim, int E, int64_t ldx, float eps, int xdt, int pdt) {
    extern __shared__ float xn[];   // [dim] + 8 scratch
    float* red = xn + dim;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r = rows ? rows[blockIdx.x] : (int64_t)blockIdx.x;
    float s = 0.f;
    for (int k = tid; k < dim; k += 256) { const float v = load_elem(x, xdt, r * ldx + k); xn[k] = v; s += v; }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)dim;
    float q = 0.f;
    for (int k = tid; k < dim; k += 256) { const float d = xn[k] - mean; q = fmaf(d, d, q); }
    q = wave_sum(q);
    if (lane == 0) red[4 + wave] = q;
    __syncthreads();
    const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)dim + eps);
    for (int k = tid; k < dim; k += 256) xn[k] = (xn[k] - mean) * rstd * gamma[k] + beta[k];
    __syncthreads();
    // thread e walks proj[:, e] (coalesced across threads), 8 independent loads in flight per step
    for (int e = tid; e < E; e += 256) {
        float acc = 0.f;
        if (pdt == LECLIP_F32) acc = proj_column<float>((const float*)proj + e, xn, dim, E);
        else if (pdt == LECLIP_F16) acc = proj_column<f16_t>((const f16_t*)proj + e, xn, dim, E);
        else acc = proj_column<bf16_t>((const bf16_t*)proj + e, xn, dim, E);
        out[(int64_t)blockIdx.x * E + e] = acc;
    }
}

// --------------------------------------------------------------------- L2 normalise + scaled cosine logits
// One thread per (image, class): it streams both 2 KiB feature rows with 16-byte loads (all of it L1/L2 resident: the
// two matrices are 0.7 MB) and keeps three dot products, <i,t>, <i,i>, <t,t>; the result is
// (scale * <i,t>) / (||i|| ||t||).  21 MFLOP at B = 256: pure latency, so maximum thread-level parallelism.
__global__ __launch_bounds__(256) void l2norm_logits_kernel(const float* __restrict__ img, const float* __restrict__ txt,
                                                            float* __restrict__ logits, int64_t B, int C, int D, float scale) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * C) return;
    const int64_t b = idx / C;
    const int c = (int)(idx - b * C);
    const f32x4* ip = (const f32x4*)(img + b * D);
    const f32x4* tp = (const f32x4*)(txt + (int64_t)c * D);
    float it = 0.f, ii = 0.f, tt = 0.f;
    for (int k = 0; k < D / 4; ++k) {
        const f32x4 x = ip[k], y = tp[k];
#pragma unroll
        for (int e = 0; e < 4; ++e) { it = fmaf(x[e], y[e], it); ii = fmaf(x[e], x[e], ii); tt = fmaf(y[e], y[e], tt); }
    }
    logits[idx] = (scale * it) / (sqrtf(ii) * sqrtf(tt));
}

// ------------------------------------------------------------------- score post-processing
// Sliding-window aggregation (trainers/Caption_distill_double.py:654-660): per image and class, alpha = max over windows,
// beta = min over windows, s_ag = alpha if alpha > thr else beta, out = w_ag * s_ag + global.   One thread per (b, c).
__global__ void window_aggregate_kernel(const float* __restrict__ glob, const float* __restrict__ blocks, float* __restrict__ out,
                                        int64_t B, int W, int C, float thr, float w_ag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int64_t b = i / C;
    const int c = (int)(i - b * C);
    const float* p = blocks + b * W * C + c;
    float mx = p[0], mn = p[0];
    for (int w = 1; w < W; ++w) { const float v = p[(int64_t)w * C]; mx = fmaxf(mx, v); mn = fminf(mn, v); }
    out[i] = w_ag * (mx > thr ? mx : mn) + glob[i];
}

// Co-occurrence modulation (Caption_distill_double.py:614-618, 632-636): out = p + weight * (p @ Mn), Mn [C, C] row-normalised.
__global__ void cooccurrence_adjust_kernel(const float* __restrict__ p, const float* __restrict__ Mn, float* __restrict__ out,
                                           int64_t B, int C, float weight) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int64_t b = i / C;
    const int c = (int)(i - b * C);
    float acc = 0.f;
    for (int k = 0; k < C; ++k) acc = fmaf(p[b * C + k], Mn[(int64_t)k * C + c], acc);
    out[i] = p[i] + weight * acc;
}

// Caption-feature mixing of DenseCLIP's test branch (trainers/Caption_distill_double.py:444-448): per image, the k captions whose
// (normalised) text features are most similar to the normalised global image feature are averaged, and the global feature becomes the
// mean of itself and that average:  out[b] = (img[b] + mean_{j in topk(sim[b])} feats[j]) / 2.   sim [B][ld_sim] holds the similarities
// (exact-fp32 MFMA GEMM by the caller), feats [N][E].  One workgroup per image; the k largest similarities are taken one at a time -
// pass j finds the largest entry that comes after pass j-1's pick in (value descending, index ascending) order, i.e. torch.topk's set
// with ties resolved towards the lower index - each pass a strided scan + a block arg-max; k * N reads per image from L2.
__global__ __launch_bounds__(256) void topk_mix_kernel(const float* __restrict__ sim, const float* __restrict__ feats, const float* __restrict__ img,
                                                       float* __restrict__ out, int64_t N, int E, int k, int64_t ld_sim) {
    __shared__ float rv[4];
    __shared__ int64_t ri[4];
    __shared__ int64_t pick;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = sim + (int64_t)blockIdx.x * ld_sim;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};      // E <= 1024: feature columns tid, tid + 256, ...
    float pv = INFINITY;
    int64_t pi = -1;
    for (int j = 0; j < k; ++j) {
        float bv = -INFINITY;
        int64_t bi = N;
        for (int64_t i = tid; i < N; i += 256) {
            const float v = row[i];
            const bool after = v < pv || (v == pv && i > pi);           // not picked yet
            if (after && (v > bv || (v == bv && i < bi))) { bv = v; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int64_t oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { rv[wave] = bv; ri[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float v = rv[0];
            int64_t i = ri[0];
            for (int w = 1; w < 4; ++w) if (rv[w] > v || (rv[w] == v && ri[w] < i)) { v = rv[w]; i = ri[w]; }
            rv[0] = v;
            pick = i;
        }
        __syncthreads();
        pv = rv[0];
        pi = pick;
        __syncthreads();
        if (pi < N) {
            const float* f = feats + pi * E;
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int c = tid + 256 * u; if (c < E) acc[u] += f[c]; }
        }
    }
    const float inv = 1.0f / (float)k;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = tid + 256 * u;
        if (c < E) out[(int64_t)blockIdx.x * E + c] = 0.5f * (img[(int64_t)blockIdx.x * E + c] + acc[u] * inv);
    }
}

// ------------------------------------------------------------------------------ embeddings / prompt assembly
__global__ void embed_tokens_kernel(const int64_t* __restrict__ tokens, const float* __restrict__ table,
                                    const float* __restrict__ pos, void* __restrict__ x, int64_t n_rows, int T, int dim,
                                    int64_t vocab, int xdt) {
    const int64_t row = blockIdx.x;
    if (row >= n_rows) return;
    int64_t tok = tokens[row];
    tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
    const int t = (int)(row % T);
    for (int k = threadIdx.x; k < dim; k += blockDim.x)
        store_elem(x, xdt, row * dim + k, table[tok * dim + k] + pos[(int64_t)t * dim + k]);
}

__global__ void prompt_assemble_kernel(const float* __restrict__ prefix, const float* __restrict__ ctx,
                                       const float* __restrict__ suffix, const float* __restrict__ pos,
                                       void* __restrict__ x, int n_ctx, int T, int dim, int ctx_per_class, int xdt) {
    const int64_t row = blockIdx.x;        // c * T + t
    const int64_t c = row / T;
    const int t = (int)(row - c * T);
    const float* src;
    if (t == 0) src = prefix + c * dim;
    else if (t <= n_ctx) src = ctx + ((ctx_per_class ? c * n_ctx : 0) + (t - 1)) * (int64_t)dim;
    else src = suffix + (c * (T - 1 - n_ctx) + (t - 1 - n_ctx)) * (int64_t)dim;
    for (int k = threadIdx.x; k < dim; k += blockDim.x)
        store_elem(x, xdt, row * dim + k, src[k] + (pos ? pos[(int64_t)t * dim + k] : 0.f));
}

__global__ void add_pos_kernel(const float* __restrict__ in, const float* __restrict__ pos, void* __restrict__ x, int T, int dim,
                               int xdt) {
    const int64_t row = blockIdx.x;
    const int t = (int)(row % T);
    for (int k = threadIdx.x; k < dim; k += blockDim.x)
        store_elem(x, xdt, row * dim + k, in[row * dim + k] + pos[(int64_t)t * dim + k]);
}

__global__ void eot_index_kernel(const int64_t* __restrict__ tokens, int64_t* __restrict__ eot,
                                 int64_t* __restrict__ flat, int64_t n, int T) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int64_t best = tokens[r * T];
    int bi = 0;
    for (int t = 1; t < T; ++t) { const int64_t v = tokens[r * T + t]; if (v > best) { best = v; bi = t; } }
    if (eot) eot[r] = bi;
    if (flat) flat[r] = r * T + bi;
}

// ------------------------------------------------------------------------------------------ patch extraction
// patches[(b,gy,gx)][(c,py,px)] = image[b][c][gy*P+py][gx*P+px]; one thread per (row, c, py) run of P pixels.
__global__ void im2col_kernel(const void* __restrict__ image, void* __restrict__ patches, int64_t B, int R, int P, int Kp,
                              int idt, int odt) {
    const int G = R / P;
    const int64_t total = B * G * G * 3 * P;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int py = (int)(i % P);
    int64_t r = i / P;
    const int c = (int)(r % 3);
    r /= 3;                                  // patch row index (b, gy, gx)
    const int gx = (int)(r % G);
    const int gy = (int)((r / G) % G);
    const int64_t b = r / ((int64_t)G * G);
    const int64_t src = ((b * 3 + c) * R + (gy * P + py)) * (int64_t)R + gx * P;
    const int64_t dst = r * Kp + (c * P + py) * P;
    for (int px = 0; px < P; ++px) store_elem(patches, odt, dst + px, load_elem(image, idt, src + px));
    if (c == 2 && py == P - 1)
        for (int k = 3 * P * P; k < Kp; ++k) store_elem(patches, odt, r * Kp + k, 0.f);
}

// P % 8 == 0: one thread per (patch row, 8 consecutive k): 8 pixels of one image row (32 B fp32 / 16 B 16-bit load),
// one 16-byte store; consecutive threads write consecutive chunks of the patch matrix (fully coalesced stores).
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void im2col8_kernel(const TI* __restrict__ image, TO* __restrict__ patches, int64_t n_rows,
                                                      int R, int P, int Kp) {
    const int chunks = Kp >> 3;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows * chunks) return;
    const int64_t r = i / chunks;
    const int k0 = (int)(i - r * chunks) * 8;
    const int G = R / P;
    float v[8];
    if (k0 < 3 * P * P) {
        const int c = k0 / (P * P), rem = k0 - c * P * P, py = rem / P, px0 = rem - py * P;
        const int gx = (int)(r % G), gy = (int)((r / G) % G);
        const int64_t b = r / ((int64_t)G * G);
        const TI* src = image + ((b * 3 + c) * R + (gy * P + py)) * (int64_t)R + gx * P + px0;
#pragma unroll
        for (int e = 0; e < 8; e += 4) {
            const f32x4 t = load4<TI>(src + e);
#pragma unroll
            for (int u = 0; u < 4; ++u) v[e + u] = t[u];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    TO* dst = patches + r * Kp + k0;
#pragma unroll
    for (int e = 0; e < 8; e += 4) {
        f32x4 t;
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = v[e + u];
        store4<TO>(dst + e, t);
    }
}

template <typename TI>
void im2col8_out(const void* image, void* patches, int64_t n_rows, int R, int P, int Kp, int odt, hipStream_t s) {
    const dim3 grid((unsigned)((n_rows * (Kp / 8) + 255) / 256)), block(256);
    if (odt == LECLIP_F32) hipLaunchKernelGGL((im2col8_kernel<TI, float>), grid, block, 0, s, (const TI*)image, (float*)patches, n_rows, R, P, Kp);
    else if (odt == LECLIP_F16) hipLaunchKernelGGL((im2col8_kernel<TI, f16_t>), grid, block, 0, s, (const TI*)image, (f16_t*)patches, n_rows, R, P, Kp);
    else hipLaunchKernelGGL((im2col8_kernel<TI, bf16_t>), grid, block, 0, s, (const TI*)image, (bf16_t*)patches, n_rows, R, P, Kp);
}

// P even but not a multiple of 8 (ViT-L/14: 14-pixel patch rows, 28 bytes in 16 bits - round 5): one thread per 8 consecutive k of a patch row, its
// four pixel PAIRS decoded one by one (a pair never straddles a patch row: P and k are even), one 16-byte store; consecutive threads write consecutive
// chunks.  The one-thread-per-pixel-run kernel above moved ViT-L/14@336's 181 MB with 2-byte accesses in 236 us.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void im2col2_kernel(const TI* __restrict__ image, TO* __restrict__ patches, int64_t n_rows, int R, int P, int Kp) {
    const int chunks = Kp >> 3;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows * chunks) return;
    const int64_t r = i / chunks;
    const int k0 = (int)(i - r * chunks) * 8;
    const int G = R / P, PP = P * P;
    const int gx = (int)(r % G), gy = (int)((r / G) % G);
    const int64_t b = r / ((int64_t)G * G);
    const TI* img = image + ((b * 3) * R + gy * P) * (int64_t)R + gx * P;
    float v[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = k0 + 2 * u;
        v[2 * u] = 0.f; v[2 * u + 1] = 0.f;
        if (k < 3 * PP) {
            const int c = k / PP, rem = k - c * PP, py = rem / P, px = rem - py * P;
            const TI* src = img + ((int64_t)c * R + py) * R + px;
            if constexpr (sizeof(TI) == 4) { const f32x2 t = *(const f32x2*)src; v[2 * u] = t[0]; v[2 * u + 1] = t[1]; }
            else {
                typedef TI t2 __attribute__((ext_vector_type(2)));
                const t2 t = *(const t2*)src;
                v[2 * u] = (float)t[0]; v[2 * u + 1] = (float)t[1];
            }
        }
    }
    TO* dst = patches + r * Kp + k0;
#pragma unroll
    for (int e = 0; e < 8; e += 4) {
        f32x4 t;
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = v[e + u];
        store4<TO>(dst + e, t);
    }
}

template <typename TI>
void im2col2_out(const void* image, void* patches, int64_t n_rows, int R, int P, int Kp, int odt, hipStream_t s) {
    const dim3 grid((unsigned)((n_rows * (Kp / 8) + 255) / 256)), block(256);
    if (odt == LECLIP_F32) hipLaunchKernelGGL((im2col2_kernel<TI, float>), grid, block, 0, s, (const TI*)image, (float*)patches, n_rows, R, P, Kp);
    else if (odt == LECLIP_F16) hipLaunchKernelGGL((im2col2_kernel<TI, f16_t>), grid, block, 0, s, (const TI*)image, (f16_t*)patches, n_rows, R, P, Kp);
    else hipLaunchKernelGGL((im2col2_kernel<TI, bf16_t>), grid, block, 0, s, (const TI*)image, (bf16_t*)patches, n_rows, R, P, Kp);
}

__global__ void class_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos, void* __restrict__ X,
                                  int T, int width, int xdt) {
    const int64_t b = blockIdx.x;
    for (int k = threadIdx.x; k < width; k += blockDim.x)
        store_elem(X, xdt, b * T * (int64_t)width + k, cls[k] + pos[k]);
}

}  // namespace

int leclip_gemm_dispatch(const void* A, const void* W, int64_t M, int N, int K, int64_t lda, int64_t ldw,
                         const EpiParams& epi, int ab_dtype, hipStream_t s);

extern "C" int leclip_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, int64_t rows, int dim,
                                    int64_t ldx, int64_t ldy, float eps, leclip_dtype x_dtype, leclip_dtype y_dtype,
                                    void* stream) {
    if (!x || !gamma || !beta || !y || rows <= 0 || dim <= 0 || ldx < dim || ldy < dim) {
        leclip_set_error("layernorm: null pointer or inconsistent sizes"); return LECLIP_E_INVALID;
    }
    if (!dtype_ok(x_dtype) || !dtype_ok(y_dtype)) { leclip_set_error("layernorm: bad dtype"); return LECLIP_E_INVALID; }
    if (dim % 64 != 0 || dim > 256 * LN_MAXV) {
        leclip_set_error("layernorm: dim=%d must be a multiple of 64 and <= %d", dim, 256 * LN_MAXV); return LECLIP_E_UNSUPPORTED;
    }
    if ((ldx % 4) || (ldy % 4) || ((uintptr_t)x & 7) || ((uintptr_t)y & 7) || ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15)) {
        leclip_set_error("layernorm: rows must be 8-byte aligned (ld %% 4 == 0), gamma/beta 16-byte aligned"); return LECLIP_E_INVALID;
    }
    if (x_dtype == LECLIP_F32 && (((uintptr_t)x & 15))) { leclip_set_error("layernorm: fp32 input must be 16-byte aligned"); return LECLIP_E_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == LECLIP_F32) return ln_dispatch_out<float>(x, gamma, beta, y, rows, dim, ldx, ldy, eps, y_dtype, s);
    if (x_dtype == LECLIP_F16) return ln_dispatch_out<f16_t>(x, gamma, beta, y, rows, dim, ldx, ldy, eps, y_dtype, s);
    return ln_dispatch_out<bf16_t>(x, gamma, beta, y, rows, dim, ldx, ldy, eps, y_dtype, s);
}

extern "C" int leclip_gather_ln_proj_fwd(const void* x, const int64_t* row_index, const float* gamma, const float* beta,
                                         const void* proj, float* out, int64_t n, int dim, int E, int64_t ldx, float eps,
                                         leclip_dtype x_dtype, leclip_dtype proj_dtype, void* stream) {
    if (!x || !gamma || !beta || !proj || !out || n <= 0 || dim <= 0 || E <= 0 || ldx < dim) {
        leclip_set_error("gather_ln_proj: null pointer or inconsistent sizes"); return LECLIP_E_INVALID;
    }
    if (!dtype_ok(x_dtype) || !dtype_ok(proj_dtype)) { leclip_set_error("gather_ln_proj: bad dtype"); return LECLIP_E_INVALID; }
    if (dim > 8192) { leclip_set_error("gather_ln_proj: dim=%d > 8192", dim); return LECLIP_E_UNSUPPORTED; }
    hipLaunchKernelGGL(gather_ln_proj_kernel, dim3((unsigned)n), dim3(256), (dim + 8) * sizeof(float), (hipStream_t)stream,
                       x, row_index, gamma, beta, proj, out, dim, E, ldx, eps, (int)x_dtype, (int)proj_dtype);
    return leclip_check_launch("gather_ln_proj_kernel");
}

extern "C" int leclip_l2norm_logits_fwd(const float* img, const float* txt, float* logits, int64_t B, int C, int D,
                                        float scale, void* stream) {
    if (!img || !txt || !logits || B <= 0 || C <= 0 || D <= 0) { leclip_set_error("logits: null pointer or bad size"); return LECLIP_E_INVALID; }
    if (D % 4 != 0 || ((uintptr_t)img & 15) || ((uintptr_t)txt & 15)) {
        leclip_set_error("logits: D=%d must be a multiple of 4 and the feature matrices 16-byte aligned", D); return LECLIP_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL(l2norm_logits_kernel, dim3((unsigned)((B * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       img, txt, logits, B, C, D, scale);
    return leclip_check_launch("l2norm_logits_kernel");
}

extern "C" int leclip_embed_tokens_fwd(const int64_t* tokens, const float* table, const float* pos, void* x, int64_t n, int T,
                                       int dim, int64_t vocab, leclip_dtype x_dtype, void* stream) {
    if (!tokens || !table || !pos || !x || n <= 0 || T <= 0 || dim <= 0 || vocab <= 0 || !dtype_ok(x_dtype)) {
        leclip_set_error("embed_tokens: null pointer or bad size"); return LECLIP_E_INVALID;
    }
    hipLaunchKernelGGL(embed_tokens_kernel, dim3((unsigned)(n * T)), dim3(128), 0, (hipStream_t)stream, tokens, table, pos, x,
                       n * T, T, dim, vocab, (int)x_dtype);
    return leclip_check_launch("embed_tokens_kernel");
}

extern "C" int leclip_prompt_assemble_fwd(const float* prefix, const float* ctx, const float* suffix, const float* pos, void* x,
                                          int64_t n_cls, int n_ctx, int T, int dim, int ctx_per_class, leclip_dtype x_dtype,
                                          void* stream) {
    if (!prefix || !ctx || !suffix || !x || n_cls <= 0 || n_ctx < 0 || T <= 1 + n_ctx || dim <= 0 || !dtype_ok(x_dtype)) {
        leclip_set_error("prompt_assemble: null pointer or bad size"); return LECLIP_E_INVALID;
    }
    hipLaunchKernelGGL(prompt_assemble_kernel, dim3((unsigned)(n_cls * T)), dim3(128), 0, (hipStream_t)stream, prefix, ctx, suffix,
                       pos, x, n_ctx, T, dim, ctx_per_class, (int)x_dtype);
    return leclip_check_launch("prompt_assemble_kernel");
}

extern "C" int leclip_add_pos_fwd(const float* in, const float* pos, void* x, int64_t n, int T, int dim, leclip_dtype x_dtype,
                                  void* stream) {
    if (!in || !pos || !x || n <= 0 || T <= 0 || dim <= 0 || !dtype_ok(x_dtype)) {
        leclip_set_error("add_pos: null pointer or bad size"); return LECLIP_E_INVALID;
    }
    hipLaunchKernelGGL(add_pos_kernel, dim3((unsigned)(n * T)), dim3(128), 0, (hipStream_t)stream, in, pos, x, T, dim, (int)x_dtype);
    return leclip_check_launch("add_pos_kernel");
}

extern "C" int leclip_eot_index_fwd(const int64_t* tokens, int64_t* eot, int64_t* flat_row, int64_t n, int T, void* stream) {
    if (!tokens || (!eot && !flat_row) || n <= 0 || T <= 0) { leclip_set_error("eot_index: null pointer or bad size"); return LECLIP_E_INVALID; }
    hipLaunchKernelGGL(eot_index_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, tokens, eot, flat_row, n, T);
    return leclip_check_launch("eot_index_kernel");
}

bool leclip_gemm256_im2col_eligible(int64_t B, int R, int P, int N, int img_dtype, int w_dtype, const void* image);
int leclip_gemm256_launch_im2col(const void* image, const void* W, int64_t B, int R, int N, int64_t ldw, const EpiParams& epi, int ab_dtype,
                                 hipStream_t s);

static inline int patch_kp(int P, int w_dtype) { const int k = 3 * P * P, a = w_dtype == LECLIP_F32 ? 32 : 64; return (k + a - 1) / a * a; }

extern "C" int64_t leclip_patch_embed_workspace_bytes(int64_t B, int R, int P, leclip_dtype w_dtype) {
    if (B <= 0 || R <= 0 || P <= 0 || R % P) return LECLIP_E_INVALID;
    const int64_t G = R / P;
    return B * G * G * patch_kp(P, w_dtype) * dtype_size(w_dtype);
}

extern "C" int leclip_patch_embed_fwd(const void* image, const void* Wp, const float* class_emb, const float* pos, void* X,
                                      int64_t B, int R, int P, int width, leclip_dtype img_dtype, leclip_dtype w_dtype,
                                      leclip_dtype x_dtype, void* workspace, void* stream) {
    if (!image || !Wp || !class_emb || !pos || !X || !workspace || B <= 0 || R <= 0 || P <= 0 || R % P || width <= 0) {
        leclip_set_error("patch_embed: null pointer or inconsistent sizes"); return LECLIP_E_INVALID;
    }
    if (!dtype_ok(img_dtype) || !dtype_ok(w_dtype) || !dtype_ok(x_dtype)) { leclip_set_error("patch_embed: bad dtype"); return LECLIP_E_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    const int G = R / P, T = G * G + 1, Kp = patch_kp(P, w_dtype);
    const int64_t total = B * G * G * 3 * P;
    if (P % 8 == 0 && ((uintptr_t)image & 15) == 0) {
        const int64_t n_rows = B * G * G;
        if (img_dtype == LECLIP_F32) im2col8_out<float>(image, workspace, n_rows, R, P, Kp, (int)w_dtype, s);
        else if (img_dtype == LECLIP_F16) im2col8_out<f16_t>(image, workspace, n_rows, R, P, Kp, (int)w_dtype, s);
        else im2col8_out<bf16_t>(image, workspace, n_rows, R, P, Kp, (int)w_dtype, s);
    } else if (P % 2 == 0 && R % 2 == 0 && ((uintptr_t)image & 7) == 0) {
        const int64_t n_rows = B * G * G;
        if (img_dtype == LECLIP_F32) im2col2_out<float>(image, workspace, n_rows, R, P, Kp, (int)w_dtype, s);
        else if (img_dtype == LECLIP_F16) im2col2_out<f16_t>(image, workspace, n_rows, R, P, Kp, (int)w_dtype, s);
        else im2col2_out<bf16_t>(image, workspace, n_rows, R, P, Kp, (int)w_dtype, s);
    } else {
        hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, image, workspace, B, R, P, Kp,
                           (int)img_dtype, (int)w_dtype);
    }
    int rc = leclip_check_launch("im2col_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(class_rows_kernel, dim3((unsigned)B), dim3(256), 0, s, class_emb, pos, X, T, width, (int)x_dtype);
    rc = leclip_check_launch("class_rows_kernel");
    if (rc) return rc;
    EpiParams e;
    e.bias = nullptr; e.res = pos; e.out = X; e.ldr = width; e.ldy = width;
    e.res_dt = LECLIP_F32; e.out_dt = x_dtype; e.act = LECLIP_ACT_NONE; e.rowmap_P = G * G;
    e.ln_stats = nullptr; e.ln_colsum = nullptr; e.ln_partials = nullptr; e.ln_slots = 0; e.ln_eps = 0.f; e.stats_out = nullptr; e.stats_slots = 0; e.stats_rows = 0;
    return leclip_gemm_dispatch(workspace, Wp, B * G * G, width, Kp, Kp, Kp, e, w_dtype, s);
}

extern "C" int64_t leclip_patch_embed_ln_workspace_bytes(int64_t B, int R, int P, int width, leclip_dtype w_dtype) {
    const int64_t a = leclip_patch_embed_workspace_bytes(B, R, P, w_dtype);
    if (a < 0 || width <= 0) return LECLIP_E_INVALID;
    const int64_t G = R / P;
    return ((a + 255) & ~(int64_t)255) + B * G * G * width * dtype_size(w_dtype);
}

extern "C" int leclip_patch_embed_ln_fwd(const void* image, const void* Wp, const float* class_emb, const float* pos, const float* gamma,
                                         const float* beta, void* X, float* stats_out, int64_t B, int R, int P, int width,
                                         leclip_dtype img_dtype, leclip_dtype w_dtype, leclip_dtype x_dtype, float eps, void* workspace,
                                         void* stream) {
    if (!image || !Wp || !class_emb || !pos || !gamma || !beta || !X || !workspace || B <= 0 || R <= 0 || P <= 0 || R % P || width <= 0) {
        leclip_set_error("patch_embed_ln: null pointer or inconsistent sizes"); return LECLIP_E_INVALID;
    }
    if (!dtype_ok(img_dtype) || !dtype_ok(w_dtype) || !dtype_ok(x_dtype)) { leclip_set_error("patch_embed_ln: bad dtype"); return LECLIP_E_INVALID; }
    if (width % 64 != 0 || width > 256 * LN_MAXV) { leclip_set_error("patch_embed_ln: width=%d must be a multiple of 64 and <= %d", width, 256 * LN_MAXV); return LECLIP_E_UNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    const int G = R / P, T = G * G + 1, Kp = patch_kp(P, w_dtype);
    const int64_t n_rows = B * G * G;
    void* conv = (char*)workspace + ((leclip_patch_embed_workspace_bytes(B, R, P, w_dtype) + 255) & ~(int64_t)255);
    const bool direct = leclip_gemm256_im2col_eligible(B, R, P, width, (int)img_dtype, (int)w_dtype, image);
    if (direct) {
        // im2col-free (round 4): images in the compute dtype, 16 x 16 patches, a batch that fills the 256 x 256 kernel - the GEMM's LDS-DMA
        // gathers its A tiles from the NCHW image, no patch matrix is written or read (42 us and 2 x 77 MB per B=256 forward)
    } else if (P % 8 == 0 && ((uintptr_t)image & 15) == 0) {
        if (img_dtype == LECLIP_F32) im2col8_out<float>(image, workspace, n_rows, R, P, Kp, (int)w_dtype, s);
        else if (img_dtype == LECLIP_F16) im2col8_out<f16_t>(image, workspace, n_rows, R, P, Kp, (int)w_dtype, s);
        else im2col8_out<bf16_t>(image, workspace, n_rows, R, P, Kp, (int)w_dtype, s);
    } else if (P % 2 == 0 && R % 2 == 0 && ((uintptr_t)image & 7) == 0) {     // ViT-L/14: pixel pairs, 16-byte stores (round 5)
        if (img_dtype == LECLIP_F32) im2col2_out<float>(image, workspace, n_rows, R, P, Kp, (int)w_dtype, s);
        else if (img_dtype == LECLIP_F16) im2col2_out<f16_t>(image, workspace, n_rows, R, P, Kp, (int)w_dtype, s);
        else im2col2_out<bf16_t>(image, workspace, n_rows, R, P, Kp, (int)w_dtype, s);
    } else {
        const int64_t total = n_rows * 3 * P;
        hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, image, workspace, B, R, P, Kp, (int)img_dtype, (int)w_dtype);
    }
    int rc = leclip_check_launch("im2col_kernel");
    if (rc) return rc;
    EpiParams e;
    e.bias = nullptr; e.res = nullptr; e.out = conv; e.ldr = 0; e.ldy = width;
    e.res_dt = LECLIP_F32; e.out_dt = w_dtype; e.act = LECLIP_ACT_NONE; e.rowmap_P = 0;
    e.ln_stats = nullptr; e.ln_colsum = nullptr; e.ln_partials = nullptr; e.ln_slots = 0; e.ln_eps = 0.f; e.stats_out = nullptr; e.stats_slots = 0; e.stats_rows = 0;
    rc = direct ? leclip_gemm256_launch_im2col(image, Wp, B, R, width, Kp, e, (int)w_dtype, s)
                : leclip_gemm_dispatch(workspace, Wp, n_rows, width, Kp, Kp, Kp, e, w_dtype, s);
    if (rc) return rc;
    if (w_dtype != LECLIP_F32 && x_dtype == w_dtype && width % 256 == 0 && width <= 1024) {      // the hot configuration: half a wave per row, 16-byte accesses
        const dim3 grid8((unsigned)((B * T + 7) / 8)), block8(256);
#define LAUNCH_EMB16(TT, NV) hipLaunchKernelGGL((embed_ln_pre16_kernel<TT, NV>), grid8, block8, 0, s, (const TT*)conv, class_emb, pos, gamma, beta, (TT*)X, stats_out, B * T, T, eps)
#define LAUNCH_EMB16_T(TT) do { if (width == 256) LAUNCH_EMB16(TT, 1); else if (width == 512) LAUNCH_EMB16(TT, 2); else if (width == 768) LAUNCH_EMB16(TT, 3); else LAUNCH_EMB16(TT, 4); } while (0)
        if (w_dtype == LECLIP_F16) LAUNCH_EMB16_T(f16_t); else LAUNCH_EMB16_T(bf16_t);
#undef LAUNCH_EMB16_T
#undef LAUNCH_EMB16
        return leclip_check_launch("embed_ln_pre16_kernel");
    }
    const dim3 grid((unsigned)((B * T + 3) / 4)), block(256);
#define LAUNCH_EMB(TI, TO) hipLaunchKernelGGL((embed_ln_pre_kernel<TI, TO>), grid, block, 0, s, (const TI*)conv, class_emb, pos, gamma, beta, (TO*)X, stats_out, B * T, T, width, eps)
    if (w_dtype == LECLIP_F32) { if (x_dtype != LECLIP_F32) { leclip_set_error("patch_embed_ln: fp32 weights need an fp32 stream"); return LECLIP_E_UNSUPPORTED; } LAUNCH_EMB(float, float); }
    else if (w_dtype == LECLIP_F16) { if (x_dtype == LECLIP_F16) LAUNCH_EMB(f16_t, f16_t); else if (x_dtype == LECLIP_F32) LAUNCH_EMB(f16_t, float); else { leclip_set_error("patch_embed_ln: dtype mix"); return LECLIP_E_UNSUPPORTED; } }
    else { if (x_dtype == LECLIP_BF16) LAUNCH_EMB(bf16_t, bf16_t); else if (x_dtype == LECLIP_F32) LAUNCH_EMB(bf16_t, float); else { leclip_set_error("patch_embed_ln: dtype mix"); return LECLIP_E_UNSUPPORTED; } }
#undef LAUNCH_EMB
    return leclip_check_launch("embed_ln_pre_kernel");
}

extern "C" int leclip_row_stats_fwd(const void* x, float* stats, int64_t rows, int dim, int64_t ldx, float eps, leclip_dtype x_dtype,
                                    void* stream) {
    if (!x || !stats || rows <= 0 || dim <= 0 || ldx < dim || !dtype_ok(x_dtype)) { leclip_set_error("row_stats: bad argument"); return LECLIP_E_INVALID; }
    if (dim % 64 != 0 || dim > 256 * LN_MAXV || (ldx % 4) || ((uintptr_t)x & 7)) {
        leclip_set_error("row_stats: dim=%d must be a multiple of 64 and <= %d, rows 8-byte aligned", dim, 256 * LN_MAXV); return LECLIP_E_UNSUPPORTED;
    }
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == LECLIP_F32) hipLaunchKernelGGL((row_stats_kernel<float>), grid, block, 0, s, (const float*)x, stats, rows, dim, ldx, eps);
    else if (x_dtype == LECLIP_F16) hipLaunchKernelGGL((row_stats_kernel<f16_t>), grid, block, 0, s, (const f16_t*)x, stats, rows, dim, ldx, eps);
    else hipLaunchKernelGGL((row_stats_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)x, stats, rows, dim, ldx, eps);
    return leclip_check_launch("row_stats_kernel");
}

int leclip_ln_stats_finalize_launch(const float* partials, float* stats, int64_t rows, int slots, int dim, float eps, hipStream_t s) {
    const dim3 grid((unsigned)((rows + 255) / 256)), block(256);
    if (((uintptr_t)partials & 15) == 0 && slots % 2 == 0 && slots <= 2 * LN_MERGE_MAXV)
        hipLaunchKernelGGL(ln_stats_finalize_kernel, grid, block, 0, s, partials, stats, rows, slots, dim, eps);
    else
        hipLaunchKernelGGL(ln_stats_finalize_generic_kernel, grid, block, 0, s, partials, stats, rows, slots, dim, eps);
    return leclip_check_launch("ln_stats_finalize_kernel");
}

extern "C" int leclip_ln_stats_finalize_fwd(const float* partials, float* stats, int64_t rows, int slots, int dim, float eps,
                                            void* stream) {
    if (!partials || !stats || rows <= 0 || slots <= 0 || dim <= 0) { leclip_set_error("ln_stats_finalize: bad argument"); return LECLIP_E_INVALID; }
    return leclip_ln_stats_finalize_launch(partials, stats, rows, slots, dim, eps, (hipStream_t)stream);
}

extern "C" int leclip_window_aggregate_fwd(const float* global_logits, const float* window_logits, float* out, int64_t B, int W, int C,
                                           float threshold, float weight, void* stream) {
    if (!global_logits || !window_logits || !out || B <= 0 || W <= 0 || C <= 0) { leclip_set_error("window_aggregate: bad argument"); return LECLIP_E_INVALID; }
    hipLaunchKernelGGL(window_aggregate_kernel, dim3((unsigned)((B * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, global_logits,
                       window_logits, out, B, W, C, threshold, weight);
    return leclip_check_launch("window_aggregate_kernel");
}

extern "C" int leclip_topk_mix_fwd(const float* sim, const float* feats, const float* img, float* out, int64_t B, int64_t N, int E, int k,
                                   int64_t ld_sim, void* stream) {
    if (!sim || !feats || !img || !out || B <= 0 || N <= 0 || E <= 0 || k <= 0 || k > N || ld_sim < N) { leclip_set_error("topk_mix: bad argument"); return LECLIP_E_INVALID; }
    if (E > 1024) { leclip_set_error("topk_mix: E=%d > 1024", E); return LECLIP_E_UNSUPPORTED; }
    hipLaunchKernelGGL(topk_mix_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, sim, feats, img, out, N, E, k, ld_sim);
    return leclip_check_launch("topk_mix_kernel");
}

extern "C" int leclip_cooccurrence_adjust_fwd(const float* p, const float* Mn, float* out, int64_t B, int C, float weight, void* stream) {
    if (!p || !Mn || !out || B <= 0 || C <= 0 || p == out) { leclip_set_error("cooccurrence_adjust: bad argument (out must not alias p)"); return LECLIP_E_INVALID; }
    hipLaunchKernelGGL(cooccurrence_adjust_kernel, dim3((unsigned)((B * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, Mn, out, B, C,
                       weight);
    return leclip_check_launch("cooccurrence_adjust_kernel");
}
