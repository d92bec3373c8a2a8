// Backward kernels for the prompt-tuning step: gradients flow only to the learnable context
// vectors, i.e. through the ACTIVATIONS of the frozen text tower (reference trainers/Caption_distill_double.py:762-765
// freezes everything but prompt_learner; forward_backward at :789-897).  No weight gradients are formed.
//   dX of a linear layer     -> the forward TN GEMM kernels on a transposed weight copy (host side, hip/autograd.py)
//   LayerNorm backward (dx)  -> layernorm_bwd_kernel   (one wave per row, statistics recomputed, optional "+ upstream")
//   QuickGELU fwd / bwd      -> quickgelu_fwd_kernel / quickgelu_bwd_kernel (elementwise, 8 elements per thread)
//   attention backward       -> attn_bwd_kernel (probabilities recomputed from q, k; short sequences T <= 104)
#include "leclip_common.h"

namespace {

constexpr int LNB_MAXV = 16;

template <typename TI>
__device__ __forceinline__ f32x4 ld4(const TI* p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 ld4<bf16_t>(const bf16_t* p) {
    const bf16x4 v = *(const bf16x4*)p; f32x4 r; for (int i = 0; i < 4; ++i) r[i] = (float)v[i]; return r;
}
template <> __device__ __forceinline__ f32x4 ld4<f16_t>(const f16_t* p) {
    const f16x4 v = *(const f16x4*)p; f32x4 r; for (int i = 0; i < 4; ++i) r[i] = (float)v[i]; return r;
}
template <typename TO>
__device__ __forceinline__ void st4(TO* p, f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, f32x4 v) { bf16x4 r; for (int i = 0; i < 4; ++i) r[i] = (bf16_t)v[i]; *(bf16x4*)p = r; }
template <> __device__ __forceinline__ void st4<f16_t>(f16_t* p, f32x4 v) { f16x4 r; for (int i = 0; i < 4; ++i) r[i] = (f16_t)v[i]; *(f16x4*)p = r; }

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma,  xhat = (x - mean) * rstd;  out = dx (+ add)
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ gamma, const T* __restrict__ add,
                                                            T* __restrict__ dx, int64_t rows, int dim, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = dim >> 8, tail = dim & 255;
    const T* xr = x + row * dim;
    const T* gr = dy + row * dim;
    f32x4 xv[LNB_MAXV], gv[LNB_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i)
        if (i < nv || (i == nv && lane * 4 < tail)) {
            const int c = i * 256 + lane * 4;
            xv[i] = ld4<T>(xr + c);
            const f32x4 d = ld4<T>(gr + c), gm = *(const f32x4*)(gamma + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) gv[i][e] = d[e] * gm[e];
            s += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
        }
    const float mean = wave_sum(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i)
        if (i < nv || (i == nv && lane * 4 < tail)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = xv[i][e] - mean; q = fmaf(d, d, q); }
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)dim + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i)
        if (i < nv || (i == nv && lane * 4 < tail)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (xv[i][e] - mean) * rstd;
                xv[i][e] = xh;
                sg += gv[i][e];
                sgx = fmaf(gv[i][e], xh, sgx);
            }
        }
    const float mg = wave_sum(sg) / (float)dim, mgx = wave_sum(sgx) / (float)dim;
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i)
        if (i < nv || (i == nv && lane * 4 < tail)) {
            const int c = i * 256 + lane * 4;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rstd * (gv[i][e] - mg - xv[i][e] * mgx);
            if (add) {
                const f32x4 a = ld4<T>(add + row * dim + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += a[e];
            }
            st4<T>(dx + row * dim + c, o);
        }
}

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void quickgelu_kernel(const T* __restrict__ pre, const T* __restrict__ du, T* __restrict__ out, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 p = ld4<T>(pre + i * 4);
    f32x4 o;
    if (BWD) {
        const f32x4 g = ld4<T>(du + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sg = 1.0f / (1.0f + expf(-1.702f * p[e]));
            o[e] = g[e] * sg * (1.0f + 1.702f * p[e] * (1.0f - sg));   // d/dx [x * sigmoid(1.702 x)]
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = p[e] / (1.0f + expf(-1.702f * p[e]));
    }
    st4<T>(out + i * 4, o);
}

// ---------------------------------------------------------------------------------------------- attention backward
// One 1024-thread workgroup per (batch, head), T <= 104, head_dim 64.  Q, K, V, dO of the head in LDS as fp32 (rows
// of 68 floats: 16-byte aligned, so every contraction over d reads float4), probabilities recomputed:
//   P = softmax(scale QK^T + mask);  dV = P^T dO;  D = rowsum(dO o O) with O = P V recomputed (so dP is never stored);
//   dS = scale P o (dO V^T - D), in place over P;  dQ = dS K;  dK = dS^T Q.       fp32 arithmetic for every I/O dtype.
// Every phase is spread over all 1024 threads: (q,k) pairs for the two T x T phases, (row, 4 head dims) for the four
// T x 64 phases (a lane keeps 4 accumulators, the T x T operand is an LDS broadcast), one wave per row for softmax.
constexpr int AB_TMAX = 104;   // 4 x T x 68 + T x (T+1) + T floats of LDS <= 160 KiB
constexpr int AB_LD = 68;
constexpr int AB_THREADS = 1024;   // 4 waves per SIMD: the phases are fp32 VALU work, one wave per SIMD issues only every 4th cycle

__device__ __forceinline__ float dot64(const float* a, const float* b) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const f32x4 x = *(const f32x4*)(a + 4 * c), y = *(const f32x4*)(b + 4 * c);
        acc = fmaf(x[0], y[0], acc); acc = fmaf(x[1], y[1], acc); acc = fmaf(x[2], y[2], acc); acc = fmaf(x[3], y[3], acc);
    }
    return acc;
}

template <typename T>
__global__ __launch_bounds__(AB_THREADS) void attn_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ dout, T* __restrict__ dqkv,
                                                       int Tn, int heads, int64_t ld_qkv, int64_t ld_out, float scale, int causal) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sQ = sm;
    float* sK = sQ + Tn * AB_LD;
    float* sV = sK + Tn * AB_LD;
    float* sO = sV + Tn * AB_LD;          // dO
    float* sP = sO + Tn * AB_LD;          // [T][T+1]  P, then dS
    const int LP = Tn + 1;
    float* sD = sP + Tn * LP;             // [T]  D[q] = dO[q] . O[q]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
    const int d_model = heads * 64;
    const T* base = qkv + (int64_t)b * Tn * ld_qkv + h * 64;
    const T* dob = dout + (int64_t)b * Tn * ld_out + h * 64;
    T* dqb = dqkv + (int64_t)b * Tn * ld_qkv + h * 64;
    const int items = Tn * 16;            // (row, group of 4 head dims)
    for (int i = tid; i < items; i += AB_THREADS) {
        const int r = i >> 4, c = (i & 15) * 4;
        *(f32x4*)(sQ + r * AB_LD + c) = ld4<T>(base + (int64_t)r * ld_qkv + c);
        *(f32x4*)(sK + r * AB_LD + c) = ld4<T>(base + (int64_t)r * ld_qkv + d_model + c);
        *(f32x4*)(sV + r * AB_LD + c) = ld4<T>(base + (int64_t)r * ld_qkv + 2 * d_model + c);
        *(f32x4*)(sO + r * AB_LD + c) = ld4<T>(dob + (int64_t)r * ld_out + c);
    }
    __syncthreads();
    // scores
    for (int i = tid; i < Tn * Tn; i += AB_THREADS) {
        const int q = i / Tn, k = i - q * Tn;
        sP[q * LP + k] = (!causal || k <= q) ? dot64(sQ + q * AB_LD, sK + k * AB_LD) * scale : -3.0e38f;
    }
    __syncthreads();
    // row softmax: one wave per query row, two keys per lane (T <= 128)
    for (int q = wave; q < Tn; q += AB_THREADS / 64) {
        float* row = sP + q * LP;
        const float s0 = lane < Tn ? row[lane] : -3.0e38f, s1 = lane + 64 < Tn ? row[lane + 64] : -3.0e38f;
        const float mx = wave_max(fmaxf(s0, s1));
        const float p0 = s0 > -1.0e38f ? expf(s0 - mx) : 0.f, p1 = s1 > -1.0e38f ? expf(s1 - mx) : 0.f;
        const float inv = 1.0f / wave_sum(p0 + p1);
        if (lane < Tn) row[lane] = p0 * inv;
        if (lane + 64 < Tn) row[lane + 64] = p1 * inv;
    }
    __syncthreads();
    // dV[k][d] = sum_q P[q][k] dO[q][d];   D[q] = sum_d dO[q][d] * (sum_k P[q][k] V[k][d])
    for (int i0 = 0; i0 < items; i0 += AB_THREADS) {          // uniform trip count: the 16-lane reduction below needs whole groups
        const int i = i0 + tid;
        const bool live = i < items;
        const int r = live ? i >> 4 : 0, c = (i & 15) * 4;
        f32x4 dv = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < Tn; ++j) {
            const float pt = sP[j * LP + r];                // P[q = j][k = r]
            const float pr = sP[r * LP + j];                // P[q = r][k = j]
            const f32x4 g = *(const f32x4*)(sO + j * AB_LD + c), v = *(const f32x4*)(sV + j * AB_LD + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) { dv[e] = fmaf(pt, g[e], dv[e]); o[e] = fmaf(pr, v[e], o[e]); }
        }
        const f32x4 g = *(const f32x4*)(sO + r * AB_LD + c);
        float part = (o[0] * g[0] + o[1] * g[1]) + (o[2] * g[2] + o[3] * g[3]);
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) part += __shfl_xor(part, m);
        if (live) {
            st4<T>(dqb + (int64_t)r * ld_qkv + 2 * d_model + c, dv);
            if ((i & 15) == 0) sD[r] = part;
        }
    }
    __syncthreads();
    // dS[q][k] = scale * P[q][k] * (dO[q] . V[k] - D[q]), in place
    for (int i = tid; i < Tn * Tn; i += AB_THREADS) {
        const int q = i / Tn, k = i - q * Tn;
        const float pv = sP[q * LP + k];
        sP[q * LP + k] = pv != 0.f ? pv * (dot64(sO + q * AB_LD, sV + k * AB_LD) - sD[q]) * scale : 0.f;
    }
    __syncthreads();
    // dQ[r] = sum_k dS[r][k] K[k];   dK[r] = sum_q dS[q][r] Q[q]
    for (int i = tid; i < items; i += AB_THREADS) {
        const int r = i >> 4, c = (i & 15) * 4;
        f32x4 aq = {0.f, 0.f, 0.f, 0.f}, ak = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < Tn; ++j) {
            const float sr = sP[r * LP + j], sc = sP[j * LP + r];
            const f32x4 kk = *(const f32x4*)(sK + j * AB_LD + c), qq = *(const f32x4*)(sQ + j * AB_LD + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) { aq[e] = fmaf(sr, kk[e], aq[e]); ak[e] = fmaf(sc, qq[e], ak[e]); }
        }
        st4<T>(dqb + (int64_t)r * ld_qkv + c, aq);
        st4<T>(dqb + (int64_t)r * ld_qkv + d_model + c, ak);
    }
}

// ---------------------------------------------------------------------------------------------- attention backward, matrix cores (16-bit dtypes)
// The same mathematics as attn_bwd_kernel, T <= 96, for f16 / bf16 operands: the five T x T x 64 contractions run on v_mfma_f32_32x32x16 with fp32
// accumulation, probabilities and score gradients never touch memory.  One 256-thread workgroup per (batch, head); Q, K, V, dO of the head sit in LDS
// as 16-bit row images (96 rows x 144 B: the 16-byte pad keeps the 16-byte fragment reads of 32 consecutive rows off each other's banks; rows past T are zero).
// Both operand patterns of the forward kernel (attention.hip) are reused, each in BOTH orientations, so that a contraction index always lies inside a lane's
// accumulator registers and P / dS go from accumulators straight into the next MFMA's B operand:
//   pass A, wave = 32-query block:  S^T = K Q^T and dP^T = V dO^T (keys on accumulator rows, the query on the lane: row maximum, row sum and
//           D[q] = sum_k P[q][k] dP[q][k] (= dO[q] . O[q]) need one lane exchange);  dS^T = scale P^T o (dP^T - D);  dQ^T = K^T dS^T with K^T fragments from the
//           transposing LDS read (ds_read_b64_tr_b16), as the forward's O^T = V^T P^T.  The block's (maximum, sum, D) go to LDS.
//   pass B, wave = 32-key block:    S = Q K^T and dP = dO V^T (queries on accumulator rows, the key on the lane), P and dS rebuilt with the published row
//           statistics;  dV^T = dO^T P,  dK^T = Q^T dS  (dO^T, Q^T by transposing reads).
// P and dS are rounded to the operand type where they enter an MFMA (as the forward rounds P); everything else is fp32.
constexpr int ABM_PB = 144, ABM_ROWS = 96, ABM_IMG = ABM_ROWS * ABM_PB;
constexpr int ABM_LDS = 4 * ABM_IMG + 3 * ABM_ROWS * 4;

__device__ __forceinline__ float bw_lane32_max(float v) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float bw_lane32_sum(float v) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// lane (c = lane & 31, h = lane >> 5) holds X^T[32 i + (r & 3) + 8 (r >> 2) + 4 h][c] in o[i][r]: row c of X, 64 columns, to global (16-byte stores after
// the two lanes of a row exchanged halves - attention.hip attn_store_block)
template <typename T>
__device__ __forceinline__ void bw_store_rows(const f32x16 (&o)[2], T* op, int fh, bool on) {
    typedef typename VecOf<T>::v4 v4;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    typedef __attribute__((ext_vector_type(4))) int i32x4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            v4 wa, wb;
#pragma unroll
            for (int e = 0; e < 4; ++e) { wa[e] = (T)o[i][8 * k + e]; wb[e] = (T)o[i][8 * k + 4 + e]; }
            const u32x2 a2 = __builtin_bit_cast(u32x2, wa), b2 = __builtin_bit_cast(u32x2, wb);
            const u32x2 s0 = __builtin_amdgcn_permlane32_swap(a2[0], b2[0], false, false);
            const u32x2 s1 = __builtin_amdgcn_permlane32_swap(a2[1], b2[1], false, false);
            i32x4 w;
            w[0] = (int)s0[0]; w[1] = (int)s1[0]; w[2] = (int)s0[1]; w[3] = (int)s1[1];
            if (on) *(i32x4*)(op + 32 * i + 16 * k + 8 * fh) = w;
        }
}

template <typename T>
__global__ __launch_bounds__(256, 2) void attn_bwd_mfma_kernel(const T* __restrict__ qkv, const T* __restrict__ dout, T* __restrict__ dqkv,
                                                               int Tn, int heads, int64_t ld_qkv, int64_t ld_out, float scale, int causal) {
    typedef typename VecOf<T>::v8 v8;
    typedef typename VecOf<T>::v4 v4;
    extern __shared__ __attribute__((aligned(16))) char smb[];
    char* sQ = smb;
    char* sK = sQ + ABM_IMG;
    char* sV = sK + ABM_IMG;
    char* sG = sV + ABM_IMG;                                  // dO
    float* sM = (float*)(sG + ABM_IMG);                       // row maximum of scale * log2(e) * S
    float* sL = sM + ABM_ROWS;                                // 1 / row sum of exp2
    float* sD = sL + ABM_ROWS;                                // D[q]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
    const int d_model = heads * 64;
    const T* base = qkv + (int64_t)b * Tn * ld_qkv + h * 64;
    const T* dob = dout + (int64_t)b * Tn * ld_out + h * 64;
    T* dqb = dqkv + (int64_t)b * Tn * ld_qkv + h * 64;
    // ---- the head's four operands -> LDS images (16-byte chunks; rows past T zero)
#pragma unroll
    for (int it = 0; it < 12; ++it) {
        const int idx = it * 256 + tid;
        const int which = idx / (ABM_ROWS * 8), rem = idx - which * (ABM_ROWS * 8);
        const int r = rem >> 3, c = rem & 7;
        v8 val;
#pragma unroll
        for (int j = 0; j < 8; ++j) val[j] = (T)0.f;
        if (r < Tn) val = which < 3 ? *(const v8*)(base + (int64_t)r * ld_qkv + which * d_model + c * 8) : *(const v8*)(dob + (int64_t)r * ld_out + c * 8);
        *(v8*)(smb + which * ABM_IMG + r * ABM_PB + c * 16) = val;
    }
    __syncthreads();
    const int nb = (Tn + 31) >> 5;                            // 32-row blocks with live rows (<= 3)
    const int fr = lane & 31, fh = lane >> 5, li = lane & 15, dgrp = (lane >> 4) & 1;
    const float c2 = scale * 1.4426950408889634f;
    const int frag_off = fr * ABM_PB + fh * 16;               // fragment row fr, 16-byte chunk 2 s + fh
    const int tr_off = (4 * fh + (li >> 2)) * ABM_PB + 32 * dgrp + 8 * (li & 3);   // transposing read: row 4 fh + (li >> 2), 4 columns at 16 dgrp + 4 (li & 3)

    // ---------------- pass A: this wave's 32 queries against every key
    if (wave < nb) {
        const int qi = wave * 32 + fr;
        v8 qf[4], gf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[s] = *(const v8*)(sQ + wave * 32 * ABM_PB + frag_off + s * 32);
            gf[s] = *(const v8*)(sG + wave * 32 * ABM_PB + frag_off + s * 32);
        }
        f32x16 sc[3], dp[3];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { sc[kt][r] = 0.f; dp[kt][r] = 0.f; }
            if (kt < nb) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const v8 kf = *(const v8*)(sK + kt * 32 * ABM_PB + frag_off + s * 32);
                    const v8 vf = *(const v8*)(sV + kt * 32 * ABM_PB + frag_off + s * 32);
                    sc[kt] = mfma_32x32x16(kf, qf[s], sc[kt]);
                    dp[kt] = mfma_32x32x16(vf, gf[s], dp[kt]);
                }
            }
        }
        // masked softmax over keys (this lane: keys kt*32 + (r & 3) + 8 (r >> 2) + 4 fh of query qi)
        const int klimit = causal ? (qi < Tn - 1 ? qi : Tn - 1) : Tn - 1;
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                sc[kt][r] = key <= klimit ? sc[kt][r] * c2 : -3.0e38f;
                mx = fmaxf(mx, sc[kt][r]);
            }
        mx = bw_lane32_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(sc[kt][r] - mx);   // masked: exp2(-huge) = 0
                sc[kt][r] = pv;
                sum += pv;
            }
        sum = bw_lane32_sum(sum);
        const float inv = 1.0f / sum;
        float dd = 0.f;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { sc[kt][r] *= inv; dd = fmaf(sc[kt][r], dp[kt][r], dd); }
        dd = bw_lane32_sum(dd);
        if (fh == 0) { sM[qi] = mx; sL[qi] = inv; sD[qi] = dd; }
        // dS^T = scale P^T o (dP^T - D);  dQ^T = K^T dS^T
        f32x16 o[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            if (kt < nb) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    v8 pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (T)(scale * sc[kt][8 * s2 + j] * (dp[kt][8 * s2 + j] - dd));
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        v8 kt8;
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const v4 t4 = lds_read_tr16((const T*)(sK + (kt * 32 + 16 * s2 + 8 * u) * ABM_PB + tr_off + 64 * i));
#pragma unroll
                            for (int e = 0; e < 4; ++e) kt8[4 * u + e] = t4[e];
                        }
                        o[i] = mfma_32x32x16(kt8, pf, o[i]);
                    }
                }
            }
        }
        bw_store_rows<T>(o, dqb + (int64_t)(qi < Tn ? qi : 0) * ld_qkv, fh, qi < Tn);
    }
    __syncthreads();
    // ---------------- pass B: this wave's 32 keys against every query
    if (wave < nb) {
        const int key = wave * 32 + fr;
        v8 kf[4], vf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = *(const v8*)(sK + wave * 32 * ABM_PB + frag_off + s * 32);
            vf[s] = *(const v8*)(sV + wave * 32 * ABM_PB + frag_off + s * 32);
        }
        f32x16 ov[2], ok[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { ov[i][r] = 0.f; ok[i][r] = 0.f; }
#pragma unroll
        for (int qt = 0; qt < 3; ++qt) {
            if (qt < nb) {
                f32x16 sc, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const v8 qa = *(const v8*)(sQ + qt * 32 * ABM_PB + frag_off + s * 32);
                    const v8 ga = *(const v8*)(sG + qt * 32 * ABM_PB + frag_off + s * 32);
                    sc = mfma_32x32x16(qa, kf[s], sc);
                    dp = mfma_32x32x16(ga, vf[s], dp);
                }
                // this lane: queries qt*32 + (r & 3) + 8 (r >> 2) + 4 fh against key `key`
                f32x16 ds;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int q0 = qt * 32 + 8 * g4 + 4 * fh;
                    const f32x4 m4 = *(const f32x4*)(sM + q0), l4 = *(const f32x4*)(sL + q0), d4 = *(const f32x4*)(sD + q0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g4 + e, q = q0 + e;
                        const bool valid = key < Tn && (!causal || key <= q);
                        const float pv = valid ? __builtin_amdgcn_exp2f(sc[r] * c2 - m4[e]) * l4[e] : 0.f;
                        sc[r] = pv;
                        ds[r] = scale * pv * (dp[r] - d4[e]);
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    v8 pf, sf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { pf[j] = (T)sc[8 * s2 + j]; sf[j] = (T)ds[8 * s2 + j]; }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        v8 g8, q8;
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int off = (qt * 32 + 16 * s2 + 8 * u) * ABM_PB + tr_off + 64 * i;
                            const v4 tg = lds_read_tr16((const T*)(sG + off)), tq = lds_read_tr16((const T*)(sQ + off));
#pragma unroll
                            for (int e = 0; e < 4; ++e) { g8[4 * u + e] = tg[e]; q8[4 * u + e] = tq[e]; }
                        }
                        ov[i] = mfma_32x32x16(g8, pf, ov[i]);
                        ok[i] = mfma_32x32x16(q8, sf, ok[i]);
                    }
                }
            }
        }
        T* row = dqb + (int64_t)(key < Tn ? key : 0) * ld_qkv;
        bw_store_rows<T>(ok, row + d_model, fh, key < Tn);
        bw_store_rows<T>(ov, row + 2 * d_model, fh, key < Tn);
    }
}

template <typename T>
int ln_bwd_launch(const void* dy, const void* x, const float* gamma, const void* add, void* dx, int64_t rows, int dim, float eps, hipStream_t s) {
    hipLaunchKernelGGL((layernorm_bwd_kernel<T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (const T*)dy, (const T*)x, gamma,
                       (const T*)add, (T*)dx, rows, dim, eps);
    return leclip_check_launch("layernorm_bwd_kernel");
}
template <typename T>
int gelu_launch(const void* pre, const void* du, void* out, int64_t n, bool bwd, hipStream_t s) {
    const int64_t n4 = n / 4;
    const dim3 grid((unsigned)((n4 + 255) / 256)), block(256);
    if (bwd) hipLaunchKernelGGL((quickgelu_kernel<T, true>), grid, block, 0, s, (const T*)pre, (const T*)du, (T*)out, n4);
    else hipLaunchKernelGGL((quickgelu_kernel<T, false>), grid, block, 0, s, (const T*)pre, (const T*)du, (T*)out, n4);
    return leclip_check_launch("quickgelu_kernel");
}
template <typename T>
int attn_bwd_mfma_launch(const void* qkv, const void* dout, void* dqkv, int64_t B, int Tn, int heads, int64_t ld_qkv, int64_t ld_out, float scale,
                         int causal, hipStream_t s) {
    static bool attr_set[LECLIP_MAX_DEVICES] = {};
    leclip_set_max_lds(attn_bwd_mfma_kernel<T>, ABM_LDS, attr_set);
    hipLaunchKernelGGL((attn_bwd_mfma_kernel<T>), dim3((unsigned)(B * heads)), dim3(256), ABM_LDS, s, (const T*)qkv, (const T*)dout, (T*)dqkv, Tn, heads,
                       ld_qkv, ld_out, scale, causal);
    return leclip_check_launch("attn_bwd_mfma_kernel");
}
template <typename T>
int attn_bwd_launch(const void* qkv, const void* dout, void* dqkv, int64_t B, int Tn, int heads, int64_t ld_qkv, int64_t ld_out, float scale,
                    int causal, hipStream_t s) {
    const size_t lds = ((size_t)4 * Tn * AB_LD + (size_t)Tn * (Tn + 1) + Tn) * sizeof(float);
    static bool attr_set[LECLIP_MAX_DEVICES] = {};
    leclip_set_max_lds(attn_bwd_kernel<T>, 160 * 1024, attr_set);
    hipLaunchKernelGGL((attn_bwd_kernel<T>), dim3((unsigned)(B * heads)), dim3(AB_THREADS), lds, s, (const T*)qkv, (const T*)dout, (T*)dqkv, Tn, heads,
                       ld_qkv, ld_out, scale, causal);
    return leclip_check_launch("attn_bwd_kernel");
}

}  // namespace

extern "C" int leclip_layernorm_bwd(const void* dy, const void* x, const float* gamma, const void* add, void* dx, int64_t rows, int dim,
                                    float eps, leclip_dtype dtype, void* stream) {
    if (!dy || !x || !gamma || !dx || rows <= 0 || dim <= 0 || !dtype_ok(dtype)) { leclip_set_error("layernorm_bwd: bad argument"); return LECLIP_E_INVALID; }
    if (dim % 64 != 0 || dim > 256 * LNB_MAXV) { leclip_set_error("layernorm_bwd: dim=%d must be a multiple of 64 and <= %d", dim, 256 * LNB_MAXV); return LECLIP_E_UNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    if (dtype == LECLIP_F32) return ln_bwd_launch<float>(dy, x, gamma, add, dx, rows, dim, eps, s);
    if (dtype == LECLIP_F16) return ln_bwd_launch<f16_t>(dy, x, gamma, add, dx, rows, dim, eps, s);
    return ln_bwd_launch<bf16_t>(dy, x, gamma, add, dx, rows, dim, eps, s);
}

extern "C" int leclip_quickgelu_fwd(const void* pre, void* out, int64_t n, leclip_dtype dtype, void* stream) {
    if (!pre || !out || n <= 0 || n % 4 || !dtype_ok(dtype)) { leclip_set_error("quickgelu_fwd: bad argument (n must be a multiple of 4)"); return LECLIP_E_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    if (dtype == LECLIP_F32) return gelu_launch<float>(pre, nullptr, out, n, false, s);
    if (dtype == LECLIP_F16) return gelu_launch<f16_t>(pre, nullptr, out, n, false, s);
    return gelu_launch<bf16_t>(pre, nullptr, out, n, false, s);
}

extern "C" int leclip_quickgelu_bwd(const void* pre, const void* du, void* dpre, int64_t n, leclip_dtype dtype, void* stream) {
    if (!pre || !du || !dpre || n <= 0 || n % 4 || !dtype_ok(dtype)) { leclip_set_error("quickgelu_bwd: bad argument (n must be a multiple of 4)"); return LECLIP_E_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    if (dtype == LECLIP_F32) return gelu_launch<float>(pre, du, dpre, n, true, s);
    if (dtype == LECLIP_F16) return gelu_launch<f16_t>(pre, du, dpre, n, true, s);
    return gelu_launch<bf16_t>(pre, du, dpre, n, true, s);
}

extern "C" int leclip_attention_bwd(const void* qkv, const void* dout, void* dqkv, int64_t B, int T, int heads, int head_dim, int64_t ld_qkv,
                                    int64_t ld_out, leclip_mask mask, float scale, leclip_dtype dtype, void* stream) {
    if (!qkv || !dout || !dqkv || B <= 0 || T <= 0 || heads <= 0 || ld_qkv < 3 * heads * 64 || ld_out < heads * 64 || !dtype_ok(dtype)) {
        leclip_set_error("attention_bwd: bad argument"); return LECLIP_E_INVALID;
    }
    if ((ld_qkv % 4) || (ld_out % 4) || ((uintptr_t)qkv & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dqkv & 15)) {
        leclip_set_error("attention_bwd: qkv / dout / dqkv must be 16-byte aligned with leading dimensions that are multiples of 4");
        return LECLIP_E_INVALID;
    }
    if (head_dim != 64 || T > AB_TMAX) { leclip_set_error("attention_bwd: head_dim must be 64 and T <= %d (text tower)", AB_TMAX); return LECLIP_E_UNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    const int causal = mask == LECLIP_MASK_CAUSAL;
    if (dtype == LECLIP_F32) return attn_bwd_launch<float>(qkv, dout, dqkv, B, T, heads, ld_qkv, ld_out, scale, causal, s);
    // 16-bit operands, T <= 96, 16-byte rows: the matrix-core kernel (fp32: the vector kernel, whose arithmetic is fp32 throughout)
    const bool mfma_ok = T <= ABM_ROWS && ld_qkv % 8 == 0 && ld_out % 8 == 0;
    if (dtype == LECLIP_F16) return mfma_ok ? attn_bwd_mfma_launch<f16_t>(qkv, dout, dqkv, B, T, heads, ld_qkv, ld_out, scale, causal, s)
                                            : attn_bwd_launch<f16_t>(qkv, dout, dqkv, B, T, heads, ld_qkv, ld_out, scale, causal, s);
    return mfma_ok ? attn_bwd_mfma_launch<bf16_t>(qkv, dout, dqkv, B, T, heads, ld_qkv, ld_out, scale, causal, s)
                   : attn_bwd_launch<bf16_t>(qkv, dout, dqkv, B, T, heads, ld_qkv, ld_out, scale, causal, s);
}
