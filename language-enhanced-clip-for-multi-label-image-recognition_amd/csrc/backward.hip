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
    if (dtype == LECLIP_F16) return attn_bwd_launch<f16_t>(qkv, dout, dqkv, B, T, heads, ld_qkv, ld_out, scale, causal, s);
    return attn_bwd_launch<bf16_t>(qkv, dout, dqkv, B, T, heads, ld_qkv, ld_out, scale, causal, s);
}
