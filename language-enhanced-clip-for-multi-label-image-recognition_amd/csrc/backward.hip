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
// One 256-thread workgroup per (batch, head), T <= 104, head_dim 64.  Q, K, V, dO of the head in LDS as fp32 (rows
// padded to 65), probabilities recomputed:  P = softmax(scale QK^T + mask);  dV = P^T dO;  dP = dO V^T;
// dS = P o (dP - rowsum(P o dP));  dQ = scale dS K;  dK = scale dS^T Q.   fp32 arithmetic for every I/O dtype.
constexpr int AB_TMAX = 104;   // 4 x T x 65 + T x (T+1) floats of LDS <= 160 KiB

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ dout, T* __restrict__ dqkv,
                                                       int Tn, int heads, int64_t ld_qkv, int64_t ld_out, float scale, int causal) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sQ = sm;
    float* sK = sQ + Tn * 65;
    float* sV = sK + Tn * 65;
    float* sO = sV + Tn * 65;          // dO
    float* sP = sO + Tn * 65;          // [T][T+1]  P, then dS
    const int LP = Tn + 1;
    const int tid = threadIdx.x;
    const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
    const int d_model = heads * 64;
    const T* base = qkv + (int64_t)b * Tn * ld_qkv + h * 64;
    const T* dob = dout + (int64_t)b * Tn * ld_out + h * 64;
    for (int i = tid; i < Tn * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        sQ[r * 65 + c] = (float)base[(int64_t)r * ld_qkv + c];
        sK[r * 65 + c] = (float)base[(int64_t)r * ld_qkv + d_model + c];
        sV[r * 65 + c] = (float)base[(int64_t)r * ld_qkv + 2 * d_model + c];
        sO[r * 65 + c] = (float)dob[(int64_t)r * ld_out + c];
    }
    __syncthreads();
    // scores
    for (int i = tid; i < Tn * Tn; i += 256) {
        const int q = i / Tn, k = i - q * Tn;
        float s = -3.0e38f;
        if (!causal || k <= q) {
            float dot = 0.f;
            for (int d = 0; d < 64; ++d) dot = fmaf(sQ[q * 65 + d], sK[k * 65 + d], dot);
            s = dot * scale;
        }
        sP[q * LP + k] = s;
    }
    __syncthreads();
    // row softmax (one thread per query row)
    for (int q = tid; q < Tn; q += 256) {
        float mx = -3.0e38f;
        for (int k = 0; k < Tn; ++k) mx = fmaxf(mx, sP[q * LP + k]);
        float sum = 0.f;
        for (int k = 0; k < Tn; ++k) { const float p = sP[q * LP + k] > -1.0e38f ? expf(sP[q * LP + k] - mx) : 0.f; sP[q * LP + k] = p; sum += p; }
        const float inv = 1.0f / sum;
        for (int k = 0; k < Tn; ++k) sP[q * LP + k] *= inv;
    }
    __syncthreads();
    // dV[k][d] = sum_q P[q][k] dO[q][d]
    T* dqb = dqkv + (int64_t)b * Tn * ld_qkv + h * 64;
    for (int i = tid; i < Tn * 64; i += 256) {
        const int k = i >> 6, d = i & 63;
        float acc = 0.f;
        for (int q = 0; q < Tn; ++q) acc = fmaf(sP[q * LP + k], sO[q * 65 + d], acc);
        dqb[(int64_t)k * ld_qkv + 2 * d_model + d] = (T)acc;
    }
    __syncthreads();
    // dS = P o (dP - D),  dP[q][k] = dO[q] . V[k],  D[q] = sum_k P[q][k] dP[q][k]   (one thread per row keeps it simple)
    for (int q = tid; q < Tn; q += 256) {
        float D = 0.f;
        for (int k = 0; k < Tn; ++k) {
            float dp = 0.f;
            for (int d = 0; d < 64; ++d) dp = fmaf(sO[q * 65 + d], sV[k * 65 + d], dp);
            D = fmaf(sP[q * LP + k], dp, D);
        }
        for (int k = 0; k < Tn; ++k) {
            float dp = 0.f;
            for (int d = 0; d < 64; ++d) dp = fmaf(sO[q * 65 + d], sV[k * 65 + d], dp);
            sP[q * LP + k] = sP[q * LP + k] * (dp - D) * scale;
        }
    }
    __syncthreads();
    for (int i = tid; i < Tn * 64; i += 256) {
        const int r = i >> 6, d = i & 63;
        float aq = 0.f, ak = 0.f;
        for (int j = 0; j < Tn; ++j) {
            aq = fmaf(sP[r * LP + j], sK[j * 65 + d], aq);     // dQ[r] = sum_k dS[r][k] K[k]
            ak = fmaf(sP[j * LP + r], sQ[j * 65 + d], ak);     // dK[r] = sum_q dS[q][r] Q[q]
        }
        dqb[(int64_t)r * ld_qkv + d] = (T)aq;
        dqb[(int64_t)r * ld_qkv + d_model + d] = (T)ak;
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
    const size_t lds = ((size_t)4 * Tn * 65 + (size_t)Tn * (Tn + 1)) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((attn_bwd_kernel<T>), dim3((unsigned)(B * heads)), dim3(256), lds, s, (const T*)qkv, (const T*)dout, (T*)dqkv, Tn, heads,
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
    if (head_dim != 64 || T > AB_TMAX) { leclip_set_error("attention_bwd: head_dim must be 64 and T <= %d (text tower)", AB_TMAX); return LECLIP_E_UNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    const int causal = mask == LECLIP_MASK_CAUSAL;
    if (dtype == LECLIP_F32) return attn_bwd_launch<float>(qkv, dout, dqkv, B, T, heads, ld_qkv, ld_out, scale, causal, s);
    if (dtype == LECLIP_F16) return attn_bwd_launch<f16_t>(qkv, dout, dqkv, B, T, heads, ld_qkv, ld_out, scale, causal, s);
    return attn_bwd_launch<bf16_t>(qkv, dout, dqkv, B, T, heads, ld_qkv, ld_out, scale, causal, s);
}
