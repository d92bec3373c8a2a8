// The tail of the scoring path as ONE kernel: class-token gather -> ln_post -> @ proj -> L2-normalise -> scale * img . txt^T
// (clip/model.py:271-274, 399-404; trainers/Caption_distill_double.py:330-335), both contractions on the matrix cores:
//   * the projection [16 images, d] x [E, d]^T with v_mfma_f32_16x16x32 (bf16 / fp16 operands, the same instruction and
//     ascending-K order as the GEMM kernels) or, in the fp32 parity mode, the exact v_mfma_f32_16x16x4_f32;
//   * the cosine-logit contraction [16, E] x [C, E]^T on the fp32 features with v_mfma_f32_16x16x4_f32 (exact fp32: the
//     features are never rounded to 16 bits), divided by the two norms afterwards like the reference's normalise-then-dot.
// One 256-thread workgroup per 16 images; LayerNorm output and features live in LDS; proj^T and the text features are read
// straight from L2 (0.8 MB + 0.16 MB, shared by every workgroup).  Replaces three launches (LayerNorm, projection GEMM,
// one-thread-per-logit kernel) of round 1.  Also here: the small row-wise helpers of the prompt-tuning backward
// (cosine-logit backward w.r.t. the text features, row gather / scatter).
#include "leclip_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float acc4;
__device__ __forceinline__ acc4 mfma16(bf16x8 a, bf16x8 b, acc4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ acc4 mfma16(f16x8 a, f16x8 b, acc4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

template <typename T> __device__ __forceinline__ f32x4 tload4(const T* p);
template <> __device__ __forceinline__ f32x4 tload4<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 tload4<bf16_t>(const bf16_t* p) {
    const bf16x4 v = *(const bf16x4*)p; f32x4 r; for (int i = 0; i < 4; ++i) r[i] = (float)v[i]; return r;
}
template <> __device__ __forceinline__ f32x4 tload4<f16_t>(const f16_t* p) {
    const f16x4 v = *(const f16x4*)p; f32x4 r; for (int i = 0; i < 4; ++i) r[i] = (float)v[i]; return r;
}
template <typename T> __device__ __forceinline__ void tstore4(T* p, f32x4 v);
template <> __device__ __forceinline__ void tstore4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void tstore4<bf16_t>(bf16_t* p, f32x4 v) { bf16x4 r; for (int i = 0; i < 4; ++i) r[i] = (bf16_t)v[i]; *(bf16x4*)p = r; }
template <> __device__ __forceinline__ void tstore4<f16_t>(f16_t* p, f32x4 v) { f16x4 r; for (int i = 0; i < 4; ++i) r[i] = (f16_t)v[i]; *(f16x4*)p = r; }

struct TailArgs {
    const void* x;          // residual stream; image b's class-token row starts at x + b * row_stride (elements)
    const float* gamma;
    const float* beta;
    const void* proj_t;     // [E, d], K contiguous, same dtype as x
    const float* txt;       // [C, E] fp32 text features (un-normalised) or null (features only)
    float* feat;            // [B, E] fp32 or null
    float* logits;          // [B, C] fp32 or null
    int64_t B, row_stride;
    int d, E, C;
    float eps, scale;
};

constexpr int TAIL_MAXV = 4;   // d <= 1024

// Workgroup shape: TAIL_NW waves (1 024 threads) share 16 images.  The kernel is a latency chain - strided class rows from HBM, then
// proj^T (0.8 MB) and the text features from L2 - run by ceil(B / 16) workgroups, so what shortens it is more loads in flight per
// workgroup: 16 waves take one LayerNorm row each, 2 - 3 projection column tiles each with the B fragments FOUR k-steps ahead, one
// logit tile each (round 2: 58 us -> see profiles).
constexpr int TAIL_NW = 16;
// proj GEMM for one wave: column tiles wave, wave + TAIL_NW, ... (16 columns each, at most PT_MAX per wave), A = LayerNorm output
// in LDS.  K is the OUTER loop: the B fragments of all of the wave's tiles for a k-step are independent 16-byte loads straight
// from L2 (proj^T is shared by every workgroup), requested PT_DEPTH - 1 k-steps ahead of the MFMAs that consume them.
constexpr int PT_MAX = 3;     // E <= 768 per 16 waves
constexpr int PT_DEPTH = 4;
template <typename T>
__device__ __forceinline__ void proj_tiles(const TailArgs& a, const char* sh, int hpitch, float* sfeat, int fpitch, int wave, int lane) {
    typedef typename VecOf<T>::v8 v8;
    const int fr = lane & 15, fc = lane >> 4;
    const int ntiles = a.E >> 4;
    const int mine = (ntiles - wave + TAIL_NW - 1) / TAIL_NW;    // tiles of this wave
    const T* prow = (const T*)a.proj_t + (int64_t)(wave * 16 + fr) * a.d + fc * 8;
    const int64_t tstride = (int64_t)(16 * TAIL_NW) * a.d;       // wave's next tile: TAIL_NW tiles further
    const char* hrow = sh + fr * hpitch + fc * 16;
    const int nsteps = a.d >> 5;                                 // k-steps of 32
    acc4 acc[PT_MAX];
    v8 bq[PT_DEPTH][PT_MAX];
#pragma unroll
    for (int i = 0; i < PT_MAX; ++i) acc[i] = acc4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < PT_DEPTH - 1; ++u)
        if (u < nsteps) {
#pragma unroll
            for (int i = 0; i < PT_MAX; ++i) if (i < mine) bq[u][i] = *(const v8*)(prow + i * tstride + u * 32);
        }
    for (int s0 = 0; s0 < nsteps; s0 += PT_DEPTH) {
#pragma unroll
        for (int u = 0; u < PT_DEPTH; ++u) {                     // static ring slots: step s0 + u lives in slot u
            const int st = s0 + u;
            if (st >= nsteps) break;
            if (st + PT_DEPTH - 1 < nsteps) {
#pragma unroll
                for (int i = 0; i < PT_MAX; ++i)
                    if (i < mine) bq[(u + PT_DEPTH - 1) % PT_DEPTH][i] = *(const v8*)(prow + i * tstride + (st + PT_DEPTH - 1) * 32);
            }
            const v8 af = *(const v8*)(hrow + st * 64);
#pragma unroll
            for (int i = 0; i < PT_MAX; ++i) if (i < mine) acc[i] = mfma16(af, bq[u][i], acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < PT_MAX; ++i)
        if (i < mine) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sfeat[(4 * fc + r) * fpitch + (wave + TAIL_NW * i) * 16 + fr] = acc[i][r];
        }
}
template <>
__device__ __forceinline__ void proj_tiles<float>(const TailArgs& a, const char* sh, int hpitch, float* sfeat, int fpitch, int wave, int lane) {
    // exact fp32: v_mfma_f32_16x16x4_f32, K taken in groups of 16 with MFMA j / slot q <-> k = 16g + 4q + j, so that a lane's
    // four operands of a group are 16 contiguous bytes on both sides
    const int fr = lane & 15, fq = lane >> 4;
    const float* P = (const float*)a.proj_t;
    for (int nt = wave; nt * 16 < a.E; nt += TAIL_NW) {
        acc4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* prow = P + (int64_t)(nt * 16 + fr) * a.d + 4 * fq;
        const char* hrow = sh + fr * hpitch + 16 * fq;
#pragma unroll 4
        for (int k = 0; k < a.d; k += 16) {
            const f32x4 af = *(const f32x4*)(hrow + k * 4);
            const f32x4 bf = *(const f32x4*)(prow + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sfeat[(4 * fq + r) * fpitch + nt * 16 + fr] = acc[r];
    }
}

template <typename T>
__global__ __launch_bounds__(TAIL_NW * 64) void image_tail_kernel(TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hpitch = a.d * (int)sizeof(T) + 16;          // +16 B: the 16 rows of a fragment read fall on distinct banks
    const int fpitch = a.E + 4;
    char* sh = smem;                                       // [16][hpitch]   LayerNorm output, operand dtype
    float* sfeat = (float*)(smem + 16 * hpitch);           // [16][E + 4]    projected features, fp32
    float* sinorm = sfeat + 16 * fpitch;                   // [16]           |feature|^2
    float* stnorm = sinorm + 16;                           // [C]            |text feature|^2
    const int64_t b0 = (int64_t)blockIdx.x * 16;

    // ---- ln_post on the class-token rows: one wave per row; two-pass fp32 statistics (as layernorm_kernel)
    const int nv = a.d >> 8, tail = a.d & 255;
    constexpr int RPW = 16 / TAIL_NW;   // rows per wave
    for (int rr = 0; rr < RPW; ++rr) {
        const int row = wave * RPW + rr;
        int64_t b = b0 + row;
        b = b < a.B ? b : a.B - 1;
        const T* xr = (const T*)a.x + b * a.row_stride;
        f32x4 v[TAIL_MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < TAIL_MAXV; ++i)
            if (i < nv || (i == nv && lane * 4 < tail)) { v[i] = tload4<T>(xr + i * 256 + lane * 4); s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
        const float mean = wave_sum(s) / (float)a.d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < TAIL_MAXV; ++i)
            if (i < nv || (i == nv && lane * 4 < tail)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float dlt = v[i][e] - mean; q = fmaf(dlt, dlt, q); }
            }
        const float rstd = rsqrtf(wave_sum(q) / (float)a.d + a.eps);
#pragma unroll
        for (int i = 0; i < TAIL_MAXV; ++i)
            if (i < nv || (i == nv && lane * 4 < tail)) {
                const int c = i * 256 + lane * 4;
                const f32x4 g = *(const f32x4*)(a.gamma + c), bt = *(const f32x4*)(a.beta + c);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + bt[e];
                tstore4<T>((T*)(sh + row * hpitch) + c, o);
            }
    }
    // |t_c|^2 of the text features (every workgroup needs all C of them: 0.16 MB from L2); four rows per wave in flight
    if (a.txt) {
        for (int c0 = wave * 4; c0 < a.C; c0 += 4 * TAIL_NW) {
            float s4[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = lane * 4; k < a.E; k += 256) {
                f32x4 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int c = c0 + u < a.C ? c0 + u : a.C - 1; t[u] = *(const f32x4*)(a.txt + (int64_t)c * a.E + k); }
#pragma unroll
                for (int u = 0; u < 4; ++u) { s4[u] = fmaf(t[u][0], t[u][0], s4[u]); s4[u] = fmaf(t[u][1], t[u][1], s4[u]); s4[u] = fmaf(t[u][2], t[u][2], s4[u]); s4[u] = fmaf(t[u][3], t[u][3], s4[u]); }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const float r = wave_sum(s4[u]); if (lane == 0 && c0 + u < a.C) stnorm[c0 + u] = r; }
        }
    }
    __syncthreads();

    // ---- features = LN(x_cls) @ proj on the matrix cores
    proj_tiles<T>(a, sh, hpitch, sfeat, fpitch, wave, lane);
    __syncthreads();
    for (int rr = 0; rr < RPW; ++rr) {
        const int row = wave * RPW + rr;
        const float* fr_ = sfeat + row * fpitch;
        float s = 0.f;
        for (int k = lane * 4; k < a.E; k += 256) {
            const f32x4 t = *(const f32x4*)(fr_ + k);
            s = fmaf(t[0], t[0], s); s = fmaf(t[1], t[1], s); s = fmaf(t[2], t[2], s); s = fmaf(t[3], t[3], s);
            if (a.feat && b0 + row < a.B) *(f32x4*)(a.feat + (b0 + row) * a.E + k) = t;
        }
        s = wave_sum(s);
        if (lane == 0) sinorm[row] = s;
    }
    if (!a.txt || !a.logits) return;
    __syncthreads();

    // ---- logits[b, c] = scale * <f_b, t_c> / (|f_b| |t_c|): exact-fp32 MFMA, K in groups of 16 (slot q <-> k = 16g + 4q + j)
    const int fr = lane & 15, fq = lane >> 4;
    for (int ct = wave; ct * 16 < a.C; ct += TAIL_NW) {
        int c = ct * 16 + fr;
        const int cc = c < a.C ? c : a.C - 1;
        const float* trow = a.txt + (int64_t)cc * a.E + 4 * fq;
        const float* frow = sfeat + fr * fpitch + 4 * fq;
        acc4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int k = 0; k < a.E; k += 16) {
            const f32x4 af = *(const f32x4*)(frow + k);
            const f32x4 bf = *(const f32x4*)(trow + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], acc, 0, 0, 0);
        }
        if (c < a.C) {
            const float tn = sqrtf(stnorm[c]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * fq + r;
                if (b0 + row < a.B) a.logits[(b0 + row) * a.C + c] = (a.scale * acc[r]) / (sqrtf(sinorm[row]) * tn);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// d txt of logits = scale * normalize(img) . normalize(txt)^T (image features frozen):
//   g_c = scale * sum_b dlogits[b, c] * img_b / |img_b|,   dtxt_c = (g_c - that_c <g_c, that_c>) / |txt_c|
// One workgroup per class; the image rows' inverse norms are formed once per workgroup in LDS (chunks of 2048 images).
constexpr int LB_CHUNK = 2048;
__global__ __launch_bounds__(256) void logits_bwd_kernel(const float* __restrict__ img, const float* __restrict__ txt,
                                                         const float* __restrict__ dlogits, float* __restrict__ dtxt, int64_t B, int C,
                                                         int D, float scale) {
    __shared__ float invn[LB_CHUNK];
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x;
    float g[4] = {0.f, 0.f, 0.f, 0.f};      // D <= 1024: thread owns d = tid, tid + 256, ...
    for (int64_t bb = 0; bb < B; bb += LB_CHUNK) {
        const int nb = (int)(B - bb < LB_CHUNK ? B - bb : LB_CHUNK);
        __syncthreads();
        for (int i = wave; i < nb; i += 4) {
            const float* r = img + (bb + i) * D;
            float s = 0.f;
            for (int k = lane; k < D; k += 64) s = fmaf(r[k], r[k], s);
            s = wave_sum(s);
            if (lane == 0) invn[i] = 1.0f / sqrtf(s);
        }
        __syncthreads();
        int i = 0;
        for (; i + 8 <= nb; i += 8) {          // eight images' rows in flight per thread (the loop is a chain of L2 round trips otherwise)
            float w[8], v[8][4];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                w[j] = dlogits[(bb + i + j) * C + c] * invn[i + j];
                const float* r = img + (bb + i + j) * D;
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int k = tid + 256 * u; v[j][u] = k < D ? r[k] : 0.f; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int u = 0; u < 4; ++u) g[u] = fmaf(w[j], v[j][u], g[u]);
        }
        for (; i < nb; ++i) {
            const float w = dlogits[(bb + i) * C + c] * invn[i];
            const float* r = img + (bb + i) * D;
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int k = tid + 256 * u; if (k < D) g[u] = fmaf(w, r[k], g[u]); }
        }
    }
    const float* t = txt + (int64_t)c * D;
    float tt = 0.f, gt = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int k = tid + 256 * u; if (k < D) { g[u] *= scale; tt = fmaf(t[k], t[k], tt); gt = fmaf(g[u], t[k], gt); } }
    tt = wave_sum(tt);
    gt = wave_sum(gt);
    __syncthreads();
    if (lane == 0) { red[wave] = tt; red[4 + wave] = gt; }
    __syncthreads();
    tt = (red[0] + red[1]) + (red[2] + red[3]);
    gt = (red[4] + red[5]) + (red[6] + red[7]);
    const float tn = sqrtf(tt);
    // that = t / tn;  <g, that> = gt / tn;  dtxt = (g - that * gt / tn) / tn
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int k = tid + 256 * u; if (k < D) dtxt[(int64_t)c * D + k] = (g[u] - t[k] * (gt / tt)) / tn; }
}

// dst[i, :] = src[index[i], :]   /   dst[index[i], :] = src[i, :]   (rows of `dim` elements of `esize` bytes, 16-byte chunks)
__global__ void move_rows_kernel(const char* __restrict__ src, const int64_t* __restrict__ index, char* __restrict__ dst, int64_t n,
                                 int row_bytes, int64_t src_pitch, int64_t dst_pitch, int scatter) {
    const int64_t i = blockIdx.x;
    if (i >= n) return;
    const int64_t j = index[i];
    const char* s = src + (scatter ? i : j) * src_pitch;
    char* d = dst + (scatter ? j : i) * dst_pitch;
    if ((row_bytes & 15) == 0) {
        for (int k = threadIdx.x * 16; k < row_bytes; k += blockDim.x * 16) *(i32x4*)(d + k) = *(const i32x4*)(s + k);
    } else {      // 8-byte granules (the slot-major LayerNorm partials are rows of one (sum, M2) pair)
        for (int k = threadIdx.x * 8; k < row_bytes; k += blockDim.x * 8) *(f32x2*)(d + k) = *(const f32x2*)(s + k);
    }
}

// x[r, :] /= |x[r, :]|_2 in place (fp32 rows; one wave per row) - the patch-token features of the local branch
__global__ __launch_bounds__(256) void l2norm_rows_kernel(float* __restrict__ x, int64_t rows, int dim, int64_t ld) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float* xr = x + r * ld;
    float s = 0.f;
    for (int k = lane * 4; k < dim; k += 256) { const f32x4 t = *(const f32x4*)(xr + k); s = fmaf(t[0], t[0], s); s = fmaf(t[1], t[1], s); s = fmaf(t[2], t[2], s); s = fmaf(t[3], t[3], s); }
    const float inv = 1.0f / sqrtf(wave_sum(s));
    for (int k = lane * 4; k < dim; k += 256) { f32x4 t = *(const f32x4*)(xr + k); t[0] *= inv; t[1] *= inv; t[2] *= inv; t[3] *= inv; *(f32x4*)(xr + k) = t; }
}

// dst [cols][ld_dst] = src [rows][ld_src]^T (fp32, 32 x 32 tiles through LDS): the K-contiguous form of the position features for the
// contraction over positions in the local branch's backward.
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t rows, int cols,
                                                            int64_t ld_src, int64_t ld_dst) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const int64_t r0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t r = r0 + ty + 8 * k;
        tile[ty + 8 * k][tx] = (r < rows && c0 + tx < cols) ? src[r * ld_src + c0 + tx] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k;
        const int64_t r = r0 + tx;
        if (c < cols && r < rows) dst[(int64_t)c * ld_dst + r] = tile[tx][ty + 8 * k];
    }
}

// Backward of y = x / |x| per row (fp32): dx = (dy - yhat <yhat, dy>) / |x|, one wave per row - from the gradient w.r.t. the
// NORMALISED text features (what the similarity GEMM's backward yields) to the gradient w.r.t. the text tower's output.
__global__ __launch_bounds__(256) void l2norm_rows_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                                              int64_t rows, int dim) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* xr = x + r * dim;
    const float* gr = dy + r * dim;
    float xx = 0.f, xg = 0.f;
    for (int k = lane; k < dim; k += 64) { xx = fmaf(xr[k], xr[k], xx); xg = fmaf(xr[k], gr[k], xg); }
    xx = wave_sum(xx);
    xg = wave_sum(xg);
    const float nrm = sqrtf(xx);
    for (int k = lane; k < dim; k += 64) dx[r * dim + k] = (gr[k] - xr[k] * (xg / xx)) / nrm;
}

// Local (dense) branch pooling, trainers/Caption_distill_double.py:447-462 (image branch: patches in place of the ResNet's HxW
// positions) and :493-513 (caption-as-image training branch: the 77 token positions, `text_mask` added to both panels):
//   s[p, c]  = <position feature p, "negative" prompt c> (both normalised) + bias[p], e[p, c] the same against the evidence prompts,
//              bias[p] = -10000 where the caption's token id is 0 (:491, :497-498, :505-506), else 0
//   evidence: w = softmax_c(tmp * s * (max_c s + 1));  s <- s * w;  prob = softmax_p(tmp * e)      (winner-take-all)
//   else:     prob = softmax_p(tmp * s)
//   logits_local[c] = sum_p logit_scale * s[p, c] * prob[p, c]
// (the winner-take-all softmax is evaluated as exp(k (s - s_ext)) / sum, s_ext = the element with the largest k s - the row maximum, or
// the minimum where k < 0: under a masked position k s is ~5e9, where k s - max(k s) loses everything to rounding - and a contracted
// fma(k, s, -zmax) even yields exp(+256); the difference of the similarities themselves is exact.)
// One workgroup per image.  The classes are processed CT at a time (the softmax over p is independent per class; the
// winner-take-all row softmax needs only a per-position scale k, maximum and denominator, taken in a first pass straight
// from global memory), so the LDS footprint is P * CT * 8 bytes whatever P and C are (ViT-L/14@336: P = 576).
__global__ __launch_bounds__(256) void local_pool_kernel(const float* __restrict__ sim, const int64_t* __restrict__ mask_tok, float* __restrict__ out,
                                                         int P, int C, int CT, int64_t ld, int64_t image_stride, int64_t mask_stride, int evi_off,
                                                         float tmp, float logit_scale) {
    extern __shared__ float sm[];      // k[P] | s_ext[P] | den[P] | bias[P] | s [P][CT] (| e [P][CT])
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* src = sim + (int64_t)blockIdx.x * image_stride;
    float *rk = sm, *rz = sm + P, *rd = sm + 2 * P, *bias = sm + 3 * P, *s = sm + 4 * P, *e = s + P * CT;
    for (int p = tid; p < P; p += 256) bias[p] = (mask_tok && mask_tok[(int64_t)blockIdx.x * mask_stride + p] == 0) ? -10000.0f : 0.0f;
    __syncthreads();
    if (evi_off >= 0) {
        for (int p = wave; p < P; p += 4) {      // per position: scale, maximum and denominator of softmax_c(tmp * s * (max + 1))
            const float* row = src + (int64_t)p * ld;
            const float b = bias[p];
            float mx = -INFINITY, mn = INFINITY;
            for (int c = lane; c < C; c += 64) { mx = fmaxf(mx, row[c] + b); mn = fminf(mn, row[c] + b); }
            mx = wave_max(mx);
            mn = -wave_max(-mn);
            const float k = tmp * (mx + 1.0f);
            const float vext = k >= 0.f ? mx : mn;      // the element whose k * s is largest
            float den = 0.f;
            for (int c = lane; c < C; c += 64) den += expf(k * ((row[c] + b) - vext));
            den = wave_sum(den);
            if (lane == 0) { rk[p] = k; rz[p] = vext; rd[p] = den; }
        }
        __syncthreads();
    }
    for (int c0 = 0; c0 < C; c0 += CT) {
        const int ct = C - c0 < CT ? C - c0 : CT;
        for (int i = tid; i < P * ct; i += 256) {
            const int p = i / ct, c = i - p * ct;
            const float v = src[(int64_t)p * ld + c0 + c] + bias[p];
            if (evi_off >= 0) {
                s[p * CT + c] = v * (expf(rk[p] * (v - rz[p])) / rd[p]);
                e[p * CT + c] = src[(int64_t)p * ld + evi_off + c0 + c] + bias[p];
            } else {
                s[p * CT + c] = v;
            }
        }
        __syncthreads();
        const float* z = evi_off >= 0 ? e : s;
        for (int c = tid; c < ct; c += 256) {    // per class: softmax over positions, weighted sum
            float mx = tmp * z[c];
            for (int p = 1; p < P; ++p) mx = fmaxf(mx, tmp * z[p * CT + c]);
            float den = 0.f, num = 0.f;
            for (int p = 0; p < P; ++p) {
                const float w = expf(tmp * z[p * CT + c] - mx);
                den += w;
                num = fmaf(w, s[p * CT + c], num);
            }
            out[(int64_t)blockIdx.x * C + c0 + c] = logit_scale * num / den;
        }
        __syncthreads();
    }
}

// Backward of the pooling w.r.t. both similarity panels (prompt tuning through the local branch, reference :806-808: the ranking loss
// on `output_local`).  dneg / devi [B*P, C] contiguous: d loss / d s, d loss / d e (before the bias, which is a constant).
//   no evidence:  m[c] = sum_p s prob;                 ds[p,c] = dout[c] scale prob[p,c] (1 + tmp (s[p,c] - m[c]))
//   evidence:     s' = s w, m[c] = sum_p s' prob;      de[p,c] = dout[c] scale tmp prob (s' - m[c]);   g = dout[c] scale prob   (= d/d s')
//                 per position, with H = sum_c g s w, u_c = w_c (g_c s_c - H):  ds_c = g_c w_c + k u_c + [c == argmax_c s] tmp sum_c' u_c' s_c'
//                 (k = tmp (max_c s + 1) depends on s through its maximum: torch's max(-1) hands that gradient to the arg-max element)
// One workgroup per image, whole panels in LDS (positions x classes x 12 bytes: the caption branch is 77 x 80).
// t_ld > 0: the outputs are written TRANSPOSED, dneg / devi [C][t_ld] with element (c, b * P + p) - the K-contiguous A operand of the
// GEMM that contracts them with the position features over all B * P rows (d text features = dsim^T . features).
__global__ __launch_bounds__(256) void local_pool_bwd_kernel(const float* __restrict__ sim, const int64_t* __restrict__ mask_tok, const float* __restrict__ dout,
                                                             float* __restrict__ dneg, float* __restrict__ devi, int P, int C, int64_t ld,
                                                             int64_t image_stride, int64_t mask_stride, int evi_off, float tmp, float logit_scale,
                                                             int64_t t_ld) {
    extern __shared__ float sm[];      // k[P] | s_ext[P] | den[P] | bias[P] | amax[P] | cmx[C] | cden[C] | cm[C] | s [P][C] | g [P][C] (| e [P][C])
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* src = sim + (int64_t)blockIdx.x * image_stride;
    float *rk = sm, *rz = sm + P, *rd = sm + 2 * P, *bias = sm + 3 * P;
    int* ramax = (int*)(sm + 4 * P);
    float *cmx = sm + 5 * P, *cden = cmx + C, *cm = cden + C, *s = cm + C, *g = s + P * C, *e = g + P * C;
    const bool evi = evi_off >= 0;
    for (int p = tid; p < P; p += 256) bias[p] = (mask_tok && mask_tok[(int64_t)blockIdx.x * mask_stride + p] == 0) ? -10000.0f : 0.0f;
    __syncthreads();
    for (int i = tid; i < P * C; i += 256) {
        const int p = i / C, c = i - p * C;
        s[i] = src[(int64_t)p * ld + c] + bias[p];
        if (evi) e[i] = src[(int64_t)p * ld + evi_off + c] + bias[p];
    }
    __syncthreads();
    if (evi) {
        for (int p = wave; p < P; p += 4) {
            const float* row = s + p * C;
            float mx = -INFINITY;
            int am = 0x7fffffff;
            for (int c = lane; c < C; c += 64) mx = fmaxf(mx, row[c]);
            mx = wave_max(mx);
            for (int c = lane; c < C; c += 64) if (row[c] == mx) am = c < am ? c : am;     // first maximum
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(am, o); am = t < am ? t : am; }
            const float k = tmp * (mx + 1.0f);
            float mn = INFINITY;
            for (int c = lane; c < C; c += 64) mn = fminf(mn, row[c]);
            mn = -wave_max(-mn);
            const float vext = k >= 0.f ? mx : mn;
            float den = 0.f;
            for (int c = lane; c < C; c += 64) den += expf(k * (row[c] - vext));
            den = wave_sum(den);
            if (lane == 0) { rk[p] = k; rz[p] = vext; rd[p] = den; ramax[p] = am; }
        }
        __syncthreads();
    }
    const float* z = evi ? e : s;
    for (int c = tid; c < C; c += 256) {         // per class: softmax over positions; m[c] = sum_p s' prob
        float mx = tmp * z[c];
        for (int p = 1; p < P; ++p) mx = fmaxf(mx, tmp * z[p * C + c]);
        float den = 0.f, num = 0.f;
        for (int p = 0; p < P; ++p) {
            const float w = expf(tmp * z[p * C + c] - mx);
            float sv = s[p * C + c];
            if (evi) sv *= expf(rk[p] * (sv - rz[p])) / rd[p];
            den += w;
            num = fmaf(w, sv, num);
        }
        cmx[c] = mx; cden[c] = den; cm[c] = num / den;
    }
    __syncthreads();
    const float* dob = dout + (int64_t)blockIdx.x * C;
    const int64_t row0 = (int64_t)blockIdx.x * P;
    // element (p, c) of this image's block: row-major [rows][C], or transposed [C][t_ld]
    auto at = [&](float* base_, int p, int c) -> float& { return t_ld > 0 ? base_[(int64_t)c * t_ld + row0 + p] : base_[(row0 + p) * C + c]; };
    float* dn = dneg;
    float* de = devi;
    for (int j = tid; j < P * C; j += 256) {
        // transposed output: consecutive threads take consecutive positions of one class (contiguous stores)
        const int p = t_ld > 0 ? j % P : j / C, c = t_ld > 0 ? j / P : j - (j / C) * C;
        const int i = p * C + c;
        const float prob = expf(tmp * z[i] - cmx[c]) / cden[c];
        const float gs = dob[c] * logit_scale * prob;
        if (!evi) {
            at(dn, p, c) = gs * (1.0f + tmp * (s[i] - cm[c]));
        } else {
            const float sv = s[i] * (expf(rk[p] * (s[i] - rz[p])) / rd[p]);
            at(de, p, c) = gs * tmp * (sv - cm[c]);
            g[i] = gs;
        }
    }
    if (!evi) return;
    __syncthreads();
    for (int p = wave; p < P; p += 4) {          // through s' = s * softmax_c(k s), k = tmp (max_c s + 1)
        const float* row = s + p * C;
        const float* gr = g + p * C;
        const float k = rk[p], zm = rz[p], dnm = rd[p];
        float H = 0.f;
        for (int c = lane; c < C; c += 64) H = fmaf(gr[c] * row[c], expf(k * (row[c] - zm)) / dnm, H);
        H = wave_sum(H);
        float S = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float w = expf(k * (row[c] - zm)) / dnm;
            S = fmaf(w * (gr[c] * row[c] - H), row[c], S);
        }
        S = wave_sum(S);
        const int am = ramax[p];
        for (int c = lane; c < C; c += 64) {
            const float w = expf(k * (row[c] - zm)) / dnm;
            const float u = w * (gr[c] * row[c] - H);
            at(dn, p, c) = fmaf(k, u, gr[c] * w) + (c == am ? tmp * S : 0.0f);
        }
    }
}

template <typename T>
int launch_tail(const TailArgs& a, hipStream_t s) {
    const int lds = 16 * (a.d * (int)sizeof(T) + 16) + (16 * (a.E + 4) + 16 + a.C) * 4;
    static bool attr_set[LECLIP_MAX_DEVICES] = {};
    leclip_set_max_lds(image_tail_kernel<T>, 160 * 1024, attr_set);
    hipLaunchKernelGGL((image_tail_kernel<T>), dim3((unsigned)((a.B + 15) / 16)), dim3(TAIL_NW * 64), lds, s, a);
    return leclip_check_launch("image_tail_kernel");
}

}  // namespace

extern "C" int leclip_image_tail_fwd(const void* x, const float* gamma, const float* beta, const void* proj_t, const float* txt,
                                     float* feat, float* logits, int64_t B, int64_t row_stride, int dim, int E, int C, float eps,
                                     float scale, leclip_dtype dtype, void* stream) {
    if (!x || !gamma || !beta || !proj_t || B <= 0 || dim <= 0 || E <= 0 || row_stride < dim || (!feat && !logits) || (logits && (!txt || C <= 0))) {
        leclip_set_error("image_tail: null pointer or inconsistent sizes");
        return LECLIP_E_INVALID;
    }
    if (!dtype_ok(dtype)) { leclip_set_error("image_tail: bad dtype"); return LECLIP_E_INVALID; }
    const int kq = dtype == LECLIP_F32 ? 16 : 32;
    if (dim % 64 != 0 || dim > 256 * TAIL_MAXV || dim % kq != 0 || E % 16 != 0 || E > 16 * TAIL_NW * PT_MAX || C > 4096) {
        leclip_set_error("image_tail: dim=%d must be a multiple of 64 and <= %d, E=%d a multiple of 16 (<= %d), C=%d <= 4096", dim, 256 * TAIL_MAXV, E, 16 * TAIL_NW * PT_MAX, C);
        return LECLIP_E_UNSUPPORTED;
    }
    const int esz = dtype_size(dtype);
    if (((uintptr_t)x & 15) || ((row_stride * esz) & 15) || ((uintptr_t)proj_t & 15) || ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15) ||
        (txt && ((uintptr_t)txt & 15)) || (feat && ((uintptr_t)feat & 15))) {
        leclip_set_error("image_tail: x / proj_t / gamma / beta / txt / feat must be 16-byte aligned (row stride a multiple of 16 bytes)");
        return LECLIP_E_INVALID;
    }
    TailArgs a;
    a.x = x; a.gamma = gamma; a.beta = beta; a.proj_t = proj_t; a.txt = logits ? txt : nullptr; a.feat = feat; a.logits = logits;
    a.B = B; a.row_stride = row_stride; a.d = dim; a.E = E; a.C = logits ? C : 0; a.eps = eps; a.scale = scale;
    const size_t lds = 16 * ((size_t)dim * esz + 16) + (16 * ((size_t)E + 4) + 16 + a.C) * 4;
    if (lds > 160 * 1024) { leclip_set_error("image_tail: dim=%d E=%d C=%d needs %zu bytes of LDS", dim, E, C, lds); return LECLIP_E_UNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    if (dtype == LECLIP_F32) return launch_tail<float>(a, s);
    if (dtype == LECLIP_F16) return launch_tail<f16_t>(a, s);
    return launch_tail<bf16_t>(a, s);
}

extern "C" int leclip_l2norm_logits_bwd(const float* img, const float* txt, const float* dlogits, float* dtxt, int64_t B, int C, int D,
                                        float scale, void* stream) {
    if (!img || !txt || !dlogits || !dtxt || B <= 0 || C <= 0 || D <= 0) { leclip_set_error("logits_bwd: null pointer or bad size"); return LECLIP_E_INVALID; }
    if (D > 1024) { leclip_set_error("logits_bwd: D=%d > 1024", D); return LECLIP_E_UNSUPPORTED; }
    hipLaunchKernelGGL(logits_bwd_kernel, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, img, txt, dlogits, dtxt, B, C, D, scale);
    return leclip_check_launch("logits_bwd_kernel");
}

static int move_rows(const void* src, const int64_t* index, void* dst, int64_t n, int dim, int64_t ld_src, int64_t ld_dst, int dtype,
                     int scatter, int64_t dst_rows, void* stream, const char* what) {
    if (!src || !index || !dst || n <= 0 || dim <= 0 || ld_src < dim || ld_dst < dim || !dtype_ok((leclip_dtype)dtype)) {
        leclip_set_error("%s: null pointer or bad size", what);
        return LECLIP_E_INVALID;
    }
    const int esz = dtype_size(dtype);
    const int gran = ((int64_t)dim * esz) % 16 == 0 && (ld_src * esz) % 16 == 0 && (ld_dst * esz) % 16 == 0 && !((uintptr_t)src & 15) && !((uintptr_t)dst & 15) ? 16 : 8;
    if (((int64_t)dim * esz) % gran || (ld_src * esz) % gran || (ld_dst * esz) % gran || ((uintptr_t)src & (gran - 1)) || ((uintptr_t)dst & (gran - 1))) {
        leclip_set_error("%s: rows must be multiples of 8 bytes, 8-byte aligned (16 for the wide path)", what);
        return LECLIP_E_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    if (scatter && hipMemsetAsync(dst, 0, (size_t)dst_rows * ld_dst * esz, s) != hipSuccess) { leclip_set_error("%s: memset failed", what); return LECLIP_E_LAUNCH; }
    hipLaunchKernelGGL(move_rows_kernel, dim3((unsigned)n), dim3(64), 0, s, (const char*)src, index, (char*)dst, n, dim * esz, ld_src * esz,
                       ld_dst * esz, scatter);
    return leclip_check_launch(what);
}

extern "C" int leclip_gather_rows_fwd(const void* src, const int64_t* index, void* dst, int64_t n, int dim, int64_t ld_src, int64_t ld_dst,
                                      leclip_dtype dtype, void* stream) {
    return move_rows(src, index, dst, n, dim, ld_src, ld_dst, (int)dtype, 0, 0, stream, "gather_rows");
}

extern "C" int leclip_scatter_rows_fwd(const void* src, const int64_t* index, void* dst, int64_t n, int64_t dst_rows, int dim, int64_t ld_src,
                                       int64_t ld_dst, leclip_dtype dtype, void* stream) {
    if (dst_rows <= 0) { leclip_set_error("scatter_rows: bad size"); return LECLIP_E_INVALID; }
    return move_rows(src, index, dst, n, dim, ld_src, ld_dst, (int)dtype, 1, dst_rows, stream, "scatter_rows");
}

extern "C" int leclip_l2norm_rows_fwd(float* x, int64_t rows, int dim, int64_t ld, void* stream) {
    if (!x || rows <= 0 || dim <= 0 || ld < dim) { leclip_set_error("l2norm_rows: bad argument"); return LECLIP_E_INVALID; }
    if (dim % 4 || ld % 4 || ((uintptr_t)x & 15)) { leclip_set_error("l2norm_rows: rows must be 16-byte aligned multiples of 4 floats"); return LECLIP_E_UNSUPPORTED; }
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, rows, dim, ld);
    return leclip_check_launch("l2norm_rows_kernel");
}

extern "C" int leclip_transpose_f32_fwd(const float* src, float* dst, int64_t rows, int cols, int64_t ld_src, int64_t ld_dst, void* stream) {
    if (!src || !dst || rows <= 0 || cols <= 0 || ld_src < cols || ld_dst < rows) { leclip_set_error("transpose_f32: bad argument"); return LECLIP_E_INVALID; }
    hipLaunchKernelGGL(transpose_f32_kernel, dim3((unsigned)((rows + 31) / 32), (unsigned)((cols + 31) / 32)), dim3(256), 0, (hipStream_t)stream, src, dst, rows,
                       cols, ld_src, ld_dst);
    return leclip_check_launch("transpose_f32_kernel");
}

extern "C" int leclip_l2norm_rows_bwd(const float* x, const float* dy, float* dx, int64_t rows, int dim, void* stream) {
    if (!x || !dy || !dx || rows <= 0 || dim <= 0) { leclip_set_error("l2norm_rows_bwd: bad argument"); return LECLIP_E_INVALID; }
    hipLaunchKernelGGL(l2norm_rows_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, rows, dim);
    return leclip_check_launch("l2norm_rows_bwd_kernel");
}

static int local_pool_args_ok(const float* sim, int64_t B, int P, int C, int64_t ld, int64_t image_stride, int evidence_offset, const int64_t* mask_tokens,
                              int64_t mask_stride) {
    return sim && B > 0 && P > 0 && C > 0 && ld >= C && image_stride >= (int64_t)(P - 1) * ld + C && !(evidence_offset >= 0 && evidence_offset + C > ld) &&
           !(mask_tokens && mask_stride < P);
}

extern "C" int leclip_local_pool_fwd(const float* sim, float* out, int64_t B, int P, int C, int64_t ld, int64_t image_stride, int evidence_offset,
                                     float spatial_scale, float logit_scale, void* stream) {
    return leclip_local_pool_masked_fwd(sim, nullptr, 0, out, B, P, C, ld, image_stride, evidence_offset, spatial_scale, logit_scale, stream);
}

extern "C" int leclip_local_pool_masked_fwd(const float* sim, const int64_t* mask_tokens, int64_t mask_stride, float* out, int64_t B, int P, int C,
                                            int64_t ld, int64_t image_stride, int evidence_offset, float spatial_scale, float logit_scale, void* stream) {
    if (!out || !local_pool_args_ok(sim, B, P, C, ld, image_stride, evidence_offset, mask_tokens, mask_stride)) {
        leclip_set_error("local_pool: null pointer or inconsistent sizes");
        return LECLIP_E_INVALID;
    }
    // classes per pass: as many as fit 144 KiB beside the four per-position vectors (>= 1 for P <= 4000)
    const int panels = evidence_offset >= 0 ? 2 : 1;
    const int64_t budget = 144 * 1024 - 16 * (int64_t)P;
    int ct = budget > 0 ? (int)(budget / ((int64_t)P * 4 * panels)) : 0;
    if (ct < 1) { leclip_set_error("local_pool: P=%d positions do not fit LDS", P); return LECLIP_E_UNSUPPORTED; }
    if (ct > C) ct = C;
    const size_t lds = (size_t)P * 16 + (size_t)P * ct * 4 * panels;
    static bool attr_set[LECLIP_MAX_DEVICES] = {};
    leclip_set_max_lds(local_pool_kernel, 160 * 1024, attr_set);
    hipLaunchKernelGGL(local_pool_kernel, dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream, sim, mask_tokens, out, P, C, ct, ld, image_stride, mask_stride,
                       evidence_offset, spatial_scale, logit_scale);
    return leclip_check_launch("local_pool_kernel");
}

extern "C" int leclip_local_pool_bwd(const float* sim, const int64_t* mask_tokens, int64_t mask_stride, const float* dout, float* dneg, float* devi,
                                     int64_t B, int P, int C, int64_t ld, int64_t image_stride, int evidence_offset, float spatial_scale, float logit_scale,
                                     int64_t transposed_ld, void* stream) {
    if (!dout || !dneg || (evidence_offset >= 0) != (devi != nullptr) || (transposed_ld != 0 && transposed_ld < B * P) ||
        !local_pool_args_ok(sim, B, P, C, ld, image_stride, evidence_offset, mask_tokens, mask_stride)) {
        leclip_set_error("local_pool_bwd: null pointer or inconsistent sizes");
        return LECLIP_E_INVALID;
    }
    const size_t lds = ((size_t)5 * P + 3 * C + (size_t)P * C * (evidence_offset >= 0 ? 3 : 2)) * 4;
    if (lds > 160 * 1024) { leclip_set_error("local_pool_bwd: P=%d x C=%d panels do not fit LDS (the training branch pools 77 token positions)", P, C); return LECLIP_E_UNSUPPORTED; }
    static bool attr_set[LECLIP_MAX_DEVICES] = {};
    leclip_set_max_lds(local_pool_bwd_kernel, 160 * 1024, attr_set);
    hipLaunchKernelGGL(local_pool_bwd_kernel, dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream, sim, mask_tokens, dout, dneg, devi, P, C, ld, image_stride,
                       mask_stride, evidence_offset, spatial_scale, logit_scale, transposed_ld);
    return leclip_check_launch("local_pool_bwd_kernel");
}
