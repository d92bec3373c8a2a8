// Exact-fp32 validation GEMM: Y = act(A . W^T + bias) + residual with v_mfma_f32_32x32x2_f32
// (f32 in / f32 accumulate, bit-for-bit a k-ordered fmaf chain - no reduced-precision path exists on gfx950).
// This is the "fp32 parity mode" of the scoring path (<= 1e-3 against the CPU reference); it is not the
// throughput kernel.  64x64 tile per 256-thread workgroup, BK = 32, operands transposed into k-major LDS so
// every ds_read_b32 is conflict-free; next tile prefetched to registers while the current one is multiplied.
#include "leclip_common.h"

namespace {

constexpr int FM = 64, FN = 64, FK = 32, LD = 65;

struct GemmF32Args {
    const float* A;
    const float* W;
    int64_t M;
    int N, K;
    int64_t lda, ldw;
    EpiParams epi;
    int tiles_n;
    // split-K (few output tiles, long K: the prompt-feature gradients of the local head contract over every caption token): workgroup
    // (tile, split) accumulates K-range [split * kchunk, ...) and writes its raw fp32 tile to partial[split][M][N]; gemm_f32_reduce adds the
    // splits in index order (deterministic) and applies the epilogue.  partial == nullptr: one workgroup per tile, the whole K.
    int tiles, splits, kchunk;
    float* partial;
};

__global__ __launch_bounds__(256) void gemm_f32_64x64x32(GemmF32Args g) {
    __shared__ float As[FK * LD];
    __shared__ float Bs[FK * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int split = blockIdx.x / g.tiles, tile = blockIdx.x - split * g.tiles;
    const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
    const int64_t m0 = (int64_t)tm * FM;
    const int n0 = tn * FN;
    const int k0 = split * g.kchunk;
    const int klen = g.K - k0 < g.kchunk ? g.K - k0 : g.kchunk;

    // loader mapping: thread -> (row = tid>>2, 8 consecutive k at (tid&3)*8)
    const int lrow = tid >> 2, lk = (tid & 3) * 8;
    int64_t arow = m0 + lrow;
    arow = arow < g.M ? arow : g.M - 1;
    const float* ap = g.A + arow * g.lda + k0 + lk;
    const float* wp = g.W + (int64_t)(n0 + lrow) * g.ldw + k0 + lk;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    f32x4 ra0 = *(const f32x4*)(ap), ra1 = *(const f32x4*)(ap + 4);
    f32x4 rb0 = *(const f32x4*)(wp), rb1 = *(const f32x4*)(wp + 4);
    const int nk = klen / FK;
    for (int t = 0; t < nk; ++t) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            As[(lk + i) * LD + lrow] = ra0[i];
            As[(lk + 4 + i) * LD + lrow] = ra1[i];
            Bs[(lk + i) * LD + lrow] = rb0[i];
            Bs[(lk + 4 + i) * LD + lrow] = rb1[i];
        }
        __syncthreads();
        if (t + 1 < nk) {
            ra0 = *(const f32x4*)(ap + (t + 1) * FK);
            ra1 = *(const f32x4*)(ap + (t + 1) * FK + 4);
            rb0 = *(const f32x4*)(wp + (t + 1) * FK);
            rb1 = *(const f32x4*)(wp + (t + 1) * FK + 4);
        }
        // v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]
        const float* a = As + (lane >> 5) * LD + wr * 32 + (lane & 31);
        const float* b = Bs + (lane >> 5) * LD + wc * 32 + (lane & 31);
#pragma unroll
        for (int k = 0; k < FK; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k * LD], b[k * LD], acc, 0, 0, 0);
    }

    const EpiParams& e = g.epi;
    const int n = n0 + wc * 32 + (lane & 31);
    if (g.partial) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m < g.M) g.partial[((int64_t)split * g.M + m) * g.N + n] = acc[r];
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < g.M) {
            const float v = epi_apply<true>(e, m, n, acc[r]);
            store_elem(e.out, e.out_dt, epi_out_row(e, m) * e.ldy + n, v);
        }
    }
}

__global__ __launch_bounds__(256) void gemm_f32_reduce(GemmF32Args g) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= g.M * g.N) return;
    const int64_t m = i / g.N;
    const int n = (int)(i - m * g.N);
    float v = 0.f;
    for (int sp = 0; sp < g.splits; ++sp) v += g.partial[(int64_t)sp * g.M * g.N + i];
    store_elem(g.epi.out, g.epi.out_dt, epi_out_row(g.epi, m) * g.epi.ldy + n, epi_apply<true>(g.epi, m, n, v));
}

}  // namespace

int leclip_gemm_f32_launch(const void* A, const void* W, int64_t M, int N, int K, int64_t lda, int64_t ldw,
                           const EpiParams& epi, hipStream_t s) {
    if (N % FN != 0 || K % FK != 0) {
        leclip_set_error("gemm(f32): N=%d must be a multiple of %d and K=%d a multiple of %d", N, FN, K, FK);
        return LECLIP_E_UNSUPPORTED;
    }
    if ((lda % 4) || (ldw % 4) || ((uintptr_t)A & 15) || ((uintptr_t)W & 15)) {
        leclip_set_error("gemm(f32): A/W must be 16-byte aligned with leading dimensions that are multiples of 4");
        return LECLIP_E_INVALID;
    }
    GemmF32Args a;
    a.A = (const float*)A; a.W = (const float*)W; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.epi = epi;
    a.tiles_n = N / FN;
    const int64_t tiles = ((M + FM - 1) / FM) * a.tiles_n;
    if (tiles > 0x7fffffff) { leclip_set_error("gemm(f32): too many tiles"); return LECLIP_E_UNSUPPORTED; }
    a.tiles = (int)tiles; a.splits = 1; a.kchunk = K; a.partial = nullptr;
    // few tiles and a long contraction: spread K over ~4 workgroups per CU (each split at least 8 K-steps), partial tiles through a
    // stream-ordered scratch buffer
    if (tiles < 128 && K >= 8192) {   // (never a forward shape: those keep the single k-ordered chain, bit-identical across batch sizes)
        int want = (int)((4 * leclip_cu_count() + tiles - 1) / tiles);
        const int max_splits = K / (8 * FK);
        if (want > max_splits) want = max_splits;
        if (want > 1) {
            const int kchunk = ((K + want - 1) / want + FK - 1) / FK * FK;
            const int splits = (K + kchunk - 1) / kchunk;
            float* ws = nullptr;
            {   // keep freed scratch in the stream-ordered pool between training steps (once per device): by default it returns to the driver at each sync
                static bool pool_set[LECLIP_MAX_DEVICES] = {};
                const int dev = leclip_device_ordinal();
                if (!pool_set[dev]) {
                    hipMemPool_t pool;
                    if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
                        uint64_t keep = 1ull << 30;
                        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
                    }
                    (void)hipGetLastError();
                    pool_set[dev] = true;
                }
            }
            if (splits > 1 && hipMallocAsync((void**)&ws, (size_t)splits * M * N * sizeof(float), s) == hipSuccess) {
                a.splits = splits; a.kchunk = kchunk; a.partial = ws;
                hipLaunchKernelGGL(gemm_f32_64x64x32, dim3((unsigned)(tiles * splits)), dim3(256), 0, s, a);
                hipLaunchKernelGGL(gemm_f32_reduce, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, s, a);
                (void)hipFreeAsync(ws, s);
                return leclip_check_launch("gemm_f32_64x64x32 (split-K)");
            }
            (void)hipGetLastError();   // no scratch: the unsplit kernel below
        }
    }
    hipLaunchKernelGGL(gemm_f32_64x64x32, dim3((unsigned)tiles), dim3(256), 0, s, a);
    return leclip_check_launch("gemm_f32_64x64x32");
}
