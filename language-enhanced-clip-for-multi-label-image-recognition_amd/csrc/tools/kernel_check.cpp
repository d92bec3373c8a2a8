// Standalone (no Python) correctness + timing harness for libleclip_hip.so, used during kernel development:
//   kernel_check            -> correctness of GEMM / attention against a CPU double-precision reference
//   kernel_check bench      -> timings of the ViT-B/16 B=256 shapes
// Build: hipcc -O2 --offload-arch=gfx950 kernel_check.cpp -L../lib -lleclip_hip -o ../../lib/leclip_kernel_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include <algorithm>
#include "../../../include/leclip_hip.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

struct Buf {
    void* d = nullptr; size_t bytes = 0;
    explicit Buf(size_t b) : bytes(b) { HIPCHK(hipMalloc(&d, b)); }
    ~Buf() { if (d) (void)hipFree(d); }
    void up(const void* h) { HIPCHK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); }
    void down(void* h) { HIPCHK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); }
};

static std::mt19937 rng(12345);
static std::vector<float> randn(size_t n, float s = 1.f) { std::normal_distribution<float> d(0.f, s); std::vector<float> v(n); for (auto& x : v) x = d(rng); return v; }

// quantise to dtype and keep the rounded value in fp32 for the reference
static std::vector<uint8_t> pack(std::vector<float>& v, int dt) {
    std::vector<uint8_t> out(v.size() * (dt == LECLIP_F32 ? 4 : 2));
    for (size_t i = 0; i < v.size(); ++i) {
        if (dt == LECLIP_F32) memcpy(&out[4 * i], &v[i], 4);
        else if (dt == LECLIP_BF16) { uint16_t h = f2bf(v[i]); v[i] = bf2f(h); memcpy(&out[2 * i], &h, 2); }
        else { uint16_t h = f2h(v[i]); v[i] = h2f(h); memcpy(&out[2 * i], &h, 2); }
    }
    return out;
}
static float unpack1(const uint8_t* p, int dt, size_t i) {
    if (dt == LECLIP_F32) { float f; memcpy(&f, p + 4 * i, 4); return f; }
    uint16_t h; memcpy(&h, p + 2 * i, 2);
    return dt == LECLIP_BF16 ? bf2f(h) : h2f(h);
}
static const char* dtn(int dt) { return dt == LECLIP_F32 ? "f32" : dt == LECLIP_F16 ? "f16" : "bf16"; }

static int g_fail = 0;

static void check_gemm(int64_t M, int N, int K, int ab, int resdt, int outdt, int act, bool bias, bool res) {
    auto A = randn(M * K), W = randn((size_t)N * K, 1.f / std::sqrt((float)K)), B = randn(N), R = randn(M * N);
    auto Ap = pack(A, ab), Wp = pack(W, ab), Rp = pack(R, resdt);
    Buf dA(Ap.size()), dW(Wp.size()), dB(N * 4), dR(Rp.size()), dY((size_t)M * N * (outdt == LECLIP_F32 ? 4 : 2));
    dA.up(Ap.data()); dW.up(Wp.data()); dB.up(B.data()); dR.up(Rp.data());
    HIPCHK(hipMemset(dY.d, 0xFF, dY.bytes));
    int rc = leclip_gemm_bias_act_res_fwd(dA.d, dW.d, bias ? (float*)dB.d : nullptr, res ? dR.d : nullptr, dY.d, M, N, K, K, K, N, N,
                                          (leclip_act)act, (leclip_dtype)ab, (leclip_dtype)resdt, (leclip_dtype)outdt, nullptr);
    HIPCHK(hipDeviceSynchronize());
    if (rc) { printf("FAIL gemm rc=%d %s\n", rc, leclip_last_error()); g_fail++; return; }
    std::vector<uint8_t> Y(dY.bytes); dY.down(Y.data());
    double maxerr = 0, maxref = 0;
    for (int64_t m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
        double acc = 0;
        for (int k = 0; k < K; ++k) acc += (double)A[m * K + k] * W[(size_t)n * K + k];
        if (bias) acc += B[n];
        if (act) acc = acc / (1.0 + std::exp(-1.702 * acc));
        if (res) acc += R[m * N + n];
        double got = unpack1(Y.data(), outdt, m * N + n);
        maxerr = std::max(maxerr, std::fabs(got - acc)); maxref = std::max(maxref, std::fabs(acc));
    }
    double tol = outdt == LECLIP_F32 ? (ab == LECLIP_F32 ? 2e-5 : 2e-3) * maxref : (outdt == LECLIP_BF16 ? 8e-3 : 2e-3) * maxref;
    bool ok = maxerr <= tol && maxerr == maxerr;
    printf("%s gemm M=%lld N=%d K=%d ab=%s res=%s out=%s act=%d bias=%d: maxerr %.3e (ref max %.2f, tol %.2e)\n", ok ? "ok  " : "FAIL",
           (long long)M, N, K, dtn(ab), res ? dtn(resdt) : "-", dtn(outdt), act, bias, maxerr, maxref, tol);
    if (!ok) g_fail++;
}

// spike > 0: some key rows are set to multiples of query rows, so that a query's running maximum jumps in a LATER key chunk - by far
// more than the streaming kernel's lazy-reference threshold (factor 3: +35 in log2 units) and by less than it (factor 0.6) - the
// data-dependent rescale branch and the stale-reference path both run (cdna_hip_programming.md rule 26).
static void check_attn(int B, int T, int heads, int dt, int causal, int spike = 0) {
    const int d = heads * 64;
    auto Q = randn((size_t)B * T * 3 * d);
    if (spike) {
        for (int b = 0; b < B; ++b) for (int h = 0; h < heads; ++h) {
            struct P { int q, k; float f; } pairs[] = {{5, 140, 3.0f}, {40, 300, 3.0f}, {41, 520, 0.6f}, {T - 3, T - 2, 3.0f}, {T - 1, 130, 0.6f},
                                                       {200, 131, 3.0f}, {200, 400, 4.0f}, {333, T - 1, 3.0f}, {64, 64, 3.0f}, {65, 257, 2.0f}};
            for (auto& pr : pairs) {
                if (pr.q >= T || pr.k >= T || pr.q < 0 || pr.k < 0) continue;
                for (int e = 0; e < 64; ++e) Q[((size_t)b * T + pr.k) * 3 * d + d + h * 64 + e] = pr.f * Q[((size_t)b * T + pr.q) * 3 * d + h * 64 + e];
            }
        }
    }
    auto Qp = pack(Q, dt);
    Buf dQ(Qp.size()), dO((size_t)B * T * d * (dt == LECLIP_F32 ? 4 : 2));
    dQ.up(Qp.data());
    HIPCHK(hipMemset(dO.d, 0xFF, dO.bytes));
    int rc = leclip_attention_fwd(dQ.d, dO.d, B, T, heads, 64, 3 * d, d, causal ? LECLIP_MASK_CAUSAL : LECLIP_MASK_NONE, 0.125f, (leclip_dtype)dt, nullptr);
    HIPCHK(hipDeviceSynchronize());
    if (rc) { printf("FAIL attn rc=%d %s\n", rc, leclip_last_error()); g_fail++; return; }
    std::vector<uint8_t> O(dO.bytes); dO.down(O.data());
    double maxerr = 0;
    std::vector<double> s(T);
    for (int b = 0; b < B; ++b) for (int h = 0; h < heads; ++h) for (int q = 0; q < T; ++q) {
        const float* qr = &Q[((size_t)b * T + q) * 3 * d + h * 64];
        double mx = -1e300;
        int kl = causal ? q : T - 1;
        for (int k = 0; k <= kl; ++k) {
            const float* kr = &Q[((size_t)b * T + k) * 3 * d + d + h * 64];
            double acc = 0; for (int e = 0; e < 64; ++e) acc += (double)qr[e] * kr[e];
            s[k] = acc * 0.125; mx = std::max(mx, s[k]);
        }
        double sum = 0; for (int k = 0; k <= kl; ++k) { s[k] = std::exp(s[k] - mx); sum += s[k]; }
        for (int e = 0; e < 64; ++e) {
            double acc = 0;
            for (int k = 0; k <= kl; ++k) acc += s[k] * Q[((size_t)b * T + k) * 3 * d + 2 * d + h * 64 + e];
            acc /= sum;
            double got = unpack1(O.data(), dt, ((size_t)b * T + q) * d + h * 64 + e);
            maxerr = std::max(maxerr, std::fabs(got - acc));
        }
    }
    double tol = dt == LECLIP_F32 ? 2e-5 : (dt == LECLIP_BF16 ? 2.5e-2 : 4e-3);
    bool ok = maxerr <= tol && maxerr == maxerr;
    printf("%s attn B=%d T=%d heads=%d %s causal=%d spike=%d: maxerr %.3e (tol %.1e)\n", ok ? "ok  " : "FAIL", B, T, heads, dtn(dt), causal, spike, maxerr, tol);
    if (!ok) g_fail++;
}

static void check_gemm_identity() {
    // A = I (128x128, K=128), asymmetric W: catches transposed / permuted fragment maps exactly
    const int M = 128, N = 128, K = 128;
    std::vector<float> A(M * K, 0.f), W((size_t)N * K);
    for (int i = 0; i < M; ++i) A[i * K + i] = 1.f;
    for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) W[(size_t)n * K + k] = (float)((n * 3 + k * 7) % 64) - 20.f;
    auto Ap = pack(A, LECLIP_BF16), Wp = pack(W, LECLIP_BF16);
    Buf dA(Ap.size()), dW(Wp.size()), dY(M * N * 4);
    dA.up(Ap.data()); dW.up(Wp.data());
    int rc = leclip_gemm_bias_act_res_fwd(dA.d, dW.d, nullptr, nullptr, dY.d, M, N, K, K, K, N, N, LECLIP_ACT_NONE, LECLIP_BF16, LECLIP_F32, LECLIP_F32, nullptr);
    HIPCHK(hipDeviceSynchronize());
    std::vector<float> Y(M * N); dY.down(Y.data());
    int bad = 0;
    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) if (Y[m * N + n] != W[(size_t)n * K + m]) bad++;
    printf("%s gemm identity/asymmetric exact check rc=%d mismatches=%d\n", (bad == 0 && rc == 0) ? "ok  " : "FAIL", rc, bad);
    if (bad || rc) g_fail++;
}

static double time_ms(int iters, const std::function<void()>& fn) {
    hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) fn();
    HIPCHK(hipEventRecord(a, nullptr));
    for (int i = 0; i < iters; ++i) fn();
    HIPCHK(hipEventRecord(b, nullptr)); HIPCHK(hipEventSynchronize(b));
    float ms; HIPCHK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

static void bench() {
    const int64_t M = 256 * 197;
    struct S { int N, K, act; bool res; const char* name; } shapes[] = {
        {2304, 768, 0, false, "qkv"}, {768, 768, 0, true, "out_proj"}, {3072, 768, 1, false, "c_fc+gelu"}, {768, 3072, 0, true, "c_proj"}};
    const char* only = getenv("LECLIP_BENCH_DT");   // "f16" / "bf16": GEMM shapes of that dtype only (A/B scripts)
    const bool quick = getenv("LECLIP_BENCH_QUICK") != nullptr || only != nullptr;
    for (int dt : {LECLIP_BF16, LECLIP_F16}) for (auto& s : shapes) {
        if (only ? strcmp(only, dtn(dt)) != 0 : (quick && dt != LECLIP_BF16)) continue;
        auto A = randn(M * s.K), W = randn((size_t)s.N * s.K, 0.03f), Bv = randn(s.N), R = randn(M * s.N);
        auto Ap = pack(A, dt), Wp = pack(W, dt), Rp = pack(R, dt);
        Buf dA(Ap.size()), dW(Wp.size()), dB(s.N * 4), dR(Rp.size()), dY((size_t)M * s.N * 2);
        dA.up(Ap.data()); dW.up(Wp.data()); dB.up(Bv.data()); dR.up(Rp.data());
        double ms = time_ms(20, [&] {
            leclip_gemm_bias_act_res_fwd(dA.d, dW.d, (float*)dB.d, s.res ? dR.d : nullptr, dY.d, M, s.N, s.K, s.K, s.K, s.N, s.N,
                                         (leclip_act)s.act, (leclip_dtype)dt, (leclip_dtype)dt, (leclip_dtype)dt, nullptr);
        });
        printf("bench gemm %-10s %s M=%lld N=%d K=%d: %.3f ms  %.1f TFLOP/s\n", s.name, dtn(dt), (long long)M, s.N, s.K, ms, 2.0 * M * s.N * s.K / ms * 1e-9);
    }
    if (!quick) {
        const int B = 256, T = 197, heads = 12, d = 768;
        auto Q = randn((size_t)B * T * 3 * d);
        auto Qp = pack(Q, LECLIP_BF16);
        Buf dQ(Qp.size()), dO((size_t)B * T * d * 2);
        dQ.up(Qp.data());
        double ms = time_ms(20, [&] { leclip_attention_fwd(dQ.d, dO.d, B, T, heads, 64, 3 * d, d, LECLIP_MASK_NONE, 0.125f, LECLIP_BF16, nullptr); });
        printf("bench attn B=256 T=197 h=12 bf16: %.3f ms  %.1f TFLOP/s (4*T*T*64 per head)\n", ms, 4.0 * T * T * 64 * B * heads / ms * 1e-9);
        auto X = randn((size_t)M * d), G = randn(d), Bt = randn(d);
        auto Xp = pack(X, LECLIP_BF16);
        Buf dX(Xp.size()), dG(d * 4), dBt(d * 4), dYn((size_t)M * d * 2);
        dX.up(Xp.data()); dG.up(G.data()); dBt.up(Bt.data());
        ms = time_ms(20, [&] { leclip_layernorm_fwd(dX.d, (float*)dG.d, (float*)dBt.d, dYn.d, M, d, d, d, 1e-5f, LECLIP_BF16, LECLIP_BF16, nullptr); });
        printf("bench layernorm rows=%lld dim=768 bf16->bf16: %.3f ms  %.1f GB/s\n", (long long)M, ms, 2.0 * M * d * 2 / ms * 1e-6);
    }
}


// Attention alone: the many-heads kernels at the grids the model launches them with (kernel_check's default cases stay below
// the 1024-head threshold of the pipelined ViT-B kernel), checked against the double-precision reference, then timed.
static int attn() {
    const bool bench_only = getenv("LECLIP_ATTN_BENCH_ONLY") != nullptr;   // ablation builds compute garbage: timings only
    const char* only = getenv("LECLIP_ATTN_SHAPES");                      // "B" / "L": ViT-B or ViT-L shapes only
    if (!bench_only) for (int dt : {LECLIP_F16, LECLIP_BF16}) {
        check_attn(86, 197, 12, dt, 0);    // 1032 heads: attn_heads_kernel
        check_attn(90, 200, 12, dt, 0);
        check_attn(2, 577, 16, dt, 0);     // attn_stream_kernel
        check_attn(1, 640, 3, dt, 0);
        check_attn(1, 300, 2, dt, 1);
        check_attn(2, 577, 3, dt, 0, 1);   // maxima that jump in later chunks (lazy reference maximum: rescale branch + stale reference)
        check_attn(1, 600, 2, dt, 1, 1);
        check_attn(1, 257, 2, dt, 0, 1);
    }
    printf(g_fail ? "FAILED %d checks\n" : "ALL OK\n", g_fail);
    if (g_fail) return 1;
    struct S { int B, T, heads; const char* name; } shapes[] = {{256, 197, 12, "ViT-B/16 B=256"}, {128, 197, 12, "ViT-B/16 B=128"}, {128, 577, 16, "ViT-L/14@336 B=128"}, {64, 577, 16, "ViT-L/14@336 B=64"}};
    for (int round = 0; round < 2; ++round)
        for (auto& s : shapes) for (int dt : {LECLIP_F16, LECLIP_BF16}) {
            if (only && ((only[0] == 'L') != (s.T == 577))) continue;
            const int d = s.heads * 64;
            auto Q = randn((size_t)s.B * s.T * 3 * d, 0.8f);
            auto Qp = pack(Q, dt);
            Buf dQ(Qp.size()), dO((size_t)s.B * s.T * d * 2);
            dQ.up(Qp.data());
            double ms = time_ms(50, [&] { leclip_attention_fwd(dQ.d, dO.d, s.B, s.T, s.heads, 64, 3 * d, d, LECLIP_MASK_NONE, 0.125f, (leclip_dtype)dt, nullptr); });
            printf("bench attn %-20s %s: %.1f us  %.1f TFLOP/s  %.2f TB/s\n", s.name, dtn(dt), ms * 1e3, 4.0 * s.T * s.T * 64 * s.B * s.heads / ms * 1e-9,
                   (double)s.B * s.T * d * 2 * 4 / ms * 1e-9);
        }
    return 0;
}

// Diagnostic (variant builds with -DLECLIP_ATTN_STAMPS): where a workgroup of the streaming attention kernel spends its time.
extern "C" int leclip_attn_stamps_read(unsigned long long* host, size_t n) __attribute__((weak));
static int attnstamps() {
    if (!leclip_attn_stamps_read) { printf("library built without attention stamps\n"); return 1; }
    const int B = 128, T = 577, heads = 16, d = heads * 64, dt = LECLIP_F16;
    auto Q = randn((size_t)B * T * 3 * d, 0.8f);
    auto Qp = pack(Q, dt);
    Buf dQ(Qp.size()), dO((size_t)B * T * d * 2);
    dQ.up(Qp.data());
    for (int i = 0; i < 20; ++i) leclip_attention_fwd(dQ.d, dO.d, B, T, heads, 64, 3 * d, d, LECLIP_MASK_NONE, 0.125f, (leclip_dtype)dt, nullptr);
    HIPCHK(hipDeviceSynchronize());
    std::vector<unsigned long long> st(2048 * 2 * 16);
    leclip_attn_stamps_read(st.data(), st.size());
    const char* names[] = {"staging issue+wait", "barrier", "block 1 chunks", "block 1 stores", "block 2 chunks", "block 2 stores", "block 3 chunks", "block 3 stores"};
    for (int w = 0; w < 2; ++w) {
        std::vector<double> seg[8], total, clk;
        for (int g = 0; g < 2048; ++g) {
            const unsigned long long* e = &st[(g * 2 + w) * 16];
            const int nb = w == 0 ? 3 : 2;
            unsigned long long prev = e[0];
            const unsigned long long pts[8] = {e[1], e[2], e[3], e[4], e[5], e[6], e[7], e[8]};
            for (int k = 0; k < 2 + 2 * nb; ++k) { seg[k].push_back((double)(pts[k] - prev)); prev = pts[k]; }
            total.push_back((double)(e[12] - e[0]));
            if (e[15] > e[14]) clk.push_back((double)(e[12] - e[0]) / (double)(e[15] - e[14]) * 0.1);   // GHz: shader cycles per 10 ns tick
        }
        auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("wave %d: workgroup lifetime median %.0f cycles, in-kernel clock median %.2f GHz\n", w * 4, med(total), med(clk));
        for (int k = 0; k < 8; ++k) if (!seg[k].empty()) printf("    %-20s median %8.0f cycles\n", names[k], med(seg[k]));
    }
    return 0;
}

// Diagnostic: in-kernel timeline of the 256x256 GEMM (needs the library built with -DLECLIP_GEMM_STAMPS: make stamps).
extern "C" void leclip_gemm256_set_stamps(unsigned long long* device_buf) __attribute__((weak));
static void stamps() {
    if (!leclip_gemm256_set_stamps) { printf("library built without the stamp hook\n"); return; }
    const int64_t M = 256 * 197;
    struct S { int N, K, act; bool res; const char* name; } shapes[] = {
        {2304, 768, 0, false, "qkv"}, {768, 768, 0, true, "out_proj"}, {3072, 768, 1, false, "c_fc+gelu"}, {768, 3072, 0, true, "c_proj"}};
    const int dt = LECLIP_BF16;
    const size_t n_st = 256 * 16 * 8;
    const size_t n_fine = 256 * 2 * 240;                // per workgroup, waves 0 / 4: 12 K-tiles x 4 phases x 5 stamps (32-bit), second tile
    Buf dS(n_st * 8 + n_fine * 4);
    for (auto& s : shapes) {
        auto A = randn(M * s.K), W = randn((size_t)s.N * s.K, 0.03f), Bv = randn(s.N), R = randn(M * s.N);
        auto Ap = pack(A, dt), Wp = pack(W, dt), Rp = pack(R, dt);
        Buf dA(Ap.size()), dW(Wp.size()), dB(s.N * 4), dR(Rp.size()), dY((size_t)M * s.N * 2);
        dA.up(Ap.data()); dW.up(Wp.data()); dB.up(Bv.data()); dR.up(Rp.data());
        auto run = [&] {
            leclip_gemm_bias_act_res_fwd(dA.d, dW.d, (float*)dB.d, s.res ? dR.d : nullptr, dY.d, M, s.N, s.K, s.K, s.K, s.N, s.N,
                                         (leclip_act)s.act, (leclip_dtype)dt, (leclip_dtype)dt, (leclip_dtype)dt, nullptr);
        };
        leclip_gemm256_set_stamps(nullptr);
        for (int i = 0; i < 200; ++i) run();   // settle the clock
        HIPCHK(hipMemset(dS.d, 0, n_st * 8 + n_fine * 4));
        leclip_gemm256_set_stamps((unsigned long long*)dS.d);
        run();
        HIPCHK(hipDeviceSynchronize());
        leclip_gemm256_set_stamps(nullptr);
        std::vector<unsigned long long> st(n_st);
        HIPCHK(hipMemcpy(st.data(), dS.d, n_st * 8, hipMemcpyDeviceToHost));
        // per tile index: mean over workgroups of the five segment lengths (cycles)
        printf("stamps %-10s N=%d K=%d   (cycles, mean over workgroups; wave 0)\n", s.name, s.N, s.K);
        printf("  tile   n_wg   kloop  close+operands  prologue-issue  passes  wait-at-next-top   tile-total\n");
        for (int it = 0; it < 16; ++it) {
            double seg[6] = {0, 0, 0, 0, 0, 0};
            int n = 0, n_next = 0;
            for (int wg = 0; wg < 256; ++wg) {
                const unsigned long long* t = &st[((size_t)wg * 16 + it) * 8];
                if (!t[0] || !t[4]) continue;
                ++n;
                seg[0] += (double)(t[1] - t[0]); seg[1] += (double)(t[2] - t[1]); seg[2] += (double)(t[3] - t[2]); seg[3] += (double)(t[4] - t[3]);
                if (it + 1 < 16) {
                    const unsigned long long* u = &st[((size_t)wg * 16 + it + 1) * 8];
                    if (u[0]) { seg[4] += (double)(u[0] - t[4]); seg[5] += (double)(u[0] - t[0]); ++n_next; }
                }
            }
            if (!n) break;
            printf("  %4d  %5d  %6.0f  %14.0f  %14.0f  %6.0f  %16.0f  %11.0f\n", it, n, seg[0] / n, seg[1] / n, seg[2] / n, seg[3] / n,
                   n_next ? seg[4] / n_next : 0.0, n_next ? seg[5] / n_next : 0.0);
        }
        {   // per-phase timeline of every K-tile of the SECOND tile, mean over workgroups, waves 0 and 4 (partners on one SIMD):
            // segments: mem-cluster-issued -> [wait at barrier 1] -> [lgkmcnt] -> [16 MFMAs issue] -> [wait at barrier 2] -> next phase's memory cluster
            std::vector<unsigned> fs(n_fine);
            HIPCHK(hipMemcpy(fs.data(), (char*)dS.d + n_st * 8, n_fine * 4, hipMemcpyDeviceToHost));
            const int nkt = s.K / 64 < 12 ? s.K / 64 : 12;
            for (int wv = 0; wv < 2; ++wv) {
                std::vector<double> seg(48 * 6, 0.0);
                int n = 0;
                for (int wg = 0; wg < 256; ++wg) {
                    const unsigned* f = &fs[(size_t)(wg * 2 + wv) * 240];
                    if (!f[0] || !f[nkt * 20 - 1]) continue;
                    ++n;
                    for (int ph = 0; ph < nkt * 4; ++ph) {
                        const unsigned* q = f + ph * 5;
                        for (int k = 0; k < 4; ++k) seg[ph * 6 + k] += (double)(unsigned)(q[k + 1] - q[k]);
                        if (ph + 1 < nkt * 4) seg[ph * 6 + 4] += (double)(unsigned)(q[5] - q[4]);
                        if (ph + 1 < nkt * 4) seg[ph * 6 + 5] += (double)(unsigned)(q[5] - q[0]);
                    }
                }
                if (!n) continue;
                printf("  fine wave %d, second tile (n=%d): K-tile  [per phase: bar1 lgkm mfma16 bar2 mem(next)] x4   K-tile total\n", wv * 4, n);
                for (int kt = 0; kt < nkt; ++kt) {
                    printf("    kt%-2d", kt);
                    double tot = 0;
                    for (int ph = 0; ph < 4; ++ph) {
                        const double* q = &seg[(kt * 4 + ph) * 6];
                        printf("  | %4.0f %4.0f %4.0f %4.0f %4.0f", q[0] / n, q[1] / n, q[2] / n, q[3] / n, q[4] / n);
                        tot += q[5] / n;
                    }
                    printf("  | %6.0f\n", tot);
                }
            }
        }
        // skew between workgroups at the first and the last stamp
        unsigned long long lo0 = ~0ull, hi0 = 0, lo4 = ~0ull, hi4 = 0;
        for (int wg = 0; wg < 256; ++wg) {
            const unsigned long long* t = &st[(size_t)wg * 16 * 8];
            if (!t[0]) continue;
            lo0 = t[0] < lo0 ? t[0] : lo0; hi0 = t[0] > hi0 ? t[0] : hi0;
            unsigned long long last = 0;
            for (int it = 0; it < 16; ++it) if (t[it * 8 + 4]) last = t[it * 8 + 4];
            lo4 = last < lo4 ? last : lo4; hi4 = last > hi4 ? last : hi4;
        }
        printf("  first K-loop start spread %llu cycles, last store issue spread %llu cycles, kernel span %llu cycles\n", hi0 - lo0, hi4 - lo4, hi4 - lo0);
    }
}

// Round 5: the 384 x 256 GEMM family against the 256 x 256 one on the same call (leclip_set_gemm_family), bit for bit, every flavour the new
// kernel has - residual (+ LayerNorm partials), fused LayerNorm (+ QuickGELU), plain bias (+ QuickGELU) - ragged M, K-step counts 3 .. 96 with
// every remainder mod 3 (the ring rotates), then timings of both families on the ViT-B/16 shapes.
static int g384() {
    struct C { int64_t M; int N, K; } cases[] = {{50432, 768, 768}, {50432 - 100, 768, 3072}, {30000, 2304, 768}, {20001, 3072, 768}, {1000, 256, 192},
                                                {777, 512, 128}, {2500, 256, 320}, {4000, 1024, 1024}, {385, 256, 4096}, {383, 256, 128}};
    const bool timing_only = getenv("LECLIP_G384_BENCH_ONLY") != nullptr;
    if (!timing_only) for (int dt : {LECLIP_F16, LECLIP_BF16}) for (auto& c : cases) {
        const int64_t M = c.M; const int N = c.N, K = c.K;
        auto A = randn(M * K), W = randn((size_t)N * K, 1.f / std::sqrt((float)K)), Bv = randn(N), R = randn(M * N), CS = randn(N);
        std::vector<float> ST(2 * M);
        { std::uniform_real_distribution<float> u(0.5f, 1.5f); std::normal_distribution<float> n(0.f, 0.1f); for (int64_t m = 0; m < M; ++m) { ST[2 * m] = n(rng); ST[2 * m + 1] = u(rng); } }
        auto Ap = pack(A, dt), Wp = pack(W, dt), Rp = pack(R, dt);
        const size_t ybytes = (size_t)(M + 8) * N * 2, pbytes = (size_t)(N / 64) * M * 8 + 64;
        Buf dA(Ap.size()), dW(Wp.size()), dB(N * 4), dR(Rp.size()), dCS(N * 4), dST(ST.size() * 4), dY(ybytes), dP(pbytes);
        dA.up(Ap.data()); dW.up(Wp.data()); dB.up(Bv.data()); dR.up(Rp.data()); dCS.up(CS.data()); dST.up(ST.data());
        struct F { const char* name; bool res, stats, ln; int act; } flav[] = {{"res+stats", true, true, false, 0}, {"res", true, false, false, 0}, {"ln", false, false, true, 0},
                                                                              {"ln+gelu", false, false, true, 1}, {"bias", false, false, false, 0}, {"bias+gelu", false, false, false, 1}};
        for (auto& f : flav) {
            std::vector<uint8_t> y[2], pp[2];
            int rc = 0;
            for (int fam = 0; fam < 2; ++fam) {
                HIPCHK(hipMemset(dY.d, 0xEE, ybytes)); HIPCHK(hipMemset(dP.d, 0xEE, pbytes));
                leclip_set_gemm_family(fam ? 384 : 256);
                rc |= leclip_gemm_ln_fused_fwd(dA.d, dW.d, (float*)dB.d, f.ln ? (float*)dST.d : nullptr, f.ln ? (float*)dCS.d : nullptr, f.res ? dR.d : nullptr, dY.d,
                                               f.stats ? (float*)dP.d : nullptr, M, N, K, K, K, N, N, (leclip_act)f.act, (leclip_dtype)dt, (leclip_dtype)dt, (leclip_dtype)dt, nullptr);
                HIPCHK(hipDeviceSynchronize());
                leclip_set_gemm_family(-1);
                y[fam].resize(ybytes); pp[fam].resize(pbytes);
                dY.down(y[fam].data()); dP.down(pp[fam].data());
            }
            size_t bad = 0, first = 0;
            for (size_t i = 0; i < ybytes; ++i) if (y[0][i] != y[1][i]) { if (!bad) first = i; ++bad; }
            size_t badp = 0;
            for (size_t i = 0; i < pbytes; ++i) if (pp[0][i] != pp[1][i]) ++badp;
            // guard rows (past M) and the partials' tail must still hold the fill pattern
            size_t guard = 0;
            for (size_t i = (size_t)M * N * 2; i < ybytes; ++i) if (y[1][i] != 0xEE) ++guard;
            for (size_t i = (size_t)(N / 64) * M * 8; i < pbytes; ++i) if (pp[1][i] != 0xEE) ++guard;
            const bool ok = !rc && !bad && !badp && !guard;
            printf("%s g384 %-9s %s M=%lld N=%d K=%d: rc %d, %zu output bytes differ (first at element %zu = row %zu col %zu), %zu partial bytes differ, %zu guard bytes touched\n",
                   ok ? "ok  " : "FAIL", f.name, dtn(dt), (long long)M, N, K, rc, bad, first / 2, first / 2 / N, first / 2 % N, badp, guard);
            if (!ok) g_fail++;
        }
    }
    printf(g_fail ? "FAILED %d checks\n" : "ALL OK\n", g_fail);
    if (g_fail) return 1;
    const int64_t Ms[2] = {256 * 197, 128 * 197};
    struct S { int N, K, act; bool res, ln; const char* name; } shapes[] = {
        {2304, 768, 0, false, true, "qkv(ln)"}, {768, 768, 0, true, false, "out_proj(res+stats)"}, {3072, 768, 1, false, true, "c_fc(ln+gelu)"}, {768, 3072, 0, true, false, "c_proj(res+stats)"}};
    for (int64_t M : Ms) for (int dt : {LECLIP_F16}) for (auto& s : shapes) {
        auto A = randn(M * s.K), W = randn((size_t)s.N * s.K, 0.03f), Bv = randn(s.N), R = randn(M * s.N), CS = randn(s.N);
        std::vector<float> ST(2 * M, 1.f);
        auto Ap = pack(A, dt), Wp = pack(W, dt), Rp = pack(R, dt);
        Buf dA(Ap.size()), dW(Wp.size()), dB(s.N * 4), dR(Rp.size()), dCS(s.N * 4), dST(ST.size() * 4), dY((size_t)M * s.N * 2), dP((size_t)(s.N / 64) * M * 8);
        dA.up(Ap.data()); dW.up(Wp.data()); dB.up(Bv.data()); dR.up(Rp.data()); dCS.up(CS.data()); dST.up(ST.data());
        double t[2][3];
        for (int round = 0; round < 3; ++round) for (int fam = 0; fam < 2; ++fam) {
            leclip_set_gemm_family(fam ? 384 : 256);
            t[fam][round] = time_ms(20, [&] {
                leclip_gemm_ln_fused_fwd(dA.d, dW.d, (float*)dB.d, s.ln ? (float*)dST.d : nullptr, s.ln ? (float*)dCS.d : nullptr, s.res ? dR.d : nullptr, dY.d,
                                         s.res ? (float*)dP.d : nullptr, M, s.N, s.K, s.K, s.K, s.N, s.N, (leclip_act)s.act, (leclip_dtype)dt, (leclip_dtype)dt, (leclip_dtype)dt, nullptr);
            });
            leclip_set_gemm_family(-1);
        }
        printf("bench %-20s %s M=%lld: 256x256 %.1f / %.1f / %.1f us   384x256 %.1f / %.1f / %.1f us   (%.0f -> %.0f TFLOP/s)\n", s.name, dtn(dt), (long long)M, t[0][0] * 1e3,
               t[0][1] * 1e3, t[0][2] * 1e3, t[1][0] * 1e3, t[1][1] * 1e3, t[1][2] * 1e3, 2.0 * M * s.N * s.K / t[0][2] * 1e-9, 2.0 * M * s.N * s.K / t[1][2] * 1e-9);
    }
    return 0;
}

int main(int argc, char** argv) {
    printf("leclip ABI %d\n", leclip_abi_version());
    if (argc > 1 && !strcmp(argv[1], "bench")) { bench(); return 0; }
    if (argc > 1 && !strcmp(argv[1], "stamps")) { stamps(); return 0; }
    if (argc > 1 && !strcmp(argv[1], "attn")) { return attn(); }
    if (argc > 1 && !strcmp(argv[1], "g384")) { return g384(); }
    if (argc > 1 && !strcmp(argv[1], "attnstamps")) { return attnstamps(); }
    check_gemm_identity();
    for (int dt : {LECLIP_BF16, LECLIP_F16}) {
        check_gemm(200, 128, 64, dt, LECLIP_F32, LECLIP_F32, 0, false, false);
        check_gemm(1576, 768, 768, dt, dt, dt, 0, true, true);
        check_gemm(333, 384, 192, dt, LECLIP_F32, dt, 1, true, false);
        check_gemm(1576, 256, 3072, dt, LECLIP_F32, LECLIP_F32, 1, true, true);
    }
    // shapes that qualify for the 256x256 ping-pong kernel when LECLIP_GEMM_TILE=256 (K-tiles: 2, 3, 12; ragged M)
    for (int dt : {LECLIP_BF16, LECLIP_F16}) {
        check_gemm(700, 512, 128, dt, LECLIP_F32, LECLIP_F32, 0, false, false);
        check_gemm(1000, 256, 192, dt, dt, dt, 1, true, true);
        check_gemm(513, 768, 768, dt, dt, dt, 0, true, true);
        check_gemm(256, 256, 256, dt, LECLIP_F32, LECLIP_F32, 0, true, false);
    }
    check_gemm(200, 64, 32, LECLIP_F32, LECLIP_F32, LECLIP_F32, 0, false, false);
    check_gemm(777, 192, 160, LECLIP_F32, LECLIP_F32, LECLIP_F32, 1, true, true);
    for (int dt : {LECLIP_BF16, LECLIP_F16, LECLIP_F32}) {
        check_attn(2, 197, 3, dt, 0);
        check_attn(3, 224, 2, dt, 0);
        check_attn(2, 200, 2, dt, 1);
        if (dt != LECLIP_F32) { check_attn(1, 577, 2, dt, 0); check_attn(1, 300, 1, dt, 1); check_attn(1, 640, 1, dt, 0); check_attn(1, 577, 2, dt, 0, 1); check_attn(1, 600, 1, dt, 1, 1); }
        check_attn(3, 77, 2, dt, 1);
        check_attn(2, 17, 2, dt, 0);
        check_attn(1, 50, 1, dt, 1);
    }
    printf(g_fail ? "FAILED %d checks\n" : "ALL OK\n", g_fail);
    return g_fail ? 1 : 0;
}
