// Multi-crop test path: every sliding window of an image goes through the reference's test transform
// (torchvision Resize(S, bicubic) on the smaller edge -> CenterCrop(S) -> ToTensor -> Normalize,
// dassl/data/transforms/transforms.py:379-400) on the device, straight from the raw uint8 image to the [windows, 3, S, S] batch the
// image tower consumes.  The reference does this per window on the host through Pillow (dassl/data/data_manager.py:392-399
// `tfm(F.to_pil_image(block))`, ~570 windows per 480x640 image); the resize is therefore Pillow's 8-bit resampler
// (src/libImaging/Resample.c), reproduced BIT FOR BIT: bicubic (a = -0.5) coefficients evaluated in IEEE double in the
// library's operation order (no FMA contraction: __dmul_rn / __dadd_rn / __ddiv_rn), normalised, converted to 22-bit fixed
// point, horizontal pass -> uint8 -> vertical pass -> uint8, then /255, -mean, /std in correctly rounded fp32.
//
// Byte work, HBM / L2 bound and tiny next to the forward pass it feeds (one 16x16 output tile per 256-thread workgroup: the
// tile's coefficient sets in LDS, the horizontally resampled source rows it needs in LDS as uint8, one output pixel per thread).
#include "leclip_common.h"

namespace {

constexpr int PBITS = 32 - 8 - 2;     // Pillow PRECISION_BITS
constexpr int KMAX = 64;              // taps per output sample: ceil(2 * scale) * 2 + 1 <= 63  <=>  scale <= 15.5
constexpr int NROWS_MAX = 16 * 16 + KMAX;

struct CropArgs {
    const uint8_t* src;     // [B, 3, H, W]
    const int* win;         // [NW, 5] = y0, x0, bh, bw, pad_top (rows counted on the reflect-padded image)
    void* out;              // [B, NW, 3, S, S]
    int H, W, NW, S, out_dt;
    float mean[3], stdv[3];
};

__device__ __forceinline__ double bicubic(double x) {
    // Resample.c bicubic_filter, a = -0.5, in its operation order
    if (x < 0.0) x = -x;
    if (x < 1.0) return __dadd_rn(__dmul_rn(__dmul_rn(__dadd_rn(__dmul_rn(1.5, x), -2.5), x), x), 1.0);
    if (x < 2.0) return __dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(x, -5.0), x), 8.0), x), -4.0), -0.5);
    return 0.0;
}

// precompute_coeffs + normalize_coeffs_8bpc for ONE output index xx: taps into k[0..cnt), returns first input index
__device__ __forceinline__ int coeffs_for(int in_size, int out_size, int xx, int* k, int& cnt) {
    if (in_size == out_size) {      // Pillow skips the pass when the size does not change: identity
        k[0] = 1 << PBITS;
        cnt = 1;
        return xx;
    }
    const double scale = __ddiv_rn((double)in_size, (double)out_size);
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = __dmul_rn(2.0, filterscale);
    const double ss = __ddiv_rn(1.0, filterscale);
    const double center = __dmul_rn((double)xx + 0.5, scale);
    int xmin = (int)__dadd_rn(__dadd_rn(center, -support), 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)__dadd_rn(__dadd_rn(center, support), 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    if (xmax > KMAX) xmax = KMAX;      // (host-side validation keeps scale <= 15.5; this only guards the LDS array)
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x)
        ww = __dadd_rn(ww, bicubic(__dmul_rn(__dadd_rn(__dadd_rn((double)(x + xmin), -center), 0.5), ss)));
    for (int x = 0; x < xmax; ++x) {
        double w = bicubic(__dmul_rn(__dadd_rn(__dadd_rn((double)(x + xmin), -center), 0.5), ss));
        if (ww != 0.0) w = __ddiv_rn(w, ww);
        k[x] = w < 0.0 ? (int)__dadd_rn(-0.5, __dmul_rn(w, 4194304.0)) : (int)__dadd_rn(0.5, __dmul_rn(w, 4194304.0));
    }
    cnt = xmax;
    return xmin;
}

__device__ __forceinline__ int clip8(int v) {
    v >>= PBITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(256) void crop_resize_kernel(CropArgs a) {
    __shared__ int hk[16][KMAX], vk[16][KMAX];
    __shared__ int hmin[16], hcnt[16], vmin[16], vcnt[16];
    __shared__ unsigned char tmp[3][NROWS_MAX][16];

    const int tid = threadIdx.x;
    const int tiles = (a.S + 15) >> 4;
    const int ty = blockIdx.x / tiles, tx = blockIdx.x - ty * tiles;
    const int wi = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int* w5 = a.win + 5 * wi;
    const int y0 = w5[0], x0 = w5[1], top = w5[4];
    int bh = w5[2], bw = w5[3];
    bh = bh < 1 ? 1 : bh;
    bw = bw < 1 ? 1 : bw;
    // torchvision 0.12 _compute_resized_output_size: smaller edge -> S, longer = int(S * long / short)
    int nh, nw;
    if (bw <= bh) { nw = a.S; nh = (int)((int64_t)a.S * bh / bw); }
    else { nh = a.S; nw = (int)((int64_t)a.S * bw / bh); }
    // CenterCrop: int(round((n - S) / 2.0)), Python's round-half-to-even
    auto half_even = [](int d) { const int q = d >> 1; return (d & 1) ? q + (q & 1) : q; };
    const int ct = half_even(nh - a.S), cl = half_even(nw - a.S);

    if (tid < 16) {
        int oy = ty * 16 + tid;
        oy = oy < a.S ? oy : a.S - 1;
        int c;
        vmin[tid] = coeffs_for(bh, nh, oy + ct, vk[tid], c);
        vcnt[tid] = c;
    } else if (tid >= 64 && tid < 80) {
        const int i = tid - 64;
        int ox = tx * 16 + i;
        ox = ox < a.S ? ox : a.S - 1;
        int c;
        hmin[i] = coeffs_for(bw, nw, ox + cl, hk[i], c);
        hcnt[i] = c;
    }
    __syncthreads();
    int rlo = vmin[0], rhi = vmin[0] + vcnt[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) {
        rlo = vmin[i] < rlo ? vmin[i] : rlo;
        const int e = vmin[i] + vcnt[i];
        rhi = e > rhi ? e : rhi;
    }
    int nrows = rhi - rlo;
    nrows = nrows > NROWS_MAX ? NROWS_MAX : nrows;

    // ---- horizontal pass for the rows this tile needs: (channel, row, column) items over the 256 threads
    const uint8_t* img = a.src + b * 3 * (int64_t)a.H * a.W;
    for (int it = tid; it < 3 * nrows * 16; it += 256) {
        const int x = it & 15, r = (it >> 4) % nrows, c = (it >> 4) / nrows;
        int sr = y0 + rlo + r - top;                 // padded row -> source row: reflect without repeating the edge
        sr = sr < 0 ? -sr : sr;
        sr = sr > a.H - 1 ? 2 * (a.H - 1) - sr : sr;
        sr = sr < 0 ? 0 : (sr > a.H - 1 ? a.H - 1 : sr);        // (memory safety for malformed windows)
        const uint8_t* row = img + ((int64_t)c * a.H + sr) * a.W;
        int acc = 1 << (PBITS - 1);
        const int n = hcnt[x], base = x0 + hmin[x];
        for (int k = 0; k < n; ++k) {
            int sx = base + k;
            sx = sx > a.W - 1 ? a.W - 1 : (sx < 0 ? 0 : sx);    // (memory safety)
            acc += (int)row[sx] * hk[x][k];
        }
        tmp[c][r][x] = (unsigned char)clip8(acc);
    }
    __syncthreads();

    // ---- vertical pass + ToTensor + Normalize: one output pixel per thread
    const int oy = ty * 16 + (tid >> 4), ox = tx * 16 + (tid & 15);
    if (oy >= a.S || ox >= a.S) return;
    const int ly = tid >> 4, lx = tid & 15;
    const int n = vcnt[ly], r0 = vmin[ly] - rlo;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int acc = 1 << (PBITS - 1);
        for (int k = 0; k < n; ++k) {
            const int r = r0 + k;
            acc += (int)tmp[c][r < NROWS_MAX ? r : NROWS_MAX - 1][lx] * vk[ly][k];
        }
        const float u = (float)clip8(acc);
        const float f = __fdiv_rn(__fsub_rn(__fdiv_rn(u, 255.0f), a.mean[c]), a.stdv[c]);
        store_elem(a.out, a.out_dt, (((b * a.NW + wi) * 3 + c) * a.S + oy) * (int64_t)a.S + ox, f);
    }
}

}  // namespace

extern "C" int leclip_crop_resize_fwd(const uint8_t* src, int64_t B, int H, int W, const int32_t* windows, int NW, void* out, int S,
                                      const float* mean3, const float* std3, leclip_dtype out_dtype, void* stream) {
    if (!src || !windows || !out || !mean3 || !std3 || B <= 0 || H <= 0 || W <= 0 || NW <= 0 || S <= 0 || !dtype_ok(out_dtype)) {
        leclip_set_error("crop_resize: null pointer or bad size");
        return LECLIP_E_INVALID;
    }
    if (B > 65535 || NW > 65535) { leclip_set_error("crop_resize: at most 65535 images / windows per call"); return LECLIP_E_UNSUPPORTED; }
    CropArgs a;
    a.src = src; a.win = windows; a.out = out; a.H = H; a.W = W; a.NW = NW; a.S = S; a.out_dt = (int)out_dtype;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean3[c]; a.stdv[c] = std3[c]; }     // host pointers: six scalars of the transform
    const int tiles = (S + 15) / 16;
    hipLaunchKernelGGL(crop_resize_kernel, dim3((unsigned)(tiles * tiles), (unsigned)NW, (unsigned)B), dim3(256), 0, (hipStream_t)stream, a);
    return leclip_check_launch("crop_resize_kernel");
}
