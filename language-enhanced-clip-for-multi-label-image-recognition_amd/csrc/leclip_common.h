// Shared device helpers for the gfx950 kernels (wave64, MFMA 32x32x16, fp32 accumulate).
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

// Epilogue description shared by the GEMM kernels.
struct EpiParams {
    const float* bias;   // [N] or null
    const void* res;     // residual or null
    void* out;
    int64_t ldr, ldy;
    int res_dt, out_dt, act;
    int rowmap_P;        // patch-embed mode: >0 => out row = m + m/P + 1, residual row = m % P + 1
};

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
