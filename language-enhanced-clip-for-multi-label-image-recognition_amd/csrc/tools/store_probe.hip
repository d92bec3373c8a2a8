// Microbenchmark (diagnostic, not part of the library): how fast can one CU / the whole chip write a 256x256 bf16
// output tile, depending on the shape of each wave-instruction's footprint?  Mirrors the GEMM epilogue's store stream
// without any compute.  Build: hipcc --offload-arch=gfx950 -O3 store_probe.hip -o store_probe ; run: store_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

// pattern 0: wave-instr = 8 rows x 128 B (GEMM epilogue today: wave owns a 64-column strip)
// pattern 1: wave-instr = 2 rows x 512 B (whole tile rows)
// pattern 2: wave-instr = 1 KiB contiguous (tile-blocked output)
// pattern 3: as 0, non-temporal
// pattern 4: as 0 but 8-byte stores (twice the instructions)
// pattern 5: wave-instr = 16 rows x 32 B (8 B per lane; what a store straight from a transposed MFMA accumulator block gives)
template <int P>
__global__ __launch_bounds__(512) void store_tiles(char* out, int64_t ld_bytes, int tiles_n, int tiles_total, int v0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    i32x4 val = {lane, wave, (int)blockIdx.x, v0};
    for (int t = blockIdx.x; t < tiles_total; t += gridDim.x) {
        const int tm = t / tiles_n, tn = t - tm * tiles_n;
        char* base = out + (int64_t)tm * 256 * ld_bytes + (int64_t)tn * 512;
        if (P == 0 || P == 3 || P == 4) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = wm * 128 + q * 8 + (lane >> 3);
                char* p = base + (int64_t)row * ld_bytes + wn * 128 + (lane & 7) * 16;
                if (P == 0) *(i32x4*)p = val;
                if (P == 3) __builtin_nontemporal_store(val, (i32x4*)p);
                if (P == 4) { *(i32x2*)p = i32x2{val[0], val[1]}; *(i32x2*)(p + 8) = i32x2{val[2], val[3]}; }
            }
        } else if (P == 5) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = wm * 128 + i * 16 + (lane & 15);
                    char* p = base + (int64_t)row * ld_bytes + wn * 128 + j * 32 + (lane >> 4) * 8;
                    *(i32x2*)p = i32x2{val[0], val[1]};
                }
        } else if (P == 1) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = wave * 32 + q * 2 + (lane >> 5);
                char* p = base + (int64_t)row * ld_bytes + (lane & 31) * 16;
                *(i32x4*)p = val;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                char* p = out + ((int64_t)t * 128 + wave * 16 + q) * 1024 + lane * 16;
                *(i32x4*)p = val;
            }
        }
        val[3] += 1;
    }
}

template <int P>
static float run(char* out, int64_t ld, int tiles_n, int tiles, int grid, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(store_tiles<P>, dim3(grid), dim3(512), 0, 0, out, ld, tiles_n, tiles, 0);
    hipEventRecord(a, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(store_tiles<P>, dim3(grid), dim3(512), 0, 0, out, ld, tiles_n, tiles, r);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const int64_t M = 50432;
    const int N = 2304, tiles_n = N / 256, tiles = (int)(M / 256) * tiles_n;
    char* out = nullptr;
    if (hipMalloc(&out, (size_t)M * N * 2) != hipSuccess) return 1;
    const char* names[6] = {"8x128B", "2x512B", "1KiB", "8x128B-nt", "8x128B-b64", "16x32B-b64"};
    for (int grid : {256, 64, 32, 8}) {
        float ms[6];
        ms[0] = run<0>(out, (int64_t)N * 2, tiles_n, tiles, grid, 10);
        ms[1] = run<1>(out, (int64_t)N * 2, tiles_n, tiles, grid, 10);
        ms[2] = run<2>(out, (int64_t)N * 2, tiles_n, tiles, grid, 10);
        ms[3] = run<3>(out, (int64_t)N * 2, tiles_n, tiles, grid, 10);
        ms[4] = run<4>(out, (int64_t)N * 2, tiles_n, tiles, grid, 10);
        ms[5] = run<5>(out, (int64_t)N * 2, tiles_n, tiles, grid, 10);
        for (int p = 0; p < 6; ++p) {
            const double bytes = (double)tiles * 131072.0;
            printf("grid %3d %-11s %.3f ms  %.0f GB/s  %.2f us/tile/CU  %.1f B/ns/CU\n", grid, names[p], ms[p], bytes / ms[p] * 1e-6,
                   ms[p] * 1e3 / ((tiles + grid - 1) / grid), bytes / grid / (ms[p] * 1e6));
        }
    }
    hipFree(out);
    return 0;
}
