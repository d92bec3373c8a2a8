// Probe: does gfx950 execute scalar memory atomics (s_atomic_add with return), and what does one cost?  Every wave of every workgroup
// draws `n` tickets from one counter; the host checks that the tickets are a permutation of 0 .. total-1 and prints the mean latency.
// Build: hipcc -O2 --offload-arch=gfx950 satomic_probe.hip -o ../../lib/satomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__global__ void draw(unsigned* counter, unsigned* out, unsigned long long* cycles, int n) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned long long t = 0;
    for (int i = 0; i < n; ++i) {
        unsigned tk;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_mov_b32 %0, 1\n\ts_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(tk) : "s"(counter) : "memory");
        t += __builtin_amdgcn_s_memtime() - t0;
        if ((threadIdx.x & 63) == 0) out[wave * n + i] = tk;
    }
    if ((threadIdx.x & 63) == 0) cycles[wave] = t;
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 256, tpb = argc > 2 ? atoi(argv[2]) : 512, waves = wgs * tpb / 64, n = 16, total = waves * n;
    unsigned *counter, *out;
    unsigned long long* cyc;
    hipMalloc(&counter, 4); hipMalloc(&out, total * 4); hipMalloc(&cyc, waves * 8);
    hipMemset(counter, 0, 4);
    hipLaunchKernelGGL(draw, dim3(wgs), dim3(tpb), 0, 0, counter, out, cyc, n);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<unsigned> h(total); std::vector<unsigned long long> c(waves);
    hipMemcpy(h.data(), out, total * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    bool ok = true;
    for (int i = 0; i < total; ++i) ok = ok && h[i] == (unsigned)i;
    double s = 0; for (auto v : c) s += (double)v;
    printf("tickets %s (%d draws by %d waves), mean latency %.0f shader-clock ticks per draw (%d workgroups x %d threads drawing at once)\n", ok ? "unique 0..N-1" : "WRONG", total, waves, s / total, wgs, tpb);
    return ok ? 0 : 1;
}
