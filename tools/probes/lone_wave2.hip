// tools/probes/lone_wave2.hip -- issue cost vs dependent latency of f64 vector instructions for ONE wave on a CU (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o lone_wave2 lone_wave2.hip && ./lone_wave2
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 4000
#define U 32

__global__ void probe(double* out, long long* cyc, int mode, int n, double seed, int lanes) {
    if ((int)threadIdx.x >= lanes) return;
    double a = seed + threadIdx.x * 1e-9, b = a + 1, c = a + 2, d = a + 3, w = 1.0000001, k = 1e-9;
    long long t0 = __builtin_readcyclecounter();
    if (mode == 0) { for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < U; ++u) a = fma(a, w, k); } }                                  // one dependent chain
    else if (mode == 1) { for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < U / 4; ++u) { a = fma(a, w, k); b = fma(b, w, k); c = fma(c, w, k); d = fma(d, w, k); } } }   // four chains
    else if (mode == 2) { for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < U; ++u) a = a * w + k; } }                                     // mul + add, dependent (contract off)
    else if (mode == 3) { for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < U; ++u) a = fmax(fmin(a * w, 1e300), -1e300); } }               // mul, min, max dependent
    else if (mode == 4) { for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < U; ++u) a = (a < b) ? a * w : c; } }                            // cmp, mul, select
    else if (mode == 5) { for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < U; ++u) a = (double)(int)(a * w) + 0.5; } }                      // mul, cvt, cvt, add
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    double* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 64 * 8); (void)hipMalloc(&cyc, 16);
    const char* names[] = {"fma, 1 chain", "fma, 4 chains", "mul+add chain (2 instr)", "mul+min+max chain (3 instr)",
                           "cmp+mul+select chain (3-4 instr)", "mul+cvt+cvt+add chain (4 instr)"};
    for (int lanes : {64, 4}) for (int mode = 0; mode < 6; ++mode) {
        long long h = 0;
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc, mode, N, 1.25, lanes); (void)hipDeviceSynchronize(); }
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("%2d lanes: %-36s %6.2f cycles per unrolled element\n", lanes, names[mode], (double)h / N / U);
    }
    return 0;
}
