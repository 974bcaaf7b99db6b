// Shader clock under load: every wave runs a dependent chain of FP64 FMAs and records the shader-clock cycles
// (s_memtime) and the constant 100 MHz real-time counter (s_memrealtime) it took.  cycles / real time = the clock the
// CUs actually ran at.  Usage: clock_load [waves_per_simd 1..8] [iterations]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ void __launch_bounds__(256) burn(double* out, long long* cyc, long long* rt, int iters, double a, double b) {
    double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    const long long c0 = clock64(), r0 = wall_clock64();
    for (int i = 0; i < iters; i += 16) {
#pragma unroll
        for (int rep = 0; rep < 16; ++rep) { x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b); }
    }
    const long long c1 = clock64(), r1 = wall_clock64();
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    out[gid] = x0 + x1 + x2 + x3;
    if ((threadIdx.x & 63) == 0) { cyc[gid >> 6] = c1 - c0; rt[gid >> 6] = r1 - r0; }
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 4, iters = argc > 2 ? atoi(argv[2]) : 2000000;
    for (int blocks : {1, 256 * wps}) {          // one workgroup alone, then wps waves on every SIMD of 256 CUs
        const int n = blocks * 256;
        double* out; long long *cyc, *rt;
        hipMalloc(&out, n * sizeof(double)); hipMalloc(&cyc, n / 64 * 8); hipMalloc(&rt, n / 64 * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        burn<<<blocks, 256>>>(out, cyc, rt, 1000, 0.999, 0.001);
        hipEventRecord(e0);
        burn<<<blocks, 256>>>(out, cyc, rt, iters, 0.999, 0.001);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> c(n / 64), r(n / 64);
        hipMemcpy(c.data(), cyc, n / 64 * 8, hipMemcpyDeviceToHost); hipMemcpy(r.data(), rt, n / 64 * 8, hipMemcpyDeviceToHost);
        double cs = 0, rs = 0; for (int i = 0; i < n / 64; ++i) { cs += c[i]; rs += r[i]; }
        const double flops = 2.0 * 4 * iters * (double)n / (ms * 1e-3);
        printf("blocks %5d: %.2f ms, memtime ticks per wave %.3e, realtime ticks (100 MHz) per wave %.3e -> memtime rate %.1f MHz; "
               "%.2f TFLOP/s fp64; cycles per wave FMA if clock = 2400 MHz: %.2f\n",
               blocks, ms, cs / (n / 64), rs / (n / 64), cs / rs * 100.0, flops / 1e12,
               ms * 1e-3 * 2.4e9 / (4.0 * iters) / (blocks >= 256 ? wps : 1));
        hipFree(out); hipFree(cyc); hipFree(rt);
    }
    return 0;
}
