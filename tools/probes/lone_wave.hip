// tools/probes/lone_wave.hip -- what one wavefront alone on a CU pays per dependent instruction (gfx950).
// The 999-ray fan runs one wave per CU, so its time is a chain of dependent latencies, not an issue rate.
//   hipcc --offload-arch=gfx950 -O3 -o lone_wave lone_wave.hip && ./lone_wave
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define N 20000

__global__ void probe(double* out, long long* cyc, int mode, int n, double seed) {
    __shared__ double lds[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = (double)((i * 7 + 3) & 2047);
    __syncthreads();
    double v = seed + threadIdx.x * 1e-9, w = 1.0000001;
    long long t0 = __builtin_readcyclecounter();       // s_memtime: shader cycles
    long long w0 = wall_clock64();                      // 100 MHz
    if (mode == 0) { for (int i = 0; i < n; ++i) v = fma(v, w, 1e-9); }                       // dependent f64 FMA
    else if (mode == 1) { for (int i = 0; i < n; ++i) v = 1.0 / v + 1.5; }                      // dependent f64 division
    else if (mode == 2) { for (int i = 0; i < n; ++i) v = sqrt(v) + 1.5; }                      // dependent f64 sqrt
    else if (mode == 3) { int j = threadIdx.x; for (int i = 0; i < n; ++i) j = (int)lds[j & 2047]; v = j; }   // dependent LDS read (+cvt)
    else if (mode == 4) { for (int i = 0; i < n; ++i) { if (v > 1e300) v = 0.5; v = v * w; } }  // compare + branch/select + mul
    else if (mode == 5) { for (int i = 0; i < n; ++i) { unsigned long long b = __ballot(v > 0.5); if (b == 0) v += 1.0; v = v * w; } }  // ballot -> scalar branch
    long long t1 = __builtin_readcyclecounter();
    long long w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 1024 * 256 * 8); hipMalloc(&cyc, 16);
    const char* names[] = {"f64 fma", "f64 div (+add)", "f64 sqrt (+add)", "LDS read (+cvt,and)", "cmp+select+mul", "ballot+sbranch+mul"};
    for (int blocks : {1, 256}) {
        for (int mode = 0; mode < 6; ++mode) {
            long long h[2];
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, out, cyc, mode, N, 1.25);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
            double ns = h[1] * 10.0;    // 100 MHz ticks
            printf("%3d wave(s), one per CU: %-22s %7.1f shader cycles / iteration, %7.2f ns / iteration, clock %.2f GHz\n",
                   blocks, names[mode], (double)h[0] / N, ns / N, h[0] / ns);
        }
    }
    return 0;
}
