// tools/probes/lone_wave3.hip -- what ONE wave pays to issue a global store vs an LDS write per loop iteration (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 20000
__global__ void probe(double* out, long long* cyc, int mode, int n, int lanes) {
    __shared__ double lds[64 * 64];
    if ((int)threadIdx.x >= lanes) return;
    double a = 1.25 + threadIdx.x * 1e-9, w = 1.0000001;
    double* p = out + threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        a = fma(a, w, 1e-9); a = fma(a, w, 1e-9); a = fma(a, w, 1e-9); a = fma(a, w, 1e-9);     // 4 dependent FMAs of "work"
        if (mode == 1) { p[(size_t)i * 64] = a; }
        else if (mode == 2) { p[(size_t)i * 64] = a; p[(size_t)(n + i) * 64] = a + 1; }
        else if (mode == 3) { lds[(i & 63) * 64 + threadIdx.x] = a; }
        else if (mode == 4) { lds[(i & 31) * 64 + threadIdx.x] = a; lds[(32 + (i & 31)) * 64 + threadIdx.x] = a + 1; }
    }
    long long t1 = __builtin_readcyclecounter();
    out[(size_t)2 * n * 64 + threadIdx.x] = a + lds[threadIdx.x];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* out; long long* cyc;
    (void)hipMalloc(&out, (size_t)(2 * N + 2) * 64 * 8); (void)hipMalloc(&cyc, 16);
    const char* names[] = {"4 fma only", "+ 1 global store", "+ 2 global stores", "+ 1 LDS write", "+ 2 LDS writes"};
    for (int lanes : {64, 4}) for (int mode = 0; mode < 5; ++mode) {
        long long h = 0;
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc, mode, N, lanes); (void)hipDeviceSynchronize(); }
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("%2d lanes: %-20s %7.1f cycles per iteration\n", lanes, names[mode], (double)h / N);
    }
    return 0;
}
