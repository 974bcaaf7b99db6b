// Issue cadence of one wave alone on its SIMD, per kind of vector instruction: N independent chains per lane, cycles
// (s_memtime) per instruction.  Usage: op_cadence [iterations]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int OP, int CH>
__global__ void __launch_bounds__(64) chain(double* out, long long* cyc, int iters, double a, double b) {
    double x[CH]; float y[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) { x[k] = threadIdx.x * 1e-3 + k; y[k] = (float)x[k]; }
    const float af = (float)a, bf = (float)b;
    const long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int rep = 0; rep < 16; ++rep) {          // 16 rounds per trip: the loop's own scalar work and taken branch are amortised
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            if (OP == 0) x[k] = fma(x[k], a, b);
            if (OP == 1) x[k] = x[k] * a;
            if (OP == 2) x[k] = x[k] + b;
            if (OP == 3) y[k] = fmaf(y[k], af, bf);
            if (OP == 4) x[k] = fmax(x[k], b);
            if (OP == 5) asm volatile("v_rcp_f64 %0, %0" : "+v"(x[k]));
            if (OP == 6) { long long t = __double_as_longlong(x[k]); t += 3; x[k] = __longlong_as_double(t); }
            if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(y[k]) : "v"(af));
            if (OP == 8) asm volatile("s_nop 0");
        }
      }
    }
    const long long c1 = clock64();
    double s = 0; for (int k = 0; k < CH; ++k) s += x[k] + y[k];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = c1 - c0;
}

template <int OP, int CH>
static void run(const char* name, int iters) {
    double* out; long long* cyc; long long h;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 8);
    chain<OP, CH><<<1, 64>>>(out, cyc, 100, 0.999, 0.001);
    chain<OP, CH><<<1, 64>>>(out, cyc, iters, 0.999, 0.001);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s chains %d: %6.2f cycles per instruction\n", name, CH, (double)h / ((double)iters * CH * 16));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    run<0, 1>("v_fma_f64 (dependent)", iters);  run<0, 4>("v_fma_f64", iters);  run<0, 8>("v_fma_f64", iters);
    run<1, 1>("v_mul_f64 (dependent)", iters);  run<1, 4>("v_mul_f64", iters);  run<1, 8>("v_mul_f64", iters);
    run<2, 1>("v_add_f64 (dependent)", iters);  run<2, 4>("v_add_f64", iters);  run<2, 8>("v_add_f64", iters);
    run<3, 1>("v_fma_f32 (dependent)", iters);  run<3, 4>("v_fma_f32", iters);  run<3, 8>("v_fma_f32", iters);
    run<4, 1>("v_max_f64 (dependent)", iters);  run<4, 4>("v_max_f64", iters);
    run<5, 1>("v_rcp_f64 (dependent)", iters);  run<5, 4>("v_rcp_f64", iters);
    run<6, 1>("v_add_u64 (dependent)", iters);  run<6, 4>("v_add_u64 / co pair", iters);
    run<7, 4>("v_cndmask_b32", iters);
    run<8, 1>("s_nop", iters);
    return 0;
}
