// Rate of f64 atomic adds into a map of `cells` doubles (default 75 021 x 4: four heat maps of the headline shape), by how
// the lanes of one instruction are placed:  G = 1: every lane its own random cell (what the tile binner's deposits look
// like);  G = 4 / 8: groups of G neighbouring lanes add to G CONSECUTIVE doubles at a random place (a run's deposits for G
// maps of a map-minor layout H[cell][map] carried by one instruction).  Also `active`: how many lanes of the wave take
// part (the deposits of a wave-row are a handful of run heads).  Usage: atomic_coalesce [cells] [iterations]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ unsigned int rng(unsigned int& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int G>
__global__ void __launch_bounds__(256) adds(double* H, unsigned int groups_in_map, int iters, int active) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (lane >= active) return;
    unsigned int s = 1234567u + 977u * (unsigned int)wave + 31u * (unsigned int)(lane / G);
    for (int i = 0; i < iters; ++i) {
        const unsigned int g = rng(s) % groups_in_map;
        atomicAdd(&H[(size_t)g * G + (lane % G)], 1.0);
    }
}

int main(int argc, char** argv) {
    const size_t cells = argc > 1 ? (size_t)atoll(argv[1]) : 75021u * 4u;
    const int iters = argc > 2 ? atoi(argv[2]) : 2000;
    double* H;
    if (hipMalloc(&H, cells * sizeof(double)) != hipSuccess) return 1;
    hipMemset(H, 0, cells * sizeof(double));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 4096;
    for (int active : {64, 32, 16, 8}) {
        for (int G : {1, 4, 8}) {
            if (active % G) continue;
            auto launch = [&]() {
                if (G == 1) hipLaunchKernelGGL(adds<1>, dim3(blocks), dim3(256), 0, 0, H, (unsigned int)(cells / 1), iters, active);
                if (G == 4) hipLaunchKernelGGL(adds<4>, dim3(blocks), dim3(256), 0, 0, H, (unsigned int)(cells / 4), iters, active);
                if (G == 8) hipLaunchKernelGGL(adds<8>, dim3(blocks), dim3(256), 0, 0, H, (unsigned int)(cells / 8), iters, active);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            const double n_adds = (double)blocks * 4 * active * iters;
            printf("active lanes %2d, %d consecutive doubles per group: %7.2f ms, %6.2f G adds/s, %6.2f G groups/s\n",
                   active, G, ms, n_adds / ms / 1e6, n_adds / G / ms / 1e6);
        }
    }
    if (hipGetLastError() != hipSuccess) { printf("error\n"); return 1; }
    return 0;
}
