// Micro-benchmark (not part of the product): cost of one pcg64 step on gfx950, to know the ALU floor
// of the read generator (~1206 steps per read pair).   hipcc --offload-arch=gfx950 -O3 -o ubench tools/ubench_pcg.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../jackalope_amd/csrc/jk_math.h"

template <int MODE>
__global__ void __launch_bounds__(1024) k(uint64_t* out, int n_steps) {
    uint32_t w[8];
    for (int i = 0; i < 8; i++) w[i] = threadIdx.x * 7919u + blockIdx.x * 104729u + i;
    jk_pcg64 e = jk_pcg_seed(w);
    uint64_t acc = 0;
    for (int i = 0; i < n_steps; i++) {
        uint64_t x = jk_pcg_next(e);
        if (MODE == 0) acc ^= x;
        if (MODE == 1) acc += (x >= 0xfffcb923a29c779aULL) ? 1 : ((x >= 0xfffe1a3e0e7c0000ULL) ? 3 : 7);   // indel-like compares
        if (MODE == 2) acc += jk_runif_index(x, 5 + (acc & 3));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    uint64_t* d; hipMalloc(&d, 8 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int steps = 20000, blocks = 1024;   // 1M lanes
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(1024), 0, 0, d, steps);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(1024), 0, 0, d, steps);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(1024), 0, 0, d, steps);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            double draws = (double)steps * blocks * 1024;
            if (rep) printf("mode %d: %.3f ms, %.3e draws/s, %.1f SIMD-cycles per wave-draw (at 2.4 GHz)\n", mode, ms, draws / (ms * 1e-3),
                            2.4e9 * 1024 / (draws / 64 / (ms * 1e-3)));
        }
    }
    return 0;
}
