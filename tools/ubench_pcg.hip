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

// the hand-scheduled device step against the plain __int128 expression, 2^14 lanes x 4096 steps
__global__ void check(uint32_t* bad) {
    uint32_t w[8];
    for (int i = 0; i < 8; i++) w[i] = (threadIdx.x + blockIdx.x * blockDim.x) * 2654435761u + i * 40503u;
    jk_pcg64 a = jk_pcg_seed(w), b = a;
    for (int i = 0; i < 4096; i++) {
        const uint64_t x = jk_pcg_next(a), y = jk_pcg_next_ref(b);
        if (x != y || a.s_hi != b.s_hi || a.s_lo != b.s_lo) atomicAdd(bad, 1u);
    }
}

int main() {
    uint64_t* d; hipMalloc(&d, 8 << 20);
    {
        uint32_t* bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(check, dim3(64), dim3(256), 0, 0, bad);
        uint32_t h = 1; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
        printf("asm step vs __int128 step: %u mismatches\n", h);
    }
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
