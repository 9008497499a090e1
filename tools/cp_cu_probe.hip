// How fast is the pool compaction on N compute units, alone on the device?  (In a run it works beside the generator on
// the CUs a launch leaves free; this separates what a CU can do from what the generator's traffic costs it.)
// One launch's worth of one read end: 224 x 1024 lanes, 3600 bytes per lane.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o /tmp/cp_cu_probe tools/cp_cu_probe.hip && /tmp/cp_cu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../jackalope_amd/csrc/jk_illumina_kernel.h"
using namespace jk;
__global__ void __launch_bounds__(1024) blocker_kernel(uint64_t ticks) {
    extern __shared__ uint32_t blk_lds[];
    if (threadIdx.x == 0) blk_lds[0] = 1;
    asm volatile("v_mov_b32 v127, 0" ::: "v127");      // 128 VGPRs x 16 waves: the whole register file, like the generator -- nothing else fits on the CU
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
// the two halves of the compaction on their own: the tile rows read as the kernel reads them (xor-reduced, one word written
// per thread), and the lanes' pieces written as it writes them (from registers)
__global__ void __launch_bounds__(256) load_only_kernel(const uint8_t* pool, const uint64_t* pool_off, uint32_t rows, uint32_t* sink) {
    __shared__ uint32_t pad[16500];                 // (as much LDS as the compaction: cannot share a CU with a blocker either)
    pad[threadIdx.x] = rows;
    const uint4* src = reinterpret_cast<const uint4*>(pool + pool_off[blockIdx.x]);
    uint4 acc = {pad[(threadIdx.x + 1) & 255] & 0u, 0, 0, 0};
    for (uint32_t k = threadIdx.x; k < rows * 16u; k += 256u * 8u) {
        uint4 r[8];
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) r[i] = src[k + i * 256u];
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) { acc.x ^= r[i].x; acc.y ^= r[i].y; acc.z ^= r[i].z; acc.w ^= r[i].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}
template <bool ALIGNED>
__global__ void __launch_bounds__(256) store_only_kernel(uint8_t* out, const uint64_t* out_off, const uint64_t* lane_bytes) {
    __shared__ uint32_t pad[16500];
    pad[threadIdx.x] = blockIdx.x;
    const uint32_t t = threadIdx.x + (pad[(threadIdx.x + 1) & 255] & 0u), j = t % 32u, lsub = t / 32u;
    for (uint32_t pass = 0; pass < 8; pass++) {
        const uint32_t lane = blockIdx.x * 64u + pass * 8u + lsub;
        const uint32_t nb = (uint32_t)lane_bytes[lane];
        uint8_t* d = out + out_off[lane] + j * 16u;
        if (ALIGNED) d = reinterpret_cast<uint8_t*>(reinterpret_cast<uintptr_t>(d) & ~(uintptr_t)15);
        const uint4 v = {t, lane, pass, nb};
        for (uint32_t b0 = j * 16u; b0 + 16u <= nb; b0 += 512u, d += 512) __builtin_memcpy(d, &v, 16);
    }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const uint32_t n_lanes = 224 * 1024, n_tiles = n_lanes / 64;
    const uint64_t per_lane = 3600, cap = 3712;             // pool capacity per lane (multiple of 4)
    std::vector<uint64_t> pool_off(n_tiles + 1), lane_bytes(n_lanes, per_lane), out_off(n_lanes);
    for (uint32_t t = 0; t <= n_tiles; t++) pool_off[t] = (uint64_t)t * 64 * cap;
    for (uint32_t l = 0; l < n_lanes; l++) { lane_bytes[l] = per_lane - (l % 7) * 3; out_off[l] = l ? out_off[l - 1] + lane_bytes[l - 1] : 0; }
    const uint64_t pool_bytes = pool_off[n_tiles] + CP_SLACK, out_bytes = out_off[n_lanes - 1] + per_lane + 64;
    uint8_t *pool, *out; uint64_t *d_po, *d_lb, *d_oo, *d_base;
    CK(hipMalloc(&pool, pool_bytes)); CK(hipMalloc(&out, out_bytes));
    CK(hipMemset(pool, 65, pool_bytes));
    CK(hipMalloc(&d_po, pool_off.size() * 8)); CK(hipMalloc(&d_lb, n_lanes * 8)); CK(hipMalloc(&d_oo, n_lanes * 8)); CK(hipMalloc(&d_base, 8));
    CK(hipMemcpy(d_po, pool_off.data(), pool_off.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_lb, lane_bytes.data(), n_lanes * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_oo, out_off.data(), n_lanes * 8, hipMemcpyHostToDevice));
    CK(hipMemset(d_base, 0, 8));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("%d CUs; %.2f GB in, %.2f GB out per launch\n", n_cu, pool_off[n_tiles] / 1e9, out_bytes / 1e9);
    // the other CUs are kept busy by workgroups that take a whole CU each (1024 threads, 120 KB of LDS) and spin on the
    // clock for a fixed time without touching memory -- what a generator launch does to the dispatcher, minus its traffic
    hipStream_t sb, sc;
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&blocker_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int use : {32, 64, 256}) {
        if (use > n_cu) continue;
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            if (use < n_cu) hipLaunchKernelGGL(blocker_kernel, dim3(n_cu - use), dim3(1024), 120 * 1024, sb, 1200000ull);   // 12 ms at 100 MHz
            CK(hipEventRecord(e0, sc));
            hipLaunchKernelGGL(compact_pools_kernel, dim3(n_tiles), dim3(CP_THREADS), 0, sc, pool, d_po, d_lb, d_oo, out, d_base, n_lanes);
            CK(hipEventRecord(e1, sc));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%3d CUs: %.3f ms  (%.1f GB/s in+out, %.1f GB/s per CU each way)\n", use, best, 2 * out_bytes / 1e6 / best, out_bytes / 1e6 / best / use);
        for (int which = 0; which < 3; which++) {
            float b2 = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                if (use < n_cu) hipLaunchKernelGGL(blocker_kernel, dim3(n_cu - use), dim3(1024), 120 * 1024, sb, 1200000ull);
                CK(hipEventRecord(e0, sc));
                if (which == 0) hipLaunchKernelGGL(load_only_kernel, dim3(n_tiles), dim3(256), 0, sc, pool, d_po, (uint32_t)(per_lane / 4), reinterpret_cast<uint32_t*>(d_base));
                else if (which == 1) hipLaunchKernelGGL(store_only_kernel<false>, dim3(n_tiles), dim3(256), 0, sc, out, d_oo, d_lb);
                else hipLaunchKernelGGL(store_only_kernel<true>, dim3(n_tiles), dim3(256), 0, sc, out, d_oo, d_lb);
                CK(hipEventRecord(e1, sc));
                CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < b2) b2 = ms;
            }
            printf("        %s: %.3f ms (%.1f GB/s per CU)\n", which == 0 ? "loads only" : which == 1 ? "stores only" : "stores only, 16-byte aligned", b2, out_bytes / 1e6 / b2 / use);
        }
        fflush(stdout);
    }
    return 0;
}
