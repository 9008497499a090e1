// Micro-benchmark (not part of the product): issue cost of the VALU instructions the generators are made of, on
// gfx950, at the generator's occupancy (1024-thread workgroups, one per CU, 4 waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_valu tools/ubench_valu.hip
// Each kernel runs `iters` x 32 copies of one instruction, as a dependent chain (DEP=1) or as 4 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(x) x x x x
#define REP32(x) REP4(x) REP4(x) REP4(x) REP4(x) REP4(x) REP4(x) REP4(x) REP4(x)

template <int OP>
__global__ void __launch_bounds__(1024) k(uint64_t* out, int iters, uint32_t m) {
    uint64_t a = threadIdx.x * 0x9e3779b97f4a7c15ULL + 1, b = a * 3, c = b * 5, d = c * 7;
    uint32_t x = threadIdx.x * 2654435761u, y = x * 3u + 1u, z = y * 5u + 7u, w = z * 9u;
    uint64_t sj = 0;
    const uint64_t t0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { REP32(asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(a), "=s"(sj) : "v"(x), "s"(m));) }
        if (OP == 1) { REP32(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "s"(m));) }
        if (OP == 2) { REP32(asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));) }
        if (OP == 3) { REP32(asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));) }
        if (OP == 4) { REP32(asm volatile("v_cmp_lt_u64_e32 vcc, %1, %2\n\tv_cndmask_b32_e32 %0, %0, %3, vcc" : "+v"(x) : "v"(a), "v"(b), "v"(y) : "vcc");) }
        if (OP == 5) { REP32(asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(a) : "v"(x));) }
        if (OP == 6) { REP32(asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(x) : "v"(y));) }
        if (OP == 7) { REP32(asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "s"(m));) }
        if (OP == 8) { REP32(asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));) }
        if (OP == 9) { REP32(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));) }
        if (OP == 10) { REP32(asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));) }
        if (OP == 11) { REP32(asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a) : "v"(b));) }
        if (OP == 12) { REP32(asm volatile("v_addc_co_u32_e64 %0, %1, %0, %2, %1\n\ts_nop 1" : "+v"(x), "+s"(sj) : "v"(y));) }
        if (OP == 13) { REP32(asm volatile("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %4, %1" : "+v"(a), "+v"(b), "=s"(sj) : "v"(x), "s"(m), "v"(y));) }
        if (OP == 14) { REP32(asm volatile("v_cmp_lt_u64_e64 %1, %2, %3\n\tv_xor_b32_e32 %0, %0, %4" : "+v"(x), "=s"(sj) : "v"(a), "v"(b), "v"(y));) }
        if (OP == 15) { REP32(asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x) : "v"(y));) }
        if (OP == 16) { REP32(asm volatile("v_mul_u32_u24_e32 %0, %0, %1" : "+v"(x) : "v"(y));) }
        if (OP == 17) { REP32(asm volatile("v_mul_hi_u32_u24_e32 %0, %0, %1" : "+v"(x) : "v"(y));) }
        if (OP == 18) { REP32(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
        if (OP == 19) { REP32(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (OP == 20) { REP32(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));) }
        if (OP == 21) { REP32(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
        if (OP == 22) { REP32(asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));) }
        if (OP == 23) { REP32(asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));) }
    }
    const uint64_t t1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + x + y + z + w;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[gridDim.x * blockDim.x] = t1 - t0;
}

template <int OP>
static void run(const char* name, int n_inst, uint64_t* d) {
    const int iters = 4000, blocks = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, d, iters, 0x9e3779b9u);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    uint64_t ticks; hipMemcpy(&ticks, d + blocks * 1024, 8, hipMemcpyDeviceToHost);
    // per SIMD: 4 waves x iters x 32 x n_inst instructions in `ticks` x 10 ns
    const double inst = 4.0 * iters * 32 * n_inst;
    printf("%-34s %8.3f ms  %6.2f ns per wave-instruction per SIMD  (= %.2f cycles at 2.4 GHz)\n", name, ms,
           ticks * 10.0 / inst, ticks * 10.0 / inst * 2.4);
}

int main() {
    uint64_t* d; hipMalloc(&d, (256 * 1024 + 8) * 8);
    run<6>("v_xor_b32", 1, d);
    run<0>("v_mad_u64_u32 (dependent)", 1, d);
    run<13>("v_mad_u64_u32 x2 (independent)", 2, d);
    run<1>("v_mul_lo_u32", 1, d);
    run<7>("v_mul_hi_u32", 1, d);
    run<8>("v_mad_u32_u24", 1, d);
    run<16>("v_mul_u32_u24", 1, d);
    run<17>("v_mul_hi_u32_u24", 1, d);
    run<22>("v_mad_i32_i24", 1, d);
    run<15>("v_pk_mul_lo_u16", 1, d);
    run<23>("v_dot4_u32_u8", 1, d);
    run<2>("v_add3_u32", 1, d);
    run<3>("v_alignbit_b32", 1, d);
    run<9>("v_perm_b32", 1, d);
    run<10>("v_bfi_b32", 1, d);
    run<4>("v_cmp_lt_u64 + v_cndmask (vcc)", 2, d);
    run<14>("v_cmp_lt_u64_e64 + v_xor", 2, d);
    run<5>("v_lshlrev_b64", 1, d);
    run<11>("v_lshl_add_u64", 1, d);
    run<12>("v_addc_co_u32 (sgpr carry)", 1, d);
    run<18>("v_fma_f64", 1, d);
    run<19>("v_mul_f64", 1, d);
    run<20>("v_fma_f32", 1, d);
    run<21>("v_pk_fma_f32", 1, d);
    return 0;
}
