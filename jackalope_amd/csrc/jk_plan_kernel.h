// jk_plan_kernel.h -- the chromosome-level reads_per_group calls of a run on the device.
//
// A task = one reads_per_group(n, chromosome sizes of a haplotype) call of the reference's set-up (src/hts.h:58-103): a
// fresh pcg64 from its 8 seed words, then per chromosome a binomial of the reads that are left.  With ~10^2 reads per
// haplotype and lane spread over two dozen chromosomes every one of these binomials has t*p < 8 and takes libstdc++'s
// waiting-time branch (random.tcc:1494-1519): a sum of -log(1 - u) / (t - x) -- one pcg64 step and one glibc log (jk_log,
// bit-equal to libm) per term.  The kernel writes the quotas where the generator reads them; a task that meets a larger
// binomial (libstdc++'s rejection algorithm: lgamma, a stateful normal deviate) is flagged and redone by the host.
#pragma once
#include "jk_math.h"

namespace jk {

struct SplitChainDev {            // GroupChain of jk_host.h for every haplotype, flattened: [hap][G - 1]
    const double* p;              // probability of group g's binomial
    const double* q;              // -log(1 - min(p, 1 - p)), computed by the host's libm
    const uint8_t* kind;          // 0 = draw, 1 = skipped, 2 = takes all remaining reads
    uint32_t G;                   // groups (chromosomes)
};

__global__ void __launch_bounds__(256)
chrom_split_kernel(uint64_t n_tasks, const uint32_t* __restrict__ task_words, const uint32_t* __restrict__ task_n,
                   const uint32_t* __restrict__ task_lane, const uint32_t* __restrict__ task_hap, SplitChainDev C,
                   uint32_t mult, uint64_t stride, uint32_t* __restrict__ quotas, uint32_t* __restrict__ redo) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_tasks) return;
    uint64_t n = task_n[k];
    const uint32_t G = C.G;
    if (G == 0 || n == 0) return;
    const uint32_t hap = task_hap[k];
    uint32_t* out = quotas + (uint64_t)hap * G * stride + task_lane[k];      // group g at out[g * stride]
    jk_pcg64 e = jk_pcg_seed(task_words + k * 8);
    const double* P = C.p + (size_t)hap * (G - 1);
    const double* Q = C.q + (size_t)hap * (G - 1);
    const uint8_t* K = C.kind + (size_t)hap * (G - 1);
    for (uint32_t g = 0; g + 1 < G; g++) {
        if (K[g] == 2) { out[(uint64_t)g * stride] = (uint32_t)(n * mult); return; }
        if (K[g] == 1) continue;
        const double p = P[g];
        const double p12 = p <= 0.5 ? p : 1.0 - p;
        if ((double)n * p12 >= 8) { redo[k] = 1u; return; }       // libstdc++'s other branch: the host redoes this task
        const double q = Q[g];
        uint64_t x = 0;
        double sum = 0.0;
        do {
            if (n == x) { x++; break; }
            const double ee = -jk_log(1.0 - jk_canonical(jk_pcg_next(e)));
            sum += ee / (double)(n - x);
            x += 1;
        } while (sum <= q);
        uint64_t got = x - 1;
        if (p12 != p) got = n - got;
        if (got) out[(uint64_t)g * stride] = (uint32_t)(got * mult);
        n -= got;
        if (n == 0) break;
    }
    if (n) out[(uint64_t)(G - 1) * stride] = (uint32_t)(n * mult);
}

// the quotas of the tasks the host redid: vals[i][g] -> quotas of task idx[i]
__global__ void __launch_bounds__(256)
chrom_split_patch_kernel(uint64_t n, const uint64_t* __restrict__ idx, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ task_lane,
                         const uint32_t* __restrict__ task_hap, uint32_t G, uint64_t stride, uint32_t* __restrict__ quotas) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * G) return;
    const uint64_t t = i / G, g = i % G, k = idx[t];
    quotas[((uint64_t)task_hap[k] * G + g) * stride + task_lane[k]] = vals[i];
}

}  // namespace jk
