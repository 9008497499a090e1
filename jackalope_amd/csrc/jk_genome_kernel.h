// create_genome on the device (create_chromosomes_, /root/reference/src/create_sequences.cpp:59-138).
//
// The reference makes the chromosomes of one OpenMP thread one after the other from one pcg64 engine:
// a gamma draw for the length, then per base AliasSampler::sample = exactly two engine outputs
// (src/alias_sampler.h:53-60).  The per-base cost is fixed, so base j of a chromosome uses outputs
// 2j and 2j+1 after the chromosome's first one, and an LCG can jump: s -> M^k s + (M^k - 1)/(M - 1) c
// (pcg_random.hpp:419-429 `advance`).  The host walks the (short) chain of length draws and jumps
// over each chromosome's bases; the device then fills every run of GENOME_RUN bases independently,
// starting from the chromosome's first state jumped 2 * (first base of the run) steps ahead.
// Bit-identical to the sequential order, parallel over all bases.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "jk_math.h"

namespace jk {

constexpr uint32_t GENOME_RUN = 2048;          // bases per thread (4096 engine outputs)
constexpr uint32_t GENOME_BLOCK = 256;

struct GenomeKernelParams {
    uint8_t* out;                   // ASCII bases
    const uint64_t* chrom_off;      // [n_chroms] byte offset in `out` (64-byte aligned)
    const uint64_t* chrom_len;      // [n_chroms]
    const uint64_t* run_first;      // [n_chroms + 1] index of each chromosome's first run
    const uint64_t* start_state;    // [n_chroms][2] hi, lo: engine state before the first base
    const uint32_t* chrom_lane;     // [n_chroms] which engine (reference thread) makes it
    const uint64_t* lane_inc;       // [n_lanes][2] hi, lo
    const uint64_t* lane_adv;       // [n_lanes][64][4]: 2^k steps as (mult hi, mult lo, plus hi, plus lo)
    uint64_t thresh[4];             // u < Prob[i]  <=>  output < thresh[i]
    uint32_t alias[4];
    uint64_t n_runs;
    uint32_t n_chroms;
};

__global__ void __launch_bounds__(GENOME_BLOCK)
create_genome_kernel(GenomeKernelParams P) {
    __shared__ uint64_t s_thresh[4];
    __shared__ uint32_t s_pick[8];               // [i] = own base character, [4 + i] = alias's
    if (threadIdx.x < 4) {
        s_thresh[threadIdx.x] = P.thresh[threadIdx.x];
        const uint32_t chars = 0x47414354u;      // "TCAG" (jlp::bases, src/jackalope_types.h:36)
        s_pick[threadIdx.x] = (chars >> (8u * threadIdx.x)) & 0xffu;
        s_pick[4 + threadIdx.x] = (chars >> (8u * P.alias[threadIdx.x])) & 0xffu;
    }
    __syncthreads();
    const uint64_t r = (uint64_t)blockIdx.x * GENOME_BLOCK + threadIdx.x;
    if (r >= P.n_runs) return;
    // chromosome of this run: last c with run_first[c] <= r
    uint32_t lo = 0, hi = P.n_chroms;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (P.run_first[mid] <= r) lo = mid; else hi = mid;
    }
    const uint32_t c = lo;
    const uint64_t j0 = (r - P.run_first[c]) * GENOME_RUN;
    const uint64_t len = P.chrom_len[c];
    const uint32_t nb = (uint32_t)((len - j0) < GENOME_RUN ? (len - j0) : GENOME_RUN);
    const uint32_t ln = P.chrom_lane[c];
    jk_pcg64 e;
    e.inc_hi = P.lane_inc[2 * ln]; e.inc_lo = P.lane_inc[2 * ln + 1];
    jk_u128 st = jk_mk128(P.start_state[2 * c], P.start_state[2 * c + 1]);
    {   // jump 2 * j0 outputs ahead: one affine map per set bit of the distance
        const uint64_t* adv = P.lane_adv + (size_t)ln * 64 * 4;
        uint64_t d = 2 * j0;
        while (d) {
            const int k = __builtin_ctzll(d);
            d &= d - 1;
            st = jk_mk128(adv[4 * k], adv[4 * k + 1]) * st + jk_mk128(adv[4 * k + 2], adv[4 * k + 3]);
        }
    }
    e.s_hi = (uint64_t)(st >> 64); e.s_lo = (uint64_t)st;
    uint8_t* dst = P.out + P.chrom_off[c] + j0;
    auto base = [&]() -> uint32_t {
        const uint64_t x1 = jk_pcg_next(e);
        const uint64_t x2 = jk_pcg_next(e);
        const uint32_t i = (uint32_t)((x1 + 1) >> 62) & 3u;      // (uint64)(runif_01 * 4); x1 = 2^64-1 (index 4,
        return s_pick[i + ((x2 < s_thresh[i]) ? 0u : 4u)];        //  out of the table in the reference) is not modelled
    };
    uint32_t k = 0;
    for (; k + 16 <= nb; k += 16) {
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t v = base();
            v |= base() << 8;
            v |= base() << 16;
            v |= base() << 24;
            w[q] = v;
        }
        *reinterpret_cast<uint4*>(dst + k) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    for (; k < nb; k++) dst[k] = (uint8_t)base();
}

}  // namespace jk
