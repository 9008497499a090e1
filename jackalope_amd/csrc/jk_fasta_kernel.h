// FASTA text -> chromosome bytes on the device (read_fasta_noind / read_fasta_ind,
// /root/reference/src/io_fasta.cpp:41-169, :183-408).
//
// The host inflates/reads the file and finds the (few) header lines; the byte-granular work the
// reference does per line or per 4 MiB piece -- dropping '\n' (and one '\r' before it in the
// non-indexed reader, src/str_manip.h:159), mapping every byte through the filter tables
// (src/str_manip.h:24-56: TCAGN stay, tcagn are upper-cased or stay, anything else becomes a zero byte)
// and packing the result -- is a stream compaction over the raw text in HBM:
//   fasta_count_kernel   kept bytes per 4 KiB block
//   (scan of the block counts)
//   fasta_pack_kernel    kept bytes, filtered, to their final positions; records where each
//                        chromosome's interval starts in the output
// A byte is kept when it lies in one of the sorted, disjoint "sequence intervals" of the text
// (one per chromosome: between header lines, or the span an .fai line describes).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace jk {

constexpr uint32_t FASTA_THREADS = 256;
constexpr uint32_t FASTA_PER_THREAD = 16;
constexpr uint32_t FASTA_BLOCK_BYTES = FASTA_THREADS * FASTA_PER_THREAD;

struct FastaParams {
    const uint8_t* text; uint64_t n;         // raw (uncompressed) file bytes; readable up to n + 16
    const uint64_t* iv_begin;                // [n_iv] sorted
    const uint64_t* iv_end;                  // [n_iv]
    uint32_t n_iv;
    int32_t strip_cr;                        // drop a '\r' that directly precedes '\n' (non-indexed reader)
    int32_t upper;                           // remove_soft_mask
    uint64_t* block_cnt;                     // [n_blocks] out of the count kernel
    const uint64_t* block_off;               // [n_blocks] exclusive scan of block_cnt
    uint8_t* out;
    uint64_t* iv_out;                        // [n_iv] output offset at which interval k starts
};

__device__ __forceinline__ uint32_t fasta_filter(uint32_t c, bool upper) {
    const uint32_t u = c & ~0x20u;           // upper-case candidate
    const bool letter = (u == 'T') | (u == 'C') | (u == 'A') | (u == 'G') | (u == 'N');
    return letter ? ((upper || !(c & 0x20u)) ? u : c) : 0u;
}

// first interval whose end is > pos
__device__ __forceinline__ uint32_t fasta_find(const FastaParams& P, uint64_t pos) {
    uint32_t lo = 0, hi = P.n_iv;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (P.iv_end[mid] <= pos) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// keep-mask of the thread's 16 bytes (bit b = byte at `pos + b` is kept); `bytes` receives them
__device__ __forceinline__ uint32_t fasta_keep_mask(const FastaParams& P, uint64_t pos, uint8_t bytes[FASTA_PER_THREAD + 1]) {
    if (pos >= P.n) return 0;
    const uint32_t avail = (uint32_t)((P.n - pos) < FASTA_PER_THREAD ? (P.n - pos) : FASTA_PER_THREAD);
    const uint4 v = *reinterpret_cast<const uint4*>(P.text + pos);         // padded allocation: always readable
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int b = 0; b < (int)FASTA_PER_THREAD; b++) bytes[b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
    bytes[FASTA_PER_THREAD] = (pos + FASTA_PER_THREAD < P.n) ? P.text[pos + FASTA_PER_THREAD] : 0;
    uint32_t k = fasta_find(P, pos);
    uint64_t ib = k < P.n_iv ? P.iv_begin[k] : ~0ULL, ie = k < P.n_iv ? P.iv_end[k] : ~0ULL;
    uint32_t mask = 0;
#pragma unroll
    for (int b = 0; b < (int)FASTA_PER_THREAD; b++) {
        const uint64_t i = pos + b;
        while (i >= ie && k < P.n_iv) {       // (rare) step to the next interval
            k++;
            ib = k < P.n_iv ? P.iv_begin[k] : ~0ULL; ie = k < P.n_iv ? P.iv_end[k] : ~0ULL;
        }
        const uint32_t c = bytes[b];
        bool keep = (uint32_t)b < avail && i >= ib && i < ie && c != '\n';
        if (P.strip_cr && c == '\r' && i + 1 < P.n && bytes[b + 1] == '\n') keep = false;
        mask |= keep ? (1u << b) : 0u;
    }
    return mask;
}

__device__ __forceinline__ uint32_t fasta_block_scan(uint32_t v, uint32_t* total) {
    __shared__ uint32_t wsum[FASTA_THREADS / 64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (uint32_t w = 0; w < FASTA_THREADS / 64; w++) { const uint32_t s = wsum[w]; if (w < wave) before += s; all += s; }
    *total = all;
    return before + incl - v;
}

__global__ void __launch_bounds__(FASTA_THREADS) fasta_count_kernel(FastaParams P) {
    uint8_t bytes[FASTA_PER_THREAD + 1];
    const uint64_t pos = (uint64_t)blockIdx.x * FASTA_BLOCK_BYTES + (uint64_t)threadIdx.x * FASTA_PER_THREAD;
    const uint32_t mask = fasta_keep_mask(P, pos, bytes);
    uint32_t total;
    (void)fasta_block_scan(__popc(mask), &total);
    if (threadIdx.x == 0) P.block_cnt[blockIdx.x] = total;
}

__global__ void __launch_bounds__(FASTA_THREADS) fasta_pack_kernel(FastaParams P) {
    uint8_t bytes[FASTA_PER_THREAD + 1];
    const uint64_t pos = (uint64_t)blockIdx.x * FASTA_BLOCK_BYTES + (uint64_t)threadIdx.x * FASTA_PER_THREAD;
    const uint32_t mask = fasta_keep_mask(P, pos, bytes);
    uint32_t total;
    const uint32_t before = fasta_block_scan(__popc(mask), &total);
    uint64_t at = P.block_off[blockIdx.x] + before;
    // intervals that begin inside this thread's span [pos, pos + 16) -- or, for the thread that owns the
    // last byte, at the very end of the text -- learn their output offset
    {
        const uint64_t span_end = pos + FASTA_PER_THREAD;
        uint32_t lo = 0, hi = P.n_iv;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (P.iv_begin[mid] < pos) lo = mid + 1; else hi = mid; }
        for (uint32_t k = lo; k < P.n_iv && (P.iv_begin[k] < span_end); k++) {
            const uint32_t b = (uint32_t)(P.iv_begin[k] - pos);
            P.iv_out[k] = at + __popc(mask & ((1u << b) - 1u));
        }
    }
#pragma unroll
    for (int b = 0; b < (int)FASTA_PER_THREAD; b++)
        if ((mask >> b) & 1u) P.out[at++] = (uint8_t)fasta_filter(bytes[b], P.upper != 0);
}

}  // namespace jk
