// jk_pacbio_kernel.h -- the PacBio (SimLoRD-style) per-read loop as two gfx950 HIP kernels.
//
// Reference: PacBioOneGenome::{one_read,re_read,append_pool} (/root/reference/src/hts_pacbio.cpp:136-485),
// PacBioReadLenSampler::sample (:27-45), PacBioPassSampler::sample (src/hts_pacbio.h:149-205),
// PacBioQualityError::{sample,update_probs,trunc_norm,fill_quals} (src/hts_pacbio.h:266-398,
// src/hts_pacbio.cpp:96-131), PacBioHaplotypes::{one_read,re_read} (src/hts_pacbio.cpp:488-551).
//
// A reference thread ("lane") is one pcg64 stream, and a read of it is three kinds of work:
//   (1) a few dozen data-dependent draws -- read length, chi-square passes, two truncated normals, start, strand;
//   (2) pass 1 (PacBioQualityError::sample): ONE draw per read position, classified against per-read thresholds;
//   (3) pass 2 (append_pool): the bases, one more draw per insertion / substitution, then a constant quality line.
// (2) and (3) are >99 % of the draws and of the bytes, and both are regular: pcg64 is an LCG, so the state k steps
// ahead is A_k * s + G_k * inc with constants that depend on k only (pcg_random.hpp:419-434 `advance`).  Hence
//
//   pb_plan_kernel   one thread per lane for (1), as the reference's thread would do it; for (2) the 64 lanes of
//                    a wave take their streams in turn and work on ONE stream together: lane j holds the state j+1
//                    steps ahead, so one 128-bit multiply-add per lane yields the draws of 64 consecutive positions,
//                    three wave-wide compares turn them into 64-bit event masks (the ballots ARE the masks), and the
//                    read's bookkeeping -- current length, spare chromosome, side of the split -- is scalar
//                    arithmetic on popcounts.  Per read it leaves a record (engine state before the first pass-2
//                    draw, geometry, qualities, exact byte count) and 2 bits per position of event masks; the draws
//                    of pass 2 are jumped over (their number is known from the masks), so the stream can go on to
//                    its next read without pass 2 having happened.
//   pb_emit_kernel   one wave per READ, whatever lane it belongs to (the record carries everything): positions are
//                    lanes -- source bases arrive as one coalesced byte per lane, output offsets are popcount
//                    prefixes of the masks (v_mbcnt), the extra draws of a block's events are 64 consecutive outputs
//                    of the read's stream fetched by rank (ds_bpermute), the quality line is 16-byte stores of a
//                    constant -- and the text goes to its final place in the lane-major FASTQ image: a lane's reads
//                    have exact sizes after the plan kernel, so one scan over the lanes gives every read its
//                    address and neither per-lane pools nor a compaction pass exist on this path.
#pragma once
#include "jk_illumina_kernel.h"
#include "jk_math2.h"

namespace jk {

enum : uint32_t {
    JK_KERR_PB_ALPHA = 4u,        // (unused since chi-square shapes below 1 are implemented)
    JK_KERR_PB_MATH = 8u,         // pow/exp argument outside the transcribed main path
    JK_KERR_PB_TOO_LONG = 16u,    // a read needed more than 2 L + 64 reference positions
    JK_KERR_PB_SPACE = 32u,       // "read_chrom_space should never exceed the chromosome length." / a position outside the reference's read buffer
};

struct PassEntry {                // per integer pass count (host table)
    double sig;                   // sigmoid(passes)
    double sqrtv;                 // sqrt(passes + sqrt_params[0])
    double a_bar;                 // (lower_thresh - mean) / sd
    double p;                     // pnorm(a_bar)            (method 0)
    uint64_t c_m; int32_t c_e;    // 1 - p in x87 extended  (method 0)
    int32_t method;               // 0: inverse-CDF draw, 1: tail rejection (lower_thresh >= mean + 5 sd)
};

// What pb_plan_kernel leaves per read for pb_emit_kernel.
struct PbRead {
    uint64_t s_lo, s_hi;          // engine state before the first draw of append_pool's position walk
    uint64_t out_off;             // byte offset of the record in its lane's text
    uint64_t mask_idx;            // first 16-byte block {plane 0 (64 bits), plane 1} of its event masks
    uint64_t read_start;
    uint32_t L, space, n_pos, split;   // read length, read_chrom_space, positions the walk visits, split_pos (clamped to L)
    uint32_t lane;                // lane of the launch (seed words -> stream increment; lane offset in the image)
    uint32_t ci;                  // chromosome / (haplotype, chromosome) cell
    uint32_t flags;               // bit 0: the read is made, bit 1: reverse strand, bit 2: L + 1 bases; bits 8-15 / 16-23: left / right quality character
    uint32_t stale_idx;           // first of its n_pos - space bytes in the stale buffer (positions past the window, see below)
};
// decimal digits of v (std::to_string(read_start) in the id line)
JK_HD uint32_t jk_dec_digits(uint64_t v) { uint32_t d = 1; while (v >= 10) { v /= 10; d++; } return d; }
constexpr uint32_t PB_HIST = 16;              // depth of the per-lane history of buffer-covering reads (see pb_plan_kernel)
constexpr uint32_t PB_PLAN_BLOCK = 256;       // 4 independent waves
constexpr uint32_t PB_SRC_CHUNK = 1024;       // window bytes pb_emit_kernel brings to LDS at a time (16 blocks of 64 positions)
constexpr uint32_t PB_MASK_CHUNK = 4096;      // 16-byte mask blocks a wave takes from the arena at a time (64 KB)

// pcg64's multiplier to the k-th power and 1 + M + ... + M^(k-1): the state k steps ahead is M^k s + G_k inc
constexpr jk_u128 PB_M = ((jk_u128)JK_PCG_MULT_HI << 64) | JK_PCG_MULT_LO;
constexpr jk_u128 pb_mpow(unsigned k) { jk_u128 r = 1; for (unsigned i = 0; i < k; i++) r *= PB_M; return r; }
constexpr jk_u128 pb_mgeo(unsigned k) { jk_u128 r = 0, p = 1; for (unsigned i = 0; i < k; i++) { r += p; p *= PB_M; } return r; }
constexpr jk_u128 PB_A64 = pb_mpow(64), PB_G64 = pb_mgeo(64);

struct PacbioKernelParams {
    GenomeDev g;
    HapDev h;
    uint32_t hap_seg;             // haplotypes read through the mutation tables (else g.chrom_off is indexed by cell: materialised)
    uint32_t n_lanes;
    uint32_t wave_lanes;          // lanes per wave (a power of two up to 64)
    const uint32_t* seeds;
    const uint64_t* lane_reads;
    const uint32_t* chrom_reads;  // [chrom or cell][lane]
    uint32_t chrom_stride;
    const uint64_t* rec_off;      // [n_lanes] first record of each lane (launch-relative)
    PbRead* recs;                 // zeroed before the launch: a slot no read reaches stays "not made"
    uint64_t* lane_bytes;
    uint64_t* lane_made;
    uint4* masks; uint64_t mask_cap; unsigned long long* mask_ctr;    // arena of 16-byte event-mask blocks, bump-allocated per wave
    uint8_t* stale; uint32_t stale_cap; uint32_t* stale_ctr;          // characters earlier reads left in the reference's `read` buffer
    uint64_t* hist;               // [2 * PB_HIST][n_lanes]
    const uint64_t* jump;         // [64][4]: M^(j+1) lo, hi; G_(j+1) lo, hi
    uint32_t undefined_as_nul;    // JK_PB_UNDEFINED_AS_NUL=1: a position outside the read buffer reads as NUL instead of ending the run
    uint32_t* err;
    // read lengths
    uint32_t use_lognormal;
    double ln_mu, ln_sigma, ln_loc, min_read_len;
    uint32_t n_lens; const uint64_t* len_thresh; const uint32_t* len_alias; const uint64_t* lens;
    // passes
    double cn[3], cs[5];
    double max_passes_d;
    const double* thr_tab; uint32_t thr_cap;      // thr_tab[min(L, thr_cap)] = qchisq(0.9925, n(L))
    // qualities / errors
    const PassEntry* pass_tab;                    // [max_passes + 1]
    double np0, np1, sp1, prob_ins, prob_del, prob_subst;
    uint64_t th_dup; uint32_t dup_all;
    uint64_t pool_size;
};

struct PbEmitParams {
    GenomeDev g;
    HapDev h;
    uint32_t n_chroms;
    const PbRead* recs; uint32_t n_recs;
    const uint32_t* seeds;        // [lanes of the launch][8]
    const uint64_t* lane_off;     // [lanes of the launch] offset of each lane's text in the launch's image
    const uint4* masks; const uint8_t* stale; const uint64_t* jump;
    uint8_t* out; const uint64_t* out_base; uint64_t out_cap;
    uint32_t* err;
    uint32_t exact_from;          // a draw whose scaled high word reaches this takes the exact index routine (0xfffffff0; tests: 0 = all)
};

// Exact integer cut points of the comparisons pass 1 makes against a per-read probability c:
// u(x) = (double)runif_01 is monotone in the raw draw x, so {x : u(x) < c} and {x : u(x) <= c} are prefixes
// [0, t); found by bisection once per read instead of converting every draw to double.
// Returns t, with t = 0 meaning "no x" and *all = true meaning "every x".
template <bool LE>
__device__ __forceinline__ uint64_t cut_point(double c, bool* all) {
    *all = false;
    auto pred = [&](uint64_t x) { const double u = jk_runif_double(x); return LE ? (u <= c) : (u < c); };
    if (!pred(0)) return 0;
    if (pred(~0ULL)) { *all = true; return 0; }
    uint64_t lo = 0, hi = ~0ULL;             // pred(lo) true, pred(hi) false
    while (hi - lo > 1) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (pred(mid)) lo = mid; else hi = mid;
    }
    return hi;
}

// ---- wave helpers ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pb_rl32(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint64_t pb_rl64(uint64_t v, uint32_t l) { return (uint64_t)pb_rl32((uint32_t)v, l) | ((uint64_t)pb_rl32((uint32_t)(v >> 32), l) << 32); }
__device__ __forceinline__ uint32_t pb_lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// number of set bits of `m` below this lane, added to `acc`
__device__ __forceinline__ uint32_t pb_mbcnt(uint64_t m, uint32_t acc) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, acc));
}

// One block of event masks to memory, from the scalar registers where the ballots leave them: gfx950 has the scalar
// stores of the gfx9 family; they go through the scalar data cache, which is written back (s_dcache_wb) before anybody
// reads them.  A vector store needs the four words moved to vector registers and the other lanes masked off first:
// 41.5 against 39.4 M reads/s on BASELINE configs[4] (-DJK_PB_NO_SSTORE for that form).
__device__ __forceinline__ void pb_store_masks(uint4* p, uint64_t plo, uint64_t phi, uint32_t lid) {
#ifndef JK_PB_NO_SSTORE
    typedef uint32_t pb_u4v __attribute__((ext_vector_type(4)));
    const pb_u4v v = {(uint32_t)plo, (uint32_t)(plo >> 32), (uint32_t)phi, (uint32_t)(phi >> 32)};
    asm volatile("s_store_dwordx4 %0, %1, 0x0" :: "s"(v), "s"(p) : "memory");
    (void)lid;
#else
    if (lid == 0) *p = make_uint4((uint32_t)plo, (uint32_t)(plo >> 32), (uint32_t)phi, (uint32_t)(phi >> 32));
#endif
}
// ... and before a lane reads mask blocks back with vector loads: the scalar stores went past this CU's vector L1, which
// may still hold an older copy of a 128-byte line they wrote into (found by the randomised sweep: a lane re-walking its
// masks saw the line as an earlier read had left it), so the L1 is invalidated (agent-scope acquire: buffer_inv sc1).
__device__ __forceinline__ void pb_masks_written() {
#ifndef JK_PB_NO_SSTORE
    asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}
// XSL-RR output of a state given as limbs (the second half of jk_pcg_next)
__device__ __forceinline__ uint64_t pb_pcg_out(uint32_t s0, uint32_t s1, uint32_t s2, uint32_t s3) {
    const uint32_t x_lo = s0 ^ s2, x_hi = s1 ^ s3, rot = s3 >> 26;
    const uint32_t a = __builtin_amdgcn_alignbit(x_hi, x_lo, rot);      // (x >> (rot & 31)) low word
    const uint32_t b = __builtin_amdgcn_alignbit(x_lo, x_hi, rot);      // ... high word
    const bool sw = (s3 >> 31) != 0;                                    // rot >= 32 swaps the words
    return ((uint64_t)(sw ? a : b) << 32) | (sw ? b : a);
}
// s <- s * A64 + c: 64 steps of the engine at once (the multiply-add of jk_pcg_next with another multiplier; c in VGPRs:
// gfx9 VALU instructions read one scalar operand)
__device__ __forceinline__ void pb_pcg_mad64(uint32_t& s0, uint32_t& s1, uint32_t& s2, uint32_t& s3, uint64_t c_lo, uint64_t c_hi) {
    const uint32_t m0 = (uint32_t)PB_A64, m1 = (uint32_t)(PB_A64 >> 32), m2 = (uint32_t)(PB_A64 >> 64), m3 = (uint32_t)(PB_A64 >> 96);
    uint64_t t0, t1, h, cA, cB, junk;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(t0), "=s"(cA) : "v"(s0), "s"(m0), "v"(c_lo));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(h), "=s"(junk) : "v"(s2), "s"(m0), "v"(c_hi));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(t1), "=s"(junk) : "v"(s1), "s"(m0), "v"(t0 >> 32));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(h), "=s"(junk) : "v"(s1), "s"(m1), "v"(h));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(t1), "=s"(cB) : "v"(s0), "s"(m1), "v"(t1));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(h), "=s"(junk) : "v"(s0), "s"(m2), "v"(h));
    const uint32_t hu = (uint32_t)(h >> 32) + (s0 * m3 + s1 * m2 + s2 * m1 + s3 * m0);
    const uint32_t n0 = (uint32_t)t0, n1 = (uint32_t)t1;
    uint32_t h_lo, h_hi;
    // (wait states of the carries as in jk_pcg_next: cA is two VALU instructions old, cB's add comes third)
    asm("v_addc_co_u32_e64 %[hl], vcc, %[h0], %[t1h], %[cA]\n\t"
        "v_mov_b32_e32 %[hh], %[hu]\n\t"
        "v_addc_co_u32_e64 %[hh], %[jk], %[hh], 0, %[cB]\n\t"
        "v_addc_co_u32_e32 %[hh], vcc, 0, %[hh], vcc"
        : [hl] "=&v"(h_lo), [hh] "=&v"(h_hi), [jk] "=&s"(junk)
        : [h0] "v"((uint32_t)h), [t1h] "v"((uint32_t)(t1 >> 32)), [cA] "s"(cA), [hu] "v"(hu), [cB] "s"(cB)
        : "vcc");
    s0 = n0; s1 = n1; s2 = h_lo; s3 = h_hi;
}
// engine::advance (pcg_random.hpp:419-434): the state `delta` steps ahead
__device__ __forceinline__ void pb_pcg_advance(jk_pcg64d& e, uint64_t delta) {
    jk_u128 cur_mult = PB_M, cur_plus = jk_mk128(e.inc_hi, e.inc_lo), acc_mult = 1, acc_plus = 0;
    while (delta > 0) {
        if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    const jk_u128 st = acc_mult * jk_mk128(((uint64_t)e.s3 << 32) | e.s2, ((uint64_t)e.s1 << 32) | e.s0) + acc_plus;
    e.s0 = (uint32_t)st; e.s1 = (uint32_t)(st >> 32); e.s2 = (uint32_t)(st >> 64); e.s3 = (uint32_t)(st >> 96);
}

// character of the reference's `read` buffer -> index into mm_nucleos (nt_map, src/hts.h:36-46)
__device__ __forceinline__ uint32_t pb_nt_of_char(uint32_t ch) { return ch == 'T' ? 0u : ch == 'C' ? 1u : ch == 'A' ? 2u : ch == 'G' ? 3u : 4u; }

// ---------------------------------------------------------------------------------------------------------------
// pb_plan_kernel: everything of a read except its text.
//
// The reference copies a read's `space` source bases into a std::string member that starts as 1000 'N's
// (src/hts_pacbio.h:533) and never shrinks (RefChrom::fill_read, src/ref_classes.h:102-116), then walks read
// positions until the read has its length (src/hts_pacbio.cpp:381-400).  When pass 1 met a deletion it could not
// record (no spare chromosome), or a duplicate lost deletions because it abuts the chromosome end (:277-285), that
// walk differs from pass 1's and can run a few positions past `space`: it then picks up what EARLIER reads of the
// same thread left in the buffer.  Those characters are resolved here, from a per-lane stack of the reads whose
// buffer content is still visible (strictly decreasing `space` from bottom to top: a new read hides every earlier
// one that was not longer; entry = {space | reverse << 32 | cell << 33, read_start}), and handed to pb_emit_kernel
// as bytes.  A position no remembered read covers is 'N' below 1000 if nothing was ever written there, the string's
// terminating NUL if it equals the string's size, and lies outside the string otherwise (undefined in the reference,
// refused here unless undefined_as_nul).
// ---------------------------------------------------------------------------------------------------------------
// Eight waves per SIMD (64 VGPRs; the per-lane set-up phases spill, the shared pass-1 loop needs few): the loop is a
// dependent chain per wave -- multiply-add, output, compares, scalar bookkeeping -- and only other waves fill its gaps,
// and at 64 VGPRs the emit kernel's waves find room on the same SIMDs (measured on four job shapes against four waves of
// 128 VGPRs: +4 to +8 % reads/s, tools/pb_planwaves_probe.sh; 28.9 against 13.8 ms per launch was the step from the 224
// VGPRs the compiler would take to 128).  api_pacbio.h sizes a launch to this number of waves per SIMD.
#ifndef JK_PB_PLAN_WAVES
#define JK_PB_PLAN_WAVES 8
#endif
#define JK_PB_PLAN_ATTR __attribute__((amdgpu_waves_per_eu(JK_PB_PLAN_WAVES, JK_PB_PLAN_WAVES)))
template <bool HAP>
__global__ void __launch_bounds__(PB_PLAN_BLOCK) JK_PB_PLAN_ATTR
pb_plan_kernel(PacbioKernelParams P) {
    // A wave carries P.wave_lanes (1..64) lanes of the launch, in its first threads: the per-lane phases are a small part of
    // the work and pass 1 uses all 64 threads whatever the number of streams, so a launch of few lanes (few reference
    // threads with many reads each) still fills the device with waves.
    const uint32_t lid = pb_lane_id();
    const uint32_t lane = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * P.wave_lanes + lid;
    const bool valid = lid < P.wave_lanes && lane < P.n_lanes;
    const uint32_t lane_c = valid ? lane : 0;

    LaneRng rng;
    rng.e = jk_pcg_limbs(jk_pcg_seed(P.seeds + (size_t)lane_c * 8));
    jk_gamma_state ln_st; ln_st.saved = 0.0; ln_st.saved_available = 0; ln_st.fail = 0;     // lognormal_distribution::_M_nd
    jk_gamma_state chi_st; chi_st.saved = 0.0; chi_st.saved_available = 0; chi_st.fail = 0;  // chi_squared -> gamma -> _M_nd

    // (a lane's read count fits 32 bits: plan_lanes refuses more)
    const uint32_t quota = valid ? (uint32_t)P.lane_reads[lane] : 0;
    uint32_t made = 0, in_pool = 0;
    const uint32_t pool_size = P.pool_size > 0xffffffffULL ? 0xffffffffu : (uint32_t)P.pool_size;
    const uint32_t n_cells = HAP ? P.h.n_haps * P.g.n_chroms : P.g.n_chroms;
    uint32_t ci = 0;
    uint32_t ccnt = (n_cells && valid) ? P.chrom_reads[lane] : 0;
    uint32_t cur_hap = 0xffffffffu;
    const uint64_t rec0 = valid ? P.rec_off[lane] : 0;
    uint64_t* const hl = P.hist + lane_c;
    const size_t hstride = (size_t)P.n_lanes;
    uint64_t out_pos = 0;             // bytes of this lane's text so far

    // this lane's jump constants: the state lid + 1 steps ahead of s is Aj s + Gj inc
    const jk_u128 Aj = jk_mk128(P.jump[4 * lid + 1], P.jump[4 * lid]);
    const jk_u128 Gj = jk_mk128(P.jump[4 * lid + 3], P.jump[4 * lid + 2]);

    // the wave's piece of the event-mask arena (wave-uniform)
    uint64_t mk_ptr = 0, mk_end = 0;

    uint32_t err = 0;
    uint64_t L = 0, read_start = 0, chrom_len = 0;
    bool is_dup = false;
    uint32_t hdepth = 0;
    uint64_t max_written = 0;         // longest window any read of this buffer had so far
    for (;;) {
        const bool alive = valid && made < quota && err == 0;
        if (__builtin_amdgcn_ballot_w64(alive) == 0) break;

        // ---------------- per lane: cell, read length, passes, error probabilities ----------------
        bool part_b = false;
        uint64_t tL0 = 0, tL1 = 0, tL2 = 0, tR0 = 0, tR1 = 0, tR2 = 0;
        uint32_t fLR = 0, split32 = 0, extra0 = 0, quals = 0;
        if (alive) do {
            if (!is_dup) {
                // chromosome / cell.  Reference genome: first chromosome with a non-zero quota; the quota is never
                // decremented on this path (src/hts_pacbio.cpp:145-151).  Haplotypes: cursor search (:498-521).
                while (ci < n_cells && ccnt == 0) { ci++; if (ci < n_cells) ccnt = P.chrom_reads[(size_t)ci * P.chrom_stride + lane]; }
                if (ci >= n_cells) { made = quota; break; }
                if (HAP) {
                    const uint32_t hap = ci / P.g.n_chroms;
                    if (hap != cur_hap) {      // a new PacBioOneHaplotype: fresh distribution objects
                        cur_hap = hap;
                        ln_st.saved = 0.0; ln_st.saved_available = 0;
                        chi_st.saved = 0.0; chi_st.saved_available = 0;
                        hdepth = 0; max_written = 0;            // ... and its own read buffer
                    }
                }
                chrom_len = HAP ? P.h.cell_size[ci] : P.g.chrom_len[ci];
                // ---- read length (src/hts_pacbio.cpp:27-45)
                if (P.use_lognormal) {
                    double rnd = 0;
                    uint32_t iters = 0;
                    for (;;) {
                        double ex;
                        if (!jk_exp(P.ln_sigma * jk_normal(ln_st, rng) + P.ln_mu, &ex)) { err |= JK_KERR_PB_MATH; ex = 0; }
                        rnd = ex + P.ln_loc;
                        if (!(rnd < P.min_read_len) || iters >= 10) break;
                        iters++;
                    }
                    if (rnd < P.min_read_len) rnd = P.min_read_len;
                    L = rnd >= 0x1p63 ? ~0ULL : (uint64_t)rnd;
                } else {
                    const uint32_t i0 = (uint32_t)jk_runif_index(rng(), P.n_lens);
                    const uint64_t x2 = rng();
                    L = P.lens[(x2 < P.len_thresh[i0]) ? i0 : P.len_alias[i0]];
                }
                if (L >= chrom_len) L = chrom_len;
            }
            if (err) break;
            if (L > 0x3fffffffULL) { err |= JK_KERR_PB_TOO_LONG; break; }     // 32-bit position counters below
            // ---- number of passes (src/hts_pacbio.h:149-205)
            const double Ld = (double)L;
            double n = P.cn[0] * (Ld < P.cn[2] ? Ld : P.cn[2]) + P.cn[1];
            if (n < 0.001) n = 0.001;
            double sc;
            if (Ld <= P.cs[2]) { sc = P.cs[0] * Ld - P.cs[1]; if (sc < 0.001) sc = 0.001; }
            else { bool ok = true; sc = P.cs[3] / jk_pow(Ld, P.cs[4], &ok); if (!ok) err |= JK_KERR_PB_MATH; }
            // chi_squared_distribution(n) = 2 * gamma(n / 2, 1) (random.h: _M_gd(__n / 2), operator() returns 2 * _M_gd(urng));
            // n / 2 < 1 takes the gamma sampler's pow branch
            const jk_gamma_param gp = jk_gamma_make(n / 2, 1.0);
            double passes = 2 * jk_gamma(gp, chi_st, rng);
            const double thr = P.thr_tab[L < P.thr_cap ? (uint32_t)L : P.thr_cap];
            while (passes > thr) passes = 2 * jk_gamma(gp, chi_st, rng);
            if (chi_st.fail) { err |= JK_KERR_PB_MATH; break; }
            passes *= sc;
            passes += 1;
            if (passes > P.max_passes_d) passes = P.max_passes_d;
            const double wholes = __builtin_trunc(passes), fraction = passes - wholes;
            double passes_left, passes_right, prop_left;
            if ((((uint64_t)wholes) & 1ULL) == 0ULL) { prop_left = fraction; passes_left = __builtin_ceil(passes); passes_right = __builtin_floor(passes); }
            else { prop_left = 1 - fraction; passes_left = __builtin_floor(passes); passes_right = __builtin_ceil(passes); }
            const uint64_t split_pos = (uint64_t)__builtin_round(Ld * prop_left);

            // ---- per-side error probabilities and qualities (update_probs, trunc_norm, fill_quals)
            double cumL[3], cumR[3];
            uint32_t qual_left = '!', qual_right = '!';
#pragma unroll
            for (int side = 0; side < 2; side++) {
                const PassEntry pe = P.pass_tab[(uint32_t)(side == 0 ? passes_left : passes_right)];
                double rnd;
                if (pe.method == 0) {
                    jk_x87 c; c.m = pe.c_m; c.e = pe.c_e;
                    const double u = jk_runif_ab(rng(), jk_x87_from_double(pe.p), c);
                    rnd = jk_qnorm(u) * P.np1 + P.np0;
                } else {
                    double u = jk_runif_double(rng());
                    double x_bar = jk_sqrt(pe.a_bar * pe.a_bar - 2 * jk_log(1 - u));
                    double v = jk_runif_double(rng());
                    while (v > (x_bar / pe.a_bar)) {
                        u = jk_runif_double(rng());
                        x_bar = jk_sqrt(pe.a_bar * pe.a_bar - 2 * jk_log(1 - u));
                        v = jk_runif_double(rng());
                    }
                    rnd = P.np1 * x_bar + P.np0;
                }
                (side == 0 ? cumL : cumR)[0] = rnd;      // parked; the exponents need both draws first
            }
#pragma unroll
            for (int side = 0; side < 2; side++) {
                double* cum = side == 0 ? cumL : cumR;
                const PassEntry pe = P.pass_tab[(uint32_t)(side == 0 ? passes_left : passes_right)];
                double expo = cum[0] * pe.sig + pe.sqrtv - P.sp1;
                if (expo < 0.6) expo = 0.6;
                bool ok = true;
                cum[0] = jk_pow(P.prob_ins, expo, &ok);
                cum[1] = jk_pow(P.prob_del, expo, &ok) + cum[0];
                cum[2] = jk_pow(P.prob_subst, expo, &ok) + cum[1];
                if (!ok) err |= JK_KERR_PB_MATH;
                const double qv = __builtin_round(-10.0 * jk_log10(cum[2]));
                // (uint64) of a negative or infinite value is undefined in the reference (src/hts_pacbio.h: `uint64 tmp =
                // std::round(...)`); x86-64 makes a huge number of a negative one -> 93, and 0 of +inf (all three
                // probabilities exactly 0: cvttsd2si of inf - 2^63 gives 2^63, whose top bit the conversion flips back)
                const uint32_t q = qv < 0 ? 93u : (qv > 1.8e19 ? 0u : (qv > 93.0 ? 93u : (uint32_t)qv));
                (side == 0 ? qual_left : qual_right) = q + 33u;
            }
            if (err) break;
            // u > cum[2]  <=>  x >= t_none ; u < cum[0]  <=>  x < t_ins ; u < cum[1]  <=>  x < t_del
            bool aL[3], aR[3];
            tL0 = cut_point<true>(cumL[2], &aL[0]); tL1 = cut_point<false>(cumL[0], &aL[1]); tL2 = cut_point<false>(cumL[1], &aL[2]);
            tR0 = cut_point<true>(cumR[2], &aR[0]); tR1 = cut_point<false>(cumR[0], &aR[1]); tR2 = cut_point<false>(cumR[1], &aR[2]);
            // bit 0: never "none", bit 1: always insertion, bit 2: always deletion (cut points that cover every draw); right side << 8
            fLR = (aL[0] ? 1u : 0u) | (aL[1] ? 2u : 0u) | (aL[2] ? 4u : 0u) | (aR[0] ? 0x100u : 0u) | (aR[1] ? 0x200u : 0u) | (aR[2] ? 0x400u : 0u);
            split32 = split_pos > L ? 0xffffffffu : (uint32_t)split_pos;     // (a split beyond L is never reached)
            const uint64_t spare = chrom_len - L;
            extra0 = spare > 0x7fffffffULL ? 0x7fffffffu : (uint32_t)spare;   // (it only matters when it reaches 0, at most one step per position)
            quals = (qual_left << 8) | (qual_right << 16);
            part_b = true;
        } while (0);

        // ---------------- the wave together: pass 1 of every lane's read, one stream at a time ----------------
        // (PacBioQualityError::sample, src/hts_pacbio.h:292-317: one draw per position)
        uint32_t r_pos = 0, r_nins = 0, r_ndel = 0, r_nsub = 0, r_fl = 0;
        uint64_t r_mk = 0;
        uint64_t todo = __builtin_amdgcn_ballot_w64(part_b);
        while (todo) {
            const uint32_t s = (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1;
            // the stream's engine and this read's constants, wave-uniform from here on
            const uint32_t b0 = pb_rl32(rng.e.s0, s), b1 = pb_rl32(rng.e.s1, s), b2 = pb_rl32(rng.e.s2, s), b3 = pb_rl32(rng.e.s3, s);
            const jk_u128 S = jk_mk128(((uint64_t)b3 << 32) | b2, ((uint64_t)b1 << 32) | b0);
            const jk_u128 I = jk_mk128(pb_rl64(rng.e.inc_hi, s), pb_rl64(rng.e.inc_lo, s));
            uint64_t tn = pb_rl64(tL0, s), ti = pb_rl64(tL1, s), td = pb_rl64(tL2, s);
            const uint64_t tnR = pb_rl64(tR0, s), tiR = pb_rl64(tR1, s), tdR = pb_rl64(tR2, s);
            const uint32_t fboth = pb_rl32(fLR, s);
            uint32_t f = fboth & 0xffu;
            const uint32_t L32 = pb_rl32((uint32_t)L, s), sp32 = pb_rl32(split32, s);
            uint32_t extra = pb_rl32(extra0, s);
            const uint32_t max_pos = 2u * L32 + 64u;                // the walk may take this many positions (L <= 2^30)
            // (a read's blocks start on a 64-byte line of their own: the scalar data cache writes lines back, and a line that a
            // lane patches below -- a duplicate losing deletions -- must not be written again by a later read's scalar stores)
            const uint32_t nb_max = (((max_pos + 63u) >> 6) + 3u) & ~3u;
            uint32_t fl = 0;
            if (mk_ptr + nb_max > mk_end) {                         // the wave's piece of the arena is used up: take another
                const uint64_t want = nb_max > PB_MASK_CHUNK ? nb_max : PB_MASK_CHUNK;
                unsigned long long got = 0;
                if (lid == 0) got = atomicAdd(P.mask_ctr, (unsigned long long)want);
                got = pb_rl64(got, 0);
                if (got + want > P.mask_cap) fl |= 4u;              // arena exhausted: the host retries with a larger one
                else { mk_ptr = got; mk_end = got + want; }
            }
            const uint64_t mk0 = mk_ptr;
            uint32_t cur = 0, upos = 0, n_ins = 0, n_del = 0, n_sub = 0;
            uint32_t e0 = b0, e1 = b1, e2 = b2, e3 = b3;            // the stream's state after pass 1
            if (L32 > 0 && !(fl & 4u)) {
                const jk_u128 st = Aj * S + Gj * I;                 // lane j: the state j + 1 steps ahead
                uint32_t s0 = (uint32_t)st, s1 = (uint32_t)(st >> 32), s2 = (uint32_t)(st >> 64), s3 = (uint32_t)(st >> 96);
                const jk_u128 C64 = PB_G64 * I;
                uint64_t c_lo = (uint64_t)C64, c_hi = (uint64_t)(C64 >> 64);
                asm volatile("" : "+v"(c_lo), "+v"(c_hi));          // (kept in vector registers: the addend of v_mad_u64_u32)
                const uint32_t Lm1 = L32 - 1u;
                bool on_right = false;
                for (;;) {
                    // Blocks that need no looking at: while the current length is more than 128 short of the split (on the
                    // left side) and of the last base, 64 spare bases remain per block and the walk's cap is far, a block is
                    // three compares and three popcounts (the length grows by at most 128 per block, the spare chromosome
                    // shrinks by at most 64).  The careful block below takes over wherever that ends.
                    if (f == 0u) {
                        uint32_t nsafe = (Lm1 - cur) >> 7;
                        if (!on_right) { const uint32_t b_ = sp32 >= cur ? (sp32 - cur) >> 7 : 0u; nsafe = nsafe < b_ ? nsafe : b_; }
                        { const uint32_t c_ = extra >> 6, d_ = (max_pos - upos) >> 6; nsafe = nsafe < c_ ? nsafe : c_; nsafe = nsafe < d_ ? nsafe : d_; }
                        if (nsafe) {
                            uint32_t si = 0, sd = 0, ss = 0;
                            uint4* mp = P.masks + mk0 + (upos >> 6);
                            for (uint32_t q = 0; q < nsafe; q++) {
                                const uint64_t xq = pb_pcg_out(s0, s1, s2, s3);
                                const uint64_t qn = __builtin_amdgcn_ballot_w64(xq >= tn), qi = __builtin_amdgcn_ballot_w64(xq < ti), qd = __builtin_amdgcn_ballot_w64(xq < td);
                                const uint64_t ins = ~qn & qi, nd_ = ~qn & ~qi, del = nd_ & qd, sub = nd_ & ~qd;
                                si += (uint32_t)__builtin_popcountll(ins); sd += (uint32_t)__builtin_popcountll(del); ss += (uint32_t)__builtin_popcountll(sub);
                                const uint64_t plo_ = ins | sub, phi_ = nd_;          // (deletion or substitution: neither none nor insertion)
                                pb_store_masks(mp + q, plo_, phi_, lid);
                                pb_pcg_mad64(s0, s1, s2, s3, c_lo, c_hi);
                            }
                            cur += 64u * nsafe + si - sd; extra += si - sd;
                            n_ins += si; n_del += sd; n_sub += ss; upos += 64u * nsafe;
                        }
                    }
                    if (upos >= max_pos) break;                          // (the cap is reached with the read unfinished: refused below)
                    const uint64_t x = pb_pcg_out(s0, s1, s2, s3);
                    // same decision tree as the reference (src/hts_pacbio.h:296-314) on 64 positions at once
                    uint64_t bn = (f & 1u) ? 0ULL : __builtin_amdgcn_ballot_w64(x >= tn);
                    uint64_t bi = (f & 2u) ? ~0ULL : __builtin_amdgcn_ballot_w64(x < ti);
                    uint64_t bd = (f & 4u) ? ~0ULL : __builtin_amdgcn_ballot_w64(x < td);
                    uint64_t plo, phi;
                    uint32_t k;
                    if (extra >= 64u && upos + 64u <= max_pos) {
                        // Every deletion of this block finds spare chromosome, so positions depend on each other through
                        // the current length only: it decides the side of the split (the thresholds change when it reaches
                        // split_pos), whether an insertion is recorded (not at the last base) and where the read ends --
                        // and before position i it is cur + i + #insertions - #deletions below i, a popcount prefix.
                        uint64_t ins = ~bn & bi, del = ~bn & ~bi & bd, sub = ~bn & ~bi & ~bd;
                        if (!on_right && cur + 128u > sp32) {
                            const uint32_t len_i = cur + lid + pb_mbcnt(ins, 0u) - pb_mbcnt(del, 0u);
                            const uint64_t m = __builtin_amdgcn_ballot_w64(len_i >= sp32);   // (exact up to its first lane: all below are left)
                            if (m != 0) {
                                on_right = true;
                                tn = tnR; ti = tiR; td = tdR; f = fboth >> 8;
                                bn = (f & 1u) ? 0ULL : __builtin_amdgcn_ballot_w64(x >= tn);
                                bi = (f & 2u) ? ~0ULL : __builtin_amdgcn_ballot_w64(x < ti);
                                bd = (f & 4u) ? ~0ULL : __builtin_amdgcn_ballot_w64(x < td);
                                const uint64_t low = (m & (0ULL - m)) - 1ULL;                 // the positions still on the left
                                ins = (ins & low) | (~bn & bi & ~low);
                                del = (del & low) | (~bn & ~bi & bd & ~low);
                                sub = (sub & low) | (~bn & ~bi & ~bd & ~low);
                            }
                        }
                        k = 64u;
                        if (cur + 128u > Lm1) {
                            const uint32_t len_i = cur + lid + pb_mbcnt(ins, 0u) - pb_mbcnt(del, 0u);
                            const uint64_t pm = __builtin_amdgcn_ballot_w64(len_i < L32);        // positions the loop reaches: a prefix
                            // an insertion at the last base is not recorded (it can only be the last position reached)
                            const uint64_t unrec = ins & __builtin_amdgcn_ballot_w64(len_i == Lm1);
                            ins &= pm & ~unrec; del &= pm; sub &= pm;
                            k = (uint32_t)__builtin_popcountll(pm);
                        }
                        const uint32_t ni = (uint32_t)__builtin_popcountll(ins), nd = (uint32_t)__builtin_popcountll(del);
                        cur += k + ni - nd; extra += ni - nd;
                        n_ins += ni; n_del += nd; n_sub += (uint32_t)__builtin_popcountll(sub);
                        plo = ins | sub; phi = del | sub;
                        upos += k;
                    } else {
                        // the chromosome is nearly used up (a deletion is only recorded while spare bases remain), or the
                        // walk is about to exceed its cap: position by position
                        plo = 0; phi = 0; k = 0;
                        while (k < 64u && cur < L32 && upos < max_pos) {
                            if (!on_right && cur >= sp32) {      // (the reference switches sides when the length reaches split_pos)
                                on_right = true;
                                tn = tnR; ti = tiR; td = tdR; f = fboth >> 8;
                                bn = (f & 1u) ? 0ULL : __builtin_amdgcn_ballot_w64(x >= tn);
                                bi = (f & 2u) ? ~0ULL : __builtin_amdgcn_ballot_w64(x < ti);
                                bd = (f & 4u) ? ~0ULL : __builtin_amdgcn_ballot_w64(x < td);
                            }
                            const uint64_t bit = 1ULL << k;
                            const bool none = (bn & bit) != 0;
                            const bool ins = !none && (bi & bit) != 0;
                            const bool del = !none && !ins && (bd & bit) != 0;
                            const bool sub = !none && !ins && !del;
                            const bool ins_rec = ins && cur < Lm1;              // an insertion at the last base is not recorded
                            const bool del_rec = del && extra > 0u;             // nor a deletion without spare chromosome
                            if (del && !del_rec) fl |= 1u;                      // the walk of append_pool will differ from this one
                            extra = extra + (ins_rec ? 1u : 0u) - (del_rec ? 1u : 0u);
                            cur += (ins_rec ? 1u : 0u) + (del ? 0u : 1u);
                            n_ins += ins_rec ? 1u : 0u; n_del += del_rec ? 1u : 0u; n_sub += sub ? 1u : 0u;
                            plo |= (ins_rec || sub) ? bit : 0ULL;        // code bit 0: insertion (1) or substitution (3)
                            phi |= (del_rec || sub) ? bit : 0ULL;        // code bit 1: deletion (2) or substitution (3)
                            k++; upos++;
                        }
                    }
                    pb_store_masks(P.masks + mk0 + ((upos - k) >> 6), plo, phi, lid);
                    if (cur >= L32 || upos >= max_pos) {                 // k >= 1 here: the stream stands behind its k-th draw of this block
                        e0 = pb_rl32(s0, k - 1u); e1 = pb_rl32(s1, k - 1u); e2 = pb_rl32(s2, k - 1u); e3 = pb_rl32(s3, k - 1u);
                        break;
                    }
                    pb_pcg_mad64(s0, s1, s2, s3, c_lo, c_hi);
                }
                if (cur < L32) fl |= 2u;
                mk_ptr = mk0 + ((((upos + 63u) >> 6) + 3u) & ~3ULL);
            }
            if (lid == s) {                                         // back to the lane that owns the stream
                rng.e.s0 = e0; rng.e.s1 = e1; rng.e.s2 = e2; rng.e.s3 = e3;
                r_pos = upos; r_nins = n_ins; r_ndel = n_del; r_nsub = n_sub; r_fl = fl; r_mk = mk0;
            }
        }
        // (the mask blocks were stored by lane 0 / from scalar registers; a lane may read its own below)
        pb_masks_written();
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");

        // ---------------- per lane again: start, strand, the record ----------------
        if (part_b) do {
            if (r_fl & 4u) { err |= JK_KERR_POOL_OVERFLOW; break; }
            if (r_fl & 2u) { err |= JK_KERR_PB_TOO_LONG; break; }
            const uint32_t pos = r_pos;
            uint32_t n_del = r_ndel;
            bool differs = (r_fl & 1u) != 0;          // append_pool's walk is not pass 1's
            uint64_t space = L + n_del - r_nins;
            bool give_up = false;
            if (!is_dup) {
                if (space < chrom_len) read_start = jk_frag_start(rng(), chrom_len - space + 1);
                else if (space == chrom_len) read_start = 0;
                else { err |= JK_KERR_PB_SPACE; break; }
            } else {
                // duplicate: drop deletions from the back until the read fits (src/hts_pacbio.cpp:277-285)
                uint64_t scan = pos;      // positions [0, pos) hold events
                while (space + read_start > chrom_len && n_del > 0) {
                    while (scan > 0) {
                        scan--;
                        uint4* wp = P.masks + r_mk + (scan >> 6);
                        const uint32_t bit = 1u << (scan & 31u);         // deletion = code 2: plane 1 set, plane 0 clear
                        uint4 wv = *wp;
                        const uint32_t p0 = (scan & 32u) ? wv.y : wv.x, p1 = (scan & 32u) ? wv.w : wv.z;
                        if ((p1 & bit) && !(p0 & bit)) {
                            if (scan & 32u) wv.w &= ~bit; else wv.z &= ~bit;
                            *wp = wv;
                            break;
                        }
                    }
                    n_del--; space--;
                    differs = true;
                }
                if (space + read_start > chrom_len) give_up = true;
            }
            if (give_up) break;
            // ---- append_pool (src/hts_pacbio.cpp:350-414): strand, then the walk's draws are jumped over
            const bool reverse = jk_runif_lt_half(rng());
            uint32_t n_walk = pos, n_draws = r_nins + r_nsub, over = 0;
            if (differs) {
                // how far the walk gets, and how many insertions / substitutions it meets: position k is visited while
                // k + #ins(< k) - #del(< k) < L
                uint64_t cur2 = 0;
                uint32_t p = 0;
                n_draws = 0;
                while (cur2 < L && p < pos) {
                    const uint4 wv = P.masks[r_mk + (p >> 6)];
                    const uint64_t lo = (uint64_t)wv.x | ((uint64_t)wv.y << 32), hi = (uint64_t)wv.z | ((uint64_t)wv.w << 32);
                    const uint64_t insm = lo & ~hi, delm = hi & ~lo;
                    const uint64_t need = L - cur2;
                    const uint64_t f64 = 64u + (uint64_t)__builtin_popcountll(insm) - (uint64_t)__builtin_popcountll(delm);
                    if (f64 < need) { cur2 += f64; n_draws += (uint32_t)__builtin_popcountll(lo); p += 64u; continue; }
                    uint32_t a = 0;                    // largest k in [0, 63] with f(k) < need (f(0) = 0)
#pragma unroll
                    for (uint32_t step = 32; step > 0; step >>= 1) {
                        const uint32_t k = a + step;
                        if (k <= 63u) {
                            const uint64_t below = (1ULL << k) - 1ULL;
                            const uint64_t fk = (uint64_t)k + (uint64_t)__builtin_popcountll(insm & below) - (uint64_t)__builtin_popcountll(delm & below);
                            a = (fk < need) ? k : a;
                        }
                    }
                    const uint32_t kcut = a + 1u;      // positions < kcut of this block are visited
                    const uint64_t procm = kcut >= 64u ? ~0ULL : ((1ULL << kcut) - 1ULL);
                    n_draws += (uint32_t)__builtin_popcountll(lo & procm);
                    // an insertion that pass 1 recorded may be the walk's last position: the read then has L + 1 bases
                    // (src/hts_pacbio.cpp:384-388 adds two at once; the quality line keeps L characters)
                    over = (uint32_t)((uint64_t)kcut + (uint64_t)__builtin_popcountll(insm & procm) - (uint64_t)__builtin_popcountll(delm & procm) - need);
                    p += kcut; cur2 = L;
                }
                if (cur2 < L || p > pos) { err |= JK_KERR_PB_SPACE; break; }      // (cannot happen: the walk never needs more positions than pass 1)
                n_walk = p;
            }
            const uint32_t hdr_len = P.g.hdr_off[ci + 1] - P.g.hdr_off[ci];
            const uint64_t out_len = (uint64_t)hdr_len + jk_dec_digits(read_start) + 3u + L + over + 3u + L + 1u;
            PbRead R;
            R.s_lo = ((uint64_t)rng.e.s1 << 32) | rng.e.s0; R.s_hi = ((uint64_t)rng.e.s3 << 32) | rng.e.s2;
            R.out_off = out_pos; R.mask_idx = r_mk; R.read_start = read_start;
            R.L = (uint32_t)L; R.space = (uint32_t)space; R.n_pos = n_walk; R.split = split32 < (uint32_t)L ? split32 : (uint32_t)L;
            R.lane = lane; R.ci = ci;
            R.flags = 1u | (reverse ? 2u : 0u) | (over ? 4u : 0u) | (quals << 0);
            R.stale_idx = 0;
            if ((uint64_t)n_walk > space) {
                // positions past this read's window: the characters earlier reads left in the buffer
                const uint32_t cnt = (uint32_t)(n_walk - space);
                const uint32_t at = atomicAdd(P.stale_ctr, cnt);
                if ((uint64_t)at + cnt > P.stale_cap) { err |= JK_KERR_POOL_OVERFLOW; break; }
                R.stale_idx = at;
                for (uint32_t t = 0; t < cnt; t++) {
                    const uint64_t q = space + t;
                    uint32_t d = hdepth;
                    uint64_t h0 = 0, h1 = 0;
                    bool found = false;
                    while (d > 0) {                      // newest first: the first one long enough wrote position q last
                        d--;
                        h0 = hl[(2 * d) * hstride]; h1 = hl[(2 * d + 1) * hstride];
                        if ((h0 & 0xffffffffULL) > q) { found = true; break; }
                    }
                    uint32_t ch;
                    if (found) {
                        const uint64_t sp_j = h0 & 0xffffffffULL;
                        const bool rev_j = (h0 >> 32) & 1ULL;
                        const uint32_t cell_j = (uint32_t)(h0 >> 33);
                        const uint64_t hpos = rev_j ? (h1 + sp_j - 1 - q) : (h1 + q);
                        uint64_t addr;
                        if (HAP && P.hap_seg) {
                            const uint64_t coff = P.g.chrom_off[cell_j % P.g.n_chroms];
                            int64_t m = hap_search(P.h, cell_j, hpos);
                            addr = hap_resolve(P.h, coff, cell_j, m, hpos).addr;
                        } else addr = P.g.chrom_off[cell_j] + hpos;
                        const uint32_t c0 = P.g.seq[addr];
                        const bool is_nt = c0 < 4u;
                        ch = is_nt ? base_char(rev_j ? (c0 ^ 2u) : c0) : (rev_j ? (c0 == 'N' ? (uint32_t)'N' : 0u) : jk_decode_other(c0));
                    } else if (q < max_written) {
                        err |= JK_KERR_PB_SPACE; ch = 0;     // written by a read the stack no longer remembers
                    } else if (q < 1000u) ch = 'N';          // never written: the string's initial content
                    else if (q == (space > max_written ? space : max_written) || P.undefined_as_nul) ch = 0;   // read[size()]: the terminator
                    else { err |= JK_KERR_PB_SPACE; ch = 0; }
                    P.stale[at + t] = (uint8_t)ch;
                }
                if (err) break;
            }
            P.recs[rec0 + made] = R;
            out_pos += out_len;
            pb_pcg_advance(rng.e, n_draws);
            // this read's window now sits in the buffer: it hides every remembered read that was not longer
            if (space > max_written) max_written = space;
            while (hdepth > 0 && (hl[(2 * (hdepth - 1)) * hstride] & 0xffffffffULL) <= space) hdepth--;
            if (hdepth == PB_HIST) {                 // forget the oldest (longest) one; a position only it covered is refused
                for (uint32_t d = 1; d < PB_HIST; d++) {
                    hl[(2 * d - 2) * hstride] = hl[(2 * d) * hstride];
                    hl[(2 * d - 1) * hstride] = hl[(2 * d + 1) * hstride];
                }
                hdepth--;
            }
            hl[(2 * hdepth) * hstride] = (space & 0xffffffffULL) | ((uint64_t)(reverse ? 1u : 0u) << 32) | ((uint64_t)ci << 33);
            hl[(2 * hdepth + 1) * hstride] = read_start;
            hdepth++;
        } while (0);
        if (part_b && err == 0) {
            if (HAP) ccnt = ccnt > 0 ? ccnt - 1 : 0;     // n_reads_vc[hap][chr]-- (one_read) / if > 0 (re_read)
            // ---- ReadWriterOneThread::create_reads tail (src/hts.h:263-278), one read end
            made += 1; in_pool += 1;
            const uint64_t xd = rng();
            const bool dup = P.dup_all || xd < P.th_dup;
            if (dup && made < quota && in_pool < pool_size) is_dup = true;
            else { is_dup = false; if (in_pool >= pool_size || made >= quota) in_pool = 0; }
        }
    }
    if (valid) {
        P.lane_bytes[lane] = out_pos;
        P.lane_made[lane] = made;
    }
    if (err) atomicOr(P.err, err);
}

// ---------------------------------------------------------------------------------------------------------------
// pb_emit_kernel: the text of one read per wave (append_pool, src/hts_pacbio.cpp:350-414 / :417-485).
//
// Memory side: a wave-wide byte access costs the texture addresser as much as a wave-wide 16-byte access (the first
// version -- one byte load and two byte stores per 64 positions -- was bound by exactly that: 70 clocks per block and
// CU).  So the source window arrives as ONE dword per lane per 256 positions and is dealt to the position lanes through
// the LDS crossbar (ds_bpermute), and the text is laid down byte-wise in a 2 KB LDS ring indexed by the low bits of its
// global address and leaves as whole 1 KB-aligned segments, 16 bytes per lane.
// SEG: the bases of a haplotype come through the mutation tables, one lookup per position (the fallback when the
// haplotypes do not fit in device memory materialised).
// ---------------------------------------------------------------------------------------------------------------
constexpr uint32_t PB_RING = 2048;
struct PbMaskBlock { uint32_t x, y, z, w; };
typedef const PbMaskBlock __attribute__((address_space(4)))* pb_cmask_t;      // (constant address space: uniform loads become s_load)

template <bool SEG>
__global__ void __launch_bounds__(64)
pb_emit_kernel(PbEmitParams P) {
    __shared__ __align__(16) uint8_t ring[PB_RING];
    __shared__ __align__(16) uint8_t srcb[2 * PB_SRC_CHUNK];     // the window's bytes, two chunks of 1 KB in turn (see stage_chunk)
    const uint32_t r = blockIdx.x;
    if (r >= P.n_recs) return;
    const PbRead R = P.recs[r];
    if (!(R.flags & 1u)) return;
    if (*P.err) return;                           // the plan kernel failed: records may be incomplete
    const uint32_t lid = pb_lane_id();
    const uint32_t L = R.L, space = R.space;
    const bool reverse = (R.flags & 2u) != 0;
    const uint32_t h0 = P.g.hdr_off[R.ci], hlen = P.g.hdr_off[R.ci + 1] - h0;
    const uint32_t nd = jk_dec_digits(R.read_start);
    const uint64_t out_len = (uint64_t)hlen + nd + 3u + L + ((R.flags >> 2) & 1u) + 3u + L + 1u;
    const uint64_t at = P.out_base[0] + P.lane_off[R.lane] + R.out_off;
    // the image is allocated for the expected size: never write past it
    if (at + out_len > P.out_cap) { if (lid == 0) atomicOr(P.err, JK_KERR_IMAGE_FULL); return; }

    // ---- the text writer: g = the next byte, `flushed` = the first byte still in the ring, both as offsets from `gbase`,
    // the ring-aligned address below the record's first byte (32 bits: a record is shorter than 2^32 bytes; the kernel's
    // bookkeeping is scalar work, and 64-bit scalar arithmetic is two instructions per operation)
    const uint64_t gaddr0 = (uint64_t)(uintptr_t)(P.out + at);
    const uint64_t gbase = gaddr0 & ~(uint64_t)(PB_RING - 1u);
    uint32_t g = (uint32_t)(gaddr0 - gbase), flushed = g;
    auto put = [&](uint32_t off, uint32_t byte) { ring[(g + off) & (PB_RING - 1u)] = (uint8_t)byte; };
    // write out the ring's bytes [flushed, upto) of the 1 KB-aligned segment that holds `flushed` (upto <= its end)
    auto flush_segment = [&](uint32_t upto) {
        __syncthreads();
        const uint32_t seg = flushed & ~1023u;
        if (flushed == seg && upto == seg + 1024u) {         // a whole segment (all but the first and last of a read)
            const uint4 v = *reinterpret_cast<const uint4*>(ring + (seg & (PB_RING - 1u)) + 16u * lid);
            reinterpret_cast<uint4*>((uintptr_t)(gbase + seg))[lid] = v;
        } else {
            const uint32_t pa = seg + 16u * lid;                 // this lane's 16-byte piece
            const uint32_t plo = pa > flushed ? pa : flushed, phi = pa + 16u < upto ? pa + 16u : upto;
            if (plo < phi) {
                if (phi - plo == 16u) {
                    const uint4 v = *reinterpret_cast<const uint4*>(ring + (pa & (PB_RING - 1u)));
                    *reinterpret_cast<uint4*>((uintptr_t)(gbase + pa)) = v;
                } else {
                    for (uint32_t a = plo; a < phi; a++) *reinterpret_cast<uint8_t*>((uintptr_t)(gbase + a)) = ring[a & (PB_RING - 1u)];
                }
            }
        }
        flushed = upto;
        __syncthreads();
    };
    auto flush_full = [&]() { while (g - (flushed & ~1023u) >= 1024u) flush_segment((flushed & ~1023u) + 1024u); };

    // ---- id line: "@<genome>-<chromosome>-" + start + "-F\n" / "-R\n"
    for (uint32_t i0 = 0; i0 < hlen; i0 += 64u) {
        if (i0 + lid < hlen) put(lid, P.g.hdr_blob[h0 + i0 + lid]);
        g += (hlen - i0 < 64u ? hlen - i0 : 64u);
        flush_full();
    }
    if (lid < nd) {
        uint32_t dg;
        if (R.read_start <= 0xffffffffULL) {      // (64-bit division is a subroutine of hundreds of instructions)
            uint32_t p10 = 1;
            for (uint32_t j = lid + 1u; j < nd; j++) p10 *= 10u;
            dg = ((uint32_t)R.read_start / p10) % 10u;
        } else {
            uint64_t p10 = 1;
            for (uint32_t j = lid + 1u; j < nd; j++) p10 *= 10u;
            dg = (uint32_t)((R.read_start / p10) % 10u);
        }
        put(lid, '0' + dg);
    }
    if (lid == 61u) put(nd, '-');
    if (lid == 62u) put(nd + 1u, reverse ? 'R' : 'F');
    if (lid == 63u) put(nd + 2u, '\n');
    g += nd + 3u;
    flush_full();

    // ---- bases: 64 positions of the walk per step, one per lane
    const uint32_t* sw = P.seeds + (size_t)R.lane * 8;
    const jk_u128 I = (jk_mk128(((uint64_t)sw[4] << 32) + sw[5], ((uint64_t)sw[6] << 32) + sw[7]) << 1) | 1;     // the stream's increment
    // The walk's draws (one per insertion / substitution, in position order): lane j keeps the engine state j + 1 steps
    // behind the last buffer of 64 draws, so the next buffer is one multiply-add by M^64 per lane.  A draw is only ever
    // used as (uint64)(runif_01 * 3) or (uint64)(runif_01 * 4): both indices are made for all 64 draws at once and
    // travel as one packed word; the (2^-28) draws whose index depends on their low word mark the whole buffer "exact".
    uint32_t ds0, ds1, ds2, ds3;
    {
        const jk_u128 st = jk_mk128(P.jump[4 * lid + 1], P.jump[4 * lid]) * jk_mk128(R.s_hi, R.s_lo) + jk_mk128(P.jump[4 * lid + 3], P.jump[4 * lid + 2]) * I;
        ds0 = (uint32_t)st; ds1 = (uint32_t)(st >> 32); ds2 = (uint32_t)(st >> 64); ds3 = (uint32_t)(st >> 96);
    }
    uint64_t dc_lo, dc_hi;
    { const jk_u128 C64 = PB_G64 * I; dc_lo = (uint64_t)C64; dc_hi = (uint64_t)(C64 >> 64); asm volatile("" : "+v"(dc_lo), "+v"(dc_hi)); }
    // Two buffers of 64 draws are kept, A (draws 0..63 of the window) and B (64..127): lane j holds draw j of each as raw
    // words and as the packed word {index of 3, index of 4 << 8}.  `used` counts the draws of A handed out (< 64 whenever a
    // block starts), so a block -- at most 64 draws -- finds its draws in A and B without refilling; when A is used up, B
    // becomes A and a new B is made (one multiply-add by M^64 per lane).
    uint32_t xa_lo = 0, xa_hi = 0, pka = 0, xb_lo = 0, xb_hi = 0, pkb = 0;
    bool xa_exact = false, xb_exact = false;      // some draw of the buffer needs runif_index32
    auto make_buffer = [&](uint32_t& x_lo, uint32_t& x_hi, uint32_t& pk, bool& ex) {
        const uint64_t xv = pb_pcg_out(ds0, ds1, ds2, ds3);
        x_lo = (uint32_t)xv; x_hi = (uint32_t)(xv >> 32);
        const uint32_t p3 = x_hi * 3u;
        pk = __umulhi(x_hi, 3u) | (__builtin_amdgcn_perm(0u, 0x47414354u, x_hi >> 30) << 8);      // {index of 3, "TCAG"[index of 4] << 8}
        ex = __builtin_amdgcn_ballot_w64(p3 >= P.exact_from || (x_hi << 2) >= P.exact_from) != 0;
    };
    make_buffer(xa_lo, xa_hi, pka, xa_exact);
    pb_pcg_mad64(ds0, ds1, ds2, ds3, dc_lo, dc_hi);
    make_buffer(xb_lo, xb_hi, pkb, xb_exact);
    uint32_t used = 0;                            // draws of A already handed out
    auto next_buffer = [&]() {
        xa_lo = xb_lo; xa_hi = xb_hi; pka = pkb; xa_exact = xb_exact;
        pb_pcg_mad64(ds0, ds1, ds2, ds3, dc_lo, dc_hi);
        make_buffer(xb_lo, xb_hi, pkb, xb_exact);
        used -= 64u;
    };
    const uint64_t coff = P.g.chrom_off[SEG ? R.ci % P.n_chroms : R.ci];
    const uint8_t* const gseq = P.g.seq;
    // forward: position p of the window is the byte at A + p; reverse: the complement of the one at A - p
    const uint64_t A = coff + (reverse ? R.read_start + space - 1u : R.read_start);
    const uint32_t rcm = reverse ? 2u : 0u;
    const pb_cmask_t cmasks = (pb_cmask_t)(uintptr_t)(P.masks + R.mask_idx);
    uint32_t cur = 0;                             // bases written so far
    const uint32_t nblk = (R.n_pos + 63u) >> 6;
    // The window's bytes come through LDS, 1 KB (16 blocks) at a time: every lane loads 16 bytes of the next chunk while the
    // current one is worked on, and they are put down at the end of the current chunk.  gfx950 counts vector loads and stores
    // with one counter and lets them complete out of order, so waiting for ANY load waits for every store issued before the
    // wait -- with a load per group of four blocks the wave stood still for a store's whole latency forty times per 10-kb
    // read (9.3 ms per launch, of which 6 were instruction issue); now it does so once per sixteen blocks.
    // Position p of the window is byte (p & 2047) ^ rx of `srcb` (reverse strand: a lane's 16 bytes lie in ascending address
    // order, i.e. in descending position order: rx = 15).
    const uint32_t rx = reverse ? 15u : 0u;
    auto load_chunk = [&](uint32_t c) -> uint4 {
        uint4 w = make_uint4(0u, 0u, 0u, 0u);
        const uint32_t p0 = c * PB_SRC_CHUNK + 16u * lid;
        if (!SEG && p0 < space) {
            const uint8_t* a = reverse ? gseq + (A - p0 - 15u) : gseq + (A + p0);
            __builtin_memcpy(&w, a, 16);
        }
        return w;
    };
    auto stage_chunk = [&](uint32_t c, const uint4& w) { *reinterpret_cast<uint4*>(srcb + (c & 1u) * PB_SRC_CHUNK + 16u * lid) = w; };
    const uint32_t n_chunks = SEG ? 0u : (space + PB_SRC_CHUNK - 1u) / PB_SRC_CHUNK;
    if (n_chunks > 0u) stage_chunk(0u, load_chunk(0u));
    uint4 nxt = n_chunks > 1u ? load_chunk(1u) : make_uint4(0u, 0u, 0u, 0u);
    const uint32_t src_lane = (uint32_t)(uintptr_t)srcb + (lid ^ rx);         // LDS address of this lane's byte of block 0
    // One block of 64 positions.  SAFE: the caller knows that the walk visits all 64, that they lie inside the window and
    // that this is not the read's last block -- true for all but the last few blocks of a read, and it takes the
    // end-of-read, end-of-window and flush tests (scalar instructions: a SIMD issues one per four clocks, like vector
    // ones, and this kernel has as many of them) out of the block.
    auto do_block = [&](uint32_t b, uint32_t mvx, uint32_t mvy, uint32_t mvz, uint32_t mvw, auto safe_tag, auto a_only_tag) {
        constexpr bool SAFE = decltype(safe_tag)::value;
        constexpr bool A_ONLY = decltype(a_only_tag)::value;        // (SAFE) all draws of the group lie in buffer A
        const uint64_t lo = (uint64_t)mvx | ((uint64_t)mvy << 32), hi = (uint64_t)mvz | ((uint64_t)mvw << 32);
        const uint64_t insm = lo & ~hi, delm = hi & ~lo, keep = ~delm;
        const uint32_t off = pb_mbcnt(insm, pb_mbcnt(keep, 0u));          // bases the positions below this lane's add
        // positions the walk visits: a prefix of the block
        const uint64_t pm = SAFE ? ~0ULL : __builtin_amdgcn_ballot_w64(off < L - cur);
        const uint64_t kp = keep & pm, ip = insm & pm, sp = lo & hi & pm;
        const uint64_t evm = lo & pm;                                     // insertions and substitutions among them: one draw each
        // ---- this block's draws, in position order (src/hts_pacbio.cpp:384-395): consecutive outputs of the stream by rank
        // substitution: mm_nucleos[nt][(uint64)(runif_01 * 3)]; insertion: jlp::bases[(uint64)(runif_01 * 4)]
        // (a SAFE block belongs to a group whose draws were counted beforehand: they all lie in A and B, and neither buffer
        //  holds a draw that needs the exact routine -- no decision is left in the block)
        uint64_t nulm = 0;                                                // draws that index past their string: its NUL
        const uint32_t ne = (uint32_t)__builtin_popcountll(evm);
        const uint32_t d = used + pb_mbcnt(evm, 0u);                      // this lane's draw, if it has one: 0..127
        const int baddr = (int)((d << 2) & 0xfcu);                        // (the same for draw d of A and draw d - 64 of B)
        uint32_t pkv;
        if (A_ONLY) pkv = (uint32_t)__builtin_amdgcn_ds_bpermute(baddr, (int)pka);
        else {
            const uint32_t pa = (uint32_t)__builtin_amdgcn_ds_bpermute(baddr, (int)pka), pb = (uint32_t)__builtin_amdgcn_ds_bpermute(baddr, (int)pkb);
            pkv = d < 64u ? pa : pb;
        }
        if (!SAFE && ne != 0u && (xa_exact || xb_exact)) {
            asm volatile("" ::: "memory");
            const uint32_t h1 = (uint32_t)__builtin_amdgcn_ds_bpermute(baddr, (int)xa_hi), l1 = (uint32_t)__builtin_amdgcn_ds_bpermute(baddr, (int)xa_lo);
            const uint32_t h2 = (uint32_t)__builtin_amdgcn_ds_bpermute(baddr, (int)xb_hi), l2 = (uint32_t)__builtin_amdgcn_ds_bpermute(baddr, (int)xb_lo);
            const uint32_t xh = d < 64u ? h1 : h2, xl = d < 64u ? l1 : l2;
            const bool is_sub = __builtin_amdgcn_inverse_ballot_w64(sp);
            const uint32_t nidx = is_sub ? 3u : 4u;
            const uint32_t code = runif_index32(((uint64_t)xh << 32) | xl, nidx);
            nulm = __builtin_amdgcn_ballot_w64(code >= nidx) & evm;
            pkv = (code & 3u) | (base_char(code & 3u) << 8);
        }
        used += ne;
        // ---- source base
        const uint32_t p = b * 64u + lid;
        uint32_t craw = 0;
        if (SEG) {
            if (__builtin_amdgcn_inverse_ballot_w64(pm) && p < space) {
                const uint64_t hpos = reverse ? (R.read_start + space - 1u - p) : (R.read_start + p);
                int64_t m = hap_search(P.h, R.ci, hpos);
                craw = gseq[hap_resolve(P.h, coff, R.ci, m, hpos).addr];
            }
        } else {
            craw = *reinterpret_cast<const __attribute__((address_space(3))) uint8_t*>((uintptr_t)(src_lane + ((b * 64u) & (2u * PB_SRC_CHUNK - 1u))));
        }
        const uint64_t outm = (SAFE || b * 64u + 64u <= space) ? 0ULL : __builtin_amdgcn_ballot_w64(p >= space);     // positions past the window
        // (SAFE, reference / materialised haplotypes: the group's 256 source bytes were tested at once)
        const uint64_t oddm = (SAFE && !SEG) ? 0ULL : (__builtin_amdgcn_ballot_w64(craw >= 4u) | outm | nulm) & pm;
        const uint32_t c = craw ^ rcm;                                    // (complemented on the reverse strand, if it is a base)
        const bool my_sub = __builtin_amdgcn_inverse_ballot_w64(sp);
        uint32_t ch, ich;
        if (oddm == 0) {
            // every visited position is T, C, A or G inside the window (bytes above the lowest of a result are not stored)
            const uint32_t c3 = pkv & 0xffu;
            const uint32_t sc = c3 + (c3 >= c ? 1u : 0u);
            ch = __builtin_amdgcn_perm(0u, 0x47414354u, my_sub ? sc : c);
            ich = pkv >> 8;
        } else {
            // characters as the reference's `read` buffer holds them (cmp_map on the reverse strand keeps N and zeroes
            // everything else that is not a base; positions past the window hold what earlier reads left)
            const bool proc = __builtin_amdgcn_inverse_ballot_w64(pm), is_nul = __builtin_amdgcn_inverse_ballot_w64(nulm);
            const uint32_t c3 = pkv & 3u;
            uint32_t bch, nt;
            if (p >= space) { bch = proc ? P.stale[R.stale_idx + (p - space)] : 0u; nt = pb_nt_of_char(bch); }
            else if (craw < 4u) { bch = base_char(c); nt = c; }
            else { bch = reverse ? (craw == 'N' ? (uint32_t)'N' : 0u) : jk_decode_other(craw); nt = 4u; }
            const uint32_t sub_ch = is_nul ? 0u : (nt < 4u ? base_char(c3 + (c3 >= nt ? 1u : 0u)) : (uint32_t)'N');
            ch = my_sub ? sub_ch : bch;
            ich = is_nul ? 0u : (pkv >> 8);
        }
        const uint32_t ri = (g + off) & (PB_RING - 1u);
        if (__builtin_amdgcn_inverse_ballot_w64(kp)) ring[ri] = (uint8_t)ch;
        if (__builtin_amdgcn_inverse_ballot_w64(ip)) ring[(ri + 1u) & (PB_RING - 1u)] = (uint8_t)ich;
        const uint32_t nb = (uint32_t)__builtin_popcountll(kp) + (uint32_t)__builtin_popcountll(ip);
        cur += nb; g += nb;
        if (!SAFE && used >= 64u) next_buffer();
    };
    for (uint32_t b0 = 0; b0 < nblk && cur < L; b0 += 4u) {
        // a group of four blocks inside the window that does not reach the end of the read
        bool fast = cur + 256u < L && (b0 + 4u) * 64u <= space && b0 + 4u < nblk;
        typedef uint32_t pb_u16v __attribute__((ext_vector_type(16)));
        pb_u16v mg = {};
        uint32_t nd4 = 0;                              // draws of the group
        if (fast) {
            // (the four blocks' masks as one 64-byte scalar load: a read's blocks start on a 64-byte line)
            mg = *reinterpret_cast<const pb_u16v __attribute__((address_space(4)))*>(cmasks + b0);
            // ... and the group's draws fit the two buffers, none of which needs the exact routine, and its 256 source bytes
            // are plain bases (lane l holds four of them)
            nd4 = (uint32_t)__builtin_popcountll((uint64_t)mg[0] | ((uint64_t)mg[1] << 32)) + (uint32_t)__builtin_popcountll((uint64_t)mg[4] | ((uint64_t)mg[5] << 32)) +
                                 (uint32_t)__builtin_popcountll((uint64_t)mg[8] | ((uint64_t)mg[9] << 32)) + (uint32_t)__builtin_popcountll((uint64_t)mg[12] | ((uint64_t)mg[13] << 32));
            bool plain = true;
            if (!SEG) {
                const uint32_t w4 = *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>((uintptr_t)((uint32_t)(uintptr_t)srcb + ((b0 * 64u) & (2u * PB_SRC_CHUNK - 1u)) + 4u * lid));
                plain = __builtin_amdgcn_ballot_w64((w4 & 0xfcfcfcfcu) != 0u) == 0;
            }
            // (bases the group adds: a position gives one unless it is deleted, and one more if it carries an insertion)
            uint32_t nb4 = 0;
#pragma unroll
            for (uint32_t q = 0; q < 4u; q++) {
                const uint64_t lo_ = (uint64_t)mg[4 * q] | ((uint64_t)mg[4 * q + 1] << 32), hi_ = (uint64_t)mg[4 * q + 2] | ((uint64_t)mg[4 * q + 3] << 32);
                nb4 += (uint32_t)__builtin_popcountll(~(hi_ & ~lo_)) + (uint32_t)__builtin_popcountll(lo_ & ~hi_);
            }
            fast = cur + nb4 < L && used + nd4 <= 128u && !xa_exact && !xb_exact && plain;
        }
        if (fast && used + nd4 <= 64u) {
#pragma unroll
            for (uint32_t q = 0; q < 4u; q++) do_block(b0 + q, mg[4 * q], mg[4 * q + 1], mg[4 * q + 2], mg[4 * q + 3], std::true_type(), std::true_type());
            flush_full();
            if (used >= 64u) next_buffer();
        } else if (fast) {
#pragma unroll
            for (uint32_t q = 0; q < 4u; q++) do_block(b0 + q, mg[4 * q], mg[4 * q + 1], mg[4 * q + 2], mg[4 * q + 3], std::true_type(), std::false_type());
            flush_full();                      // (at most 1023 + 512 bytes are pending here: the ring holds 2048)
            if (used >= 64u) next_buffer();    // (used <= 128 here)
            if (used >= 64u) next_buffer();
        } else {
#pragma unroll
            for (uint32_t q = 0; q < 4u; q++) {
                if (b0 + q >= nblk || cur >= L) break;
                do_block(b0 + q, cmasks[b0 + q].x, cmasks[b0 + q].y, cmasks[b0 + q].z, cmasks[b0 + q].w, std::false_type(), std::false_type());
                flush_full();
            }
        }
        // the last group of a chunk: the next chunk goes down beside it (into the half the chunk before this one used), and
        // the one after that is requested
        if (!SEG && (b0 & 15u) == 12u) {
            const uint32_t c = (b0 >> 4) + 1u;
            if (c < n_chunks) {
                stage_chunk(c, nxt);
                if (c + 1u < n_chunks) nxt = load_chunk(c + 1u);
            }
        }
    }
    // ---- "\n+\n": the rest of the ring leaves, the quality line goes straight to the image
    if (lid == 0) put(0u, '\n');
    if (lid == 1) put(1u, '+');
    if (lid == 2) put(2u, '\n');
    g += 3u;
    flush_full();
    if (g > flushed) flush_segment(g);
    uint8_t* const ql = reinterpret_cast<uint8_t*>((uintptr_t)(gbase + g));
    if (lid == 0) ql[L] = '\n';
    const uint32_t q_left = (R.flags >> 8) & 0xffu, q_right = (R.flags >> 16) & 0xffu, split = R.split;
    // 16-byte pieces at 16-byte aligned addresses; the pieces at the ends and the one across the split byte by byte
    const uint32_t head = (uint32_t)((16u - ((uintptr_t)ql & 15u)) & 15u);      // bytes before the first aligned piece
    if (lid < head && lid < L) ql[lid] = (uint8_t)(lid < split ? q_left : q_right);
    if (L > head) {
        const uint32_t body = L - head, n16 = body >> 4;
        for (uint32_t k = lid; k < n16; k += 64u) {
            const uint32_t o = head + k * 16u;                                   // bytes [o, o + 16) of the line
            if (o + 16u <= split || o >= split) {
                const uint32_t w = (o >= split ? q_right : q_left) * 0x01010101u;
                *reinterpret_cast<uint4*>(ql + o) = make_uint4(w, w, w, w);
            } else {
                for (uint32_t t = 0; t < 16u; t++) ql[o + t] = (uint8_t)(o + t < split ? q_left : q_right);
            }
        }
        const uint32_t tail0 = head + n16 * 16u;
        if (tail0 + lid < L) ql[tail0 + lid] = (uint8_t)(tail0 + lid < split ? q_left : q_right);
    }
}

}  // namespace jk
