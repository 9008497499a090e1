// jk_pacbio_kernel.h -- the PacBio (SimLoRD-style) per-read loop as a gfx950 HIP kernel.
//
// Reference: PacBioOneGenome::{one_read,re_read,append_pool} (/root/reference/src/hts_pacbio.cpp:136-485),
// PacBioReadLenSampler::sample (:27-45), PacBioPassSampler::sample (src/hts_pacbio.h:149-205),
// PacBioQualityError::{sample,update_probs,trunc_norm,fill_quals} (src/hts_pacbio.h:266-398,
// src/hts_pacbio.cpp:96-131), PacBioHaplotypes::{one_read,re_read} (src/hts_pacbio.cpp:488-551).
//
// Same execution model as the Illumina kernel: one GPU thread = one lane = one reference thread.  Unlike
// Illumina reads, PacBio reads differ in length by thousands of bases, so the lanes of a wave drift far
// apart in their output position and the word-interleaved pool tiles would turn every 4-byte store into
// its own 32-byte memory transaction (measured: WRITE_SIZE 8.4x the FASTQ bytes).  Here each lane owns a
// contiguous pool region and stages its text in LDS, 128 bytes per lane, leaving as whole 128-byte lines.  Per read the reference makes two
// passes over the read's positions: (1) one draw per position classifies it as plain / insertion /
// deletion / substitution, (2) the bases are emitted with one more draw per insertion or
// substitution.  Pass 1 stores 2 bits per position (as two 32-bit planes of one u64 per 32 positions) in a
// per-lane HBM scratch laid out [word][lane] (lanes advance in lock-step so the stores coalesce); pass 2
// reads them back.  Everything that only depends on an integer (pass count, read length <= chi2_n[2])
// comes from host-built tables, so the device needs exp/pow/log10/qnorm (jk_math2.h) but no nmath.
#pragma once
#include "jk_illumina_kernel.h"
#include "jk_math2.h"

namespace jk {

enum : uint32_t {
    JK_KERR_PB_ALPHA = 4u,        // chi-square shape n/2 < 1 (needs pow in the gamma sampler)
    JK_KERR_PB_MATH = 8u,         // pow/exp argument outside the transcribed main path
    JK_KERR_PB_TOO_LONG = 16u,    // a read needed more positions than the event scratch holds
    JK_KERR_PB_SPACE = 32u,       // "read_chrom_space should never exceed the chromosome length."
};

struct PassEntry {                // per integer pass count (host table)
    double sig;                   // sigmoid(passes)
    double sqrtv;                 // sqrt(passes + sqrt_params[0])
    double a_bar;                 // (lower_thresh - mean) / sd
    double p;                     // pnorm(a_bar)            (method 0)
    uint64_t c_m; int32_t c_e;    // 1 - p in x87 extended  (method 0)
    int32_t method;               // 0: inverse-CDF draw, 1: tail rejection (lower_thresh >= mean + 5 sd)
};

struct PacbioKernelParams {
    GenomeDev g;
    HapDev h;
    uint32_t n_lanes;
    const uint32_t* seeds;
    const uint64_t* lane_reads;
    const uint32_t* chrom_reads;  // [chrom or cell][lane]
    uint32_t chrom_stride;
    const uint64_t* pool_off;     // [n_tiles + 1]
    uint8_t* pool;
    uint64_t* lane_bytes;
    uint64_t* lane_made;
    uint64_t* ev;                 // [ev_words][n_lanes] 2-bit event codes
    uint32_t ev_words;
    uint32_t* xchg;               // [PB_XCHG_WORDS][n_lanes rounded up to the workgroup]: lane states on their way between threads
    uint64_t* hist;               // [2 * PB_HIST][n_lanes]: what earlier reads left in the reference's `read` buffer
    uint32_t undefined_as_nul;    // JK_PB_UNDEFINED_AS_NUL=1: a position outside that buffer reads as NUL instead of ending the run
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

// ---------------------------------------------------------------------------------------------
// Staged appender: each lane owns PB_STAGE bytes of LDS in which its text is laid down byte by byte, and whole
// 16-byte pieces leave for the lane's contiguous pool region (the rest moves to the front).  Only the owning
// thread touches its bytes.
//
// In the per-position loop the lanes of a wave fill their stages at nearly the same rate (about one byte per
// position), so flushing is done by the whole wave at once: when ANY lane is nearly full, EVERY lane writes out
// all of its complete pieces (ls_flush under a wave-uniform branch): a lane flushing on its own whenever its
// private line fills puts a ~50-instruction divergent block into ~40 % of all loop iterations.
// Byte-granular staging is what lets the word path of pass 2 write "this position's base, then the inserted base"
// unconditionally and merely advance the write offset by 0, 1 or 2 (a deleted or absent byte is overwritten by the
// next one), with no shifting or masking in registers.
// ---------------------------------------------------------------------------------------------
#ifndef JK_PB_BLOCK
#define JK_PB_BLOCK 256
#endif
constexpr int PB_BLOCK = JK_PB_BLOCK;  // lanes per workgroup = the pool the lanes of a wave are regrouped from, read by read
constexpr uint32_t PB_XCHG_WORDS = 48; // 32-bit words of lane state that move with a lane (see the regrouping in pacbio_kernel)
constexpr size_t PB_LDS_BYTES = (size_t)144 * PB_BLOCK + (size_t)4 * PB_BLOCK;   // stages + sort keys
constexpr uint32_t PB_HIST = 16;      // depth of the per-lane history of buffer-covering reads (see pacbio_kernel)
constexpr uint32_t PB_STAGE = 144;    // 128 + what one step of the word path can add beyond its flush threshold
constexpr uint32_t PB_FLUSH_AT = 112; // wave-synchronised flush threshold of the per-position loops
struct LinStream {
    uint8_t* gp;       // global address of the byte staged at lds[0] (16-byte aligned)
    uint8_t* lds;      // this lane's stage
    uint32_t off;      // bytes staged
    uint64_t pos;      // bytes appended so far
};
// write out every complete 16-byte piece of this lane's stage
__device__ __forceinline__ void ls_flush(LinStream& s) {
    const uint32_t np = s.off >> 4;
#pragma unroll
    for (uint32_t k = 0; k < PB_STAGE / 16; k++) {
        if (k < np) *reinterpret_cast<uint4*>(s.gp + k * 16) = *reinterpret_cast<const uint4*>(s.lds + k * 16);
    }
    if (np && (s.off & 15u)) *reinterpret_cast<uint4*>(s.lds) = *reinterpret_cast<const uint4*>(s.lds + np * 16);
    s.gp += np * 16;
    s.off &= 15u;
}
// general-purpose append (headers, separators): checks for room itself
__device__ __forceinline__ void ls_put(LinStream& s, uint32_t byte) {
    s.lds[s.off] = (uint8_t)byte;
    s.off++; s.pos++;
    if (s.off >= 128u) ls_flush(s);
}
// append `n` (0..2) bytes given in the low bytes of `bytes`; the caller keeps the stage from overflowing
__device__ __forceinline__ void ls_put2(LinStream& s, uint32_t bytes, uint32_t n) {
    s.lds[s.off] = (uint8_t)bytes; s.lds[s.off + 1u] = (uint8_t)(bytes >> 8);
    s.off += n; s.pos += n;
}
// `count` copies of one character (the quality line: half of a record): past the next 16-byte boundary they go
// straight to the pool as 16-byte stores of a constant, without touching LDS
__device__ __forceinline__ void ls_fill(LinStream& s, uint32_t byte, uint64_t count) {
    while (count && (s.off & 15u)) { ls_put(s, byte); count--; }
    if (count >= 16) {
        ls_flush(s);                              // off is a multiple of 16: everything staged leaves, off = 0
        const uint32_t w = byte * 0x01010101u;
        const uint4 v = make_uint4(w, w, w, w);
        const uint64_t n16 = count >> 4;
        for (uint64_t k = 0; k < n16; k++) *reinterpret_cast<uint4*>(s.gp + k * 16) = v;
        s.gp += n16 * 16; s.pos += n16 * 16;
        count &= 15u;
    }
    while (count) { ls_put(s, byte); count--; }
}
// end of the lane's stream: complete pieces, then the remaining bytes one by one
__device__ __forceinline__ void ls_finish(LinStream& s) {
    ls_flush(s);
    for (uint32_t j = 0; j < s.off; j++) s.gp[j] = s.lds[j];
}

// Lanes are regrouped read by read.  A wave runs the per-position loops of its 64 lanes in lock-step, so it takes as
// long as its longest read, and PacBio reads differ by thousands of positions (5-15 kb in BASELINE configs[4]: a third
// of the issue slots went to lanes waiting for the longest read of their wave).  The lane is still the unit of the
// computation -- its own pcg64, quotas, distribution states, output stream -- but which THREAD carries it is decided
// anew for every read: once the read's length is known, the 1024 lanes of the workgroup are ranked by it, every thread
// parks its lane's state (PB_XCHG_WORDS words, through a [word][lane] scratch in HBM, coalesced) in the slot of the
// lane's rank and takes over the lane in its own slot, so that each wave gets 64 reads of nearly the same length (the
// k-th wave the k-th 16th of the lengths).  Results do not depend on the grouping: no per-lane value is derived from
// the thread index.
template <bool HAP>
__global__ void __launch_bounds__(PB_BLOCK, 1024 / PB_BLOCK)      // 16 waves per CU, 4 per SIMD -> at most 128 VGPRs
pacbio_kernel(PacbioKernelParams P) {
    extern __shared__ __align__(16) uint8_t pb_smem[];                 // PB_LDS_BYTES of dynamic LDS (more than the static limit)
    uint8_t* const stage = pb_smem;                                    // [lane of the workgroup]: travels with the lane, not the thread
    uint32_t* const s_key = reinterpret_cast<uint32_t*>(pb_smem + PB_STAGE * PB_BLOCK);
    const uint32_t wg0 = blockIdx.x * blockDim.x;
    uint32_t lane = wg0 + threadIdx.x;
    bool valid = lane < P.n_lanes;

    LaneRng rng;
    rng.e = jk_pcg_limbs(jk_pcg_seed(P.seeds + (size_t)(valid ? lane : 0) * 8));
    jk_gamma_state ln_st; ln_st.saved = 0.0; ln_st.saved_available = 0; ln_st.fail = 0;     // lognormal_distribution::_M_nd
    jk_gamma_state chi_st; chi_st.saved = 0.0; chi_st.saved_available = 0; chi_st.fail = 0;  // chi_squared -> gamma -> _M_nd

    // (a lane's read count fits 32 bits: plan_lanes refuses more)
    uint32_t quota = valid ? (uint32_t)P.lane_reads[lane] : 0;
    uint32_t made = 0, in_pool = 0;
    const uint32_t pool_size = P.pool_size > 0xffffffffULL ? 0xffffffffu : (uint32_t)P.pool_size;
    const uint32_t n_cells = HAP ? P.h.n_haps * P.g.n_chroms : P.g.n_chroms;
    uint32_t ci = 0;
    uint32_t ccnt = (n_cells && valid) ? P.chrom_reads[lane] : 0;
    uint32_t cur_hap = 0xffffffffu;

    // what follows from the lane's identity (set again whenever the thread takes over another lane)
    uint64_t lane_cap = 0;
    uint64_t* evl = nullptr; uint64_t* hl = nullptr;
    LinStream o;
    o.gp = nullptr; o.lds = stage; o.off = 0; o.pos = 0;
    auto bind_lane = [&]() {
        const uint32_t l = valid ? lane : wg0;
        const uint32_t tile = l >> 6;
        lane_cap = (P.pool_off[tile + 1] - P.pool_off[tile]) >> 6;          // a multiple of 128 (host)
        evl = P.ev + l; hl = P.hist + l;
        o.lds = stage + (l - wg0) * PB_STAGE;
    };
    bind_lane();
    if (valid) o.gp = P.pool + P.pool_off[lane >> 6] + (uint64_t)(lane & 63u) * lane_cap;      // this lane's contiguous region

    uint32_t err = 0;
    const size_t ev_stride = (size_t)P.n_lanes;
    const uint64_t max_pos = (uint64_t)P.ev_words * 32u;
    const size_t xstride = (size_t)gridDim.x * PB_BLOCK;

    uint64_t L = 0, read_start = 0, chrom_len = 0;
    bool is_dup = false;
    // The reference copies a read's `space` source bases into a std::string member that never shrinks
    // (RefChrom::fill_read, src/ref_classes.h:102-116) and then walks read positions until the read has its
    // length (src/hts_pacbio.cpp:381-400).  For a duplicate that lost deletions because it abuts the chromosome
    // end (:277-285), or for a read as long as its chromosome, that walk can run a few positions past `space` and
    // picks up what EARLIER reads of the same thread left in the buffer.  Those bytes are reproduced from a
    // per-lane stack of the reads whose buffer content is still visible (strictly decreasing `space` from bottom
    // to top: a new read hides every earlier one that was not longer): entry = {space | reverse << 32 |
    // cell << 33, read_start}.  A position no remembered read covers is the string's terminating NUL if it
    // equals the string's size, and lies outside the string otherwise (undefined in the reference, refused here).
    uint32_t hdepth = 0;
    uint64_t buf_size = 0;            // size of that string: the longest window so far (position == size reads its NUL)
    const size_t hstride = (size_t)P.n_lanes;
    for (;;) {
        // a read = part A (cell, read length), regrouping of the workgroup's lanes by that length, part B (the rest).  A
        // `break` inside a part leaves the part: the lane then has an error bit set or its quota met and takes no more turns.
        const bool alive = valid && made < quota && err == 0;
        if (!__syncthreads_or(alive ? 1 : 0)) break;
        bool part_b = false;
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
                    hdepth = 0; buf_size = 0;               // ... and its own (empty) read buffer
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
                L = (uint64_t)rnd;
            } else {
                const uint32_t i0 = (uint32_t)jk_runif_index(rng(), P.n_lens);
                const uint64_t x2 = rng();
                L = P.lens[(x2 < P.len_thresh[i0]) ? i0 : P.len_alias[i0]];
            }
            if (L >= chrom_len) L = chrom_len;
        }
        if (err) break;
        part_b = true;
        } while (0);

        // ---- regroup: rank the workgroup's lanes by the length of the read they are about to make (idle lanes first)
        {
            const uint32_t key = part_b ? (uint32_t)(L < 0x7ffffffeULL ? L : 0x7ffffffeULL) + 1u : 0u;
            s_key[threadIdx.x] = key;
            __syncthreads();
            uint32_t rank = 0;
            const uint4* k4 = reinterpret_cast<const uint4*>(s_key);
            for (uint32_t j = 0; j < PB_BLOCK / 4; j++) {
                const uint4 v = k4[j];                  // (every lane of the wave reads the same address: a broadcast)
                const uint32_t t0 = 4u * j;
                rank += (v.x < key || (v.x == key && t0 < threadIdx.x)) ? 1u : 0u;
                rank += (v.y < key || (v.y == key && t0 + 1u < threadIdx.x)) ? 1u : 0u;
                rank += (v.z < key || (v.z == key && t0 + 2u < threadIdx.x)) ? 1u : 0u;
                rank += (v.w < key || (v.w == key && t0 + 3u < threadIdx.x)) ? 1u : 0u;
            }
            // park this lane at its rank, take the lane parked at this thread's slot.  Slots are dealt to waves so that
            // the waves sharing a SIMD get short and long reads alike (wave w of a workgroup sits on SIMD w % 4: with the
            // plain order SIMD 3 would hold the longest reads of every pass and the CU would wait for it): the 64-lane
            // rank groups go to the waves in a snake over the SIMDs -- groups s, 7 - s, 8 + s, 15 - s to SIMD s
            const uint32_t wv = threadIdx.x >> 6, pass_i = wv >> 2, simd = wv & 3u;
            const uint32_t grp = PB_BLOCK >= 512 ? 4u * pass_i + ((pass_i & 1u) ? 3u - simd : simd) : wv;
            const uint32_t my_slot = grp * 64u + (threadIdx.x & 63u);
            uint32_t* xp = P.xchg + wg0 + rank;
            uint32_t k = 0;
            auto put = [&](uint32_t v) { xp[(size_t)k * xstride] = v; k++; };
            auto put64 = [&](uint64_t v) { put((uint32_t)v); put((uint32_t)(v >> 32)); };
            put(lane); put((valid ? 1u : 0u) | (part_b ? 2u : 0u) | (is_dup ? 4u : 0u) | ((uint32_t)ln_st.saved_available << 3) | ((uint32_t)chi_st.saved_available << 4) |
                           ((uint32_t)ln_st.fail << 5) | ((uint32_t)chi_st.fail << 6));
            put(rng.e.s0); put(rng.e.s1); put(rng.e.s2); put(rng.e.s3); put64(rng.e.inc_lo); put64(rng.e.inc_hi);
            put64(jk_d2u(ln_st.saved)); put64(jk_d2u(chi_st.saved));
            put(quota); put(made); put(in_pool); put(ci); put(ccnt); put(cur_hap);
            put64((uint64_t)(uintptr_t)o.gp); put(o.off); put64(o.pos); put(err);
            put64(L); put64(read_start); put64(chrom_len); put(hdepth); put64(buf_size);
            static_assert(PB_XCHG_WORDS >= 38, "lane state does not fit its exchange record");
            __syncthreads();
            const uint32_t* gp = P.xchg + wg0 + my_slot;
            k = 0;
            auto get = [&]() -> uint32_t { const uint32_t v = gp[(size_t)k * xstride]; k++; return v; };
            auto get64 = [&]() -> uint64_t { const uint64_t a = get(); const uint64_t b = get(); return a | (b << 32); };
            lane = get();
            const uint32_t fl = get();
            valid = fl & 1u; part_b = (fl >> 1) & 1u; is_dup = (fl >> 2) & 1u;
            ln_st.saved_available = (fl >> 3) & 1u; chi_st.saved_available = (fl >> 4) & 1u; ln_st.fail = (fl >> 5) & 1u; chi_st.fail = (fl >> 6) & 1u;
            rng.e.s0 = get(); rng.e.s1 = get(); rng.e.s2 = get(); rng.e.s3 = get(); rng.e.inc_lo = get64(); rng.e.inc_hi = get64();
            ln_st.saved = jk_u2d(get64()); chi_st.saved = jk_u2d(get64());
            quota = get(); made = get(); in_pool = get(); ci = get(); ccnt = get(); cur_hap = get();
            o.gp = reinterpret_cast<uint8_t*>((uintptr_t)get64()); o.off = get(); o.pos = get64(); err = get();
            L = get64(); read_start = get64(); chrom_len = get64(); hdepth = get(); buf_size = get64();
            bind_lane();
            __syncthreads();           // (the records are rewritten in the next round)
        }

        if (part_b) do {
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
            // (uint64) of a negative value is undefined in the reference; x86-64 gives a huge number -> 93
            const uint32_t q = (qv < 0 || qv > 93.0) ? 93u : (uint32_t)qv;
            (side == 0 ? qual_left : qual_right) = q + 33u;
        }
        if (err) break;

        // ---- pass 1: one draw per position (PacBioQualityError::sample, src/hts_pacbio.h:292-317)
        // Counters are 32-bit (a read is shorter than 2^31 positions: the event scratch is), the spare chromosome
        // length is clamped to 2^31-1 (it only matters when it reaches 0, at most one step per position), and the
        // side of the split is a per-iteration select of the cut points instead of state that is switched.
        uint32_t cur = 0, pos = 0, n_ins = 0, n_del = 0;
        if (L > 0x7fffffffULL) { err |= JK_KERR_PB_TOO_LONG; break; }
        {
            // u > cum[2]  <=>  x >= t_none ; u < cum[0]  <=>  x < t_ins ; u < cum[1]  <=>  x < t_del
            uint64_t tL[3], tR[3]; bool aL[3], aR[3];
            tL[0] = cut_point<true>(cumL[2], &aL[0]); tL[1] = cut_point<false>(cumL[0], &aL[1]); tL[2] = cut_point<false>(cumL[1], &aL[2]);
            tR[0] = cut_point<true>(cumR[2], &aR[0]); tR[1] = cut_point<false>(cumR[0], &aR[1]); tR[2] = cut_point<false>(cumR[1], &aR[2]);
            // bit 0: never "none", bit 1: always insertion, bit 2: always deletion (cut points that cover every draw)
            const uint32_t fL = (aL[0] ? 1u : 0u) | (aL[1] ? 2u : 0u) | (aL[2] ? 4u : 0u);
            const uint32_t fR = (aR[0] ? 1u : 0u) | (aR[1] ? 2u : 0u) | (aR[2] ? 4u : 0u);
            const uint32_t L32 = (uint32_t)L;
            const uint32_t split32 = split_pos > 0xffffffffULL ? 0xffffffffu : (uint32_t)split_pos;
            const uint64_t spare = chrom_len - L;
            uint32_t extra = spare > 0x7fffffffULL ? 0x7fffffffu : (uint32_t)spare;
            // The cut points of the lane's current side live in registers and are swapped once, when `cur` reaches the
            // split (cur never decreases); the "covers every draw" flags are almost never set, so their tests sit
            // behind a wave-uniform switch; the two bit planes of a word are built as 32-bit values with the wave's
            // common position as the shift, and the event counts come from popcounts per word.
            uint64_t t_none = tL[0], t_ins = tL[1], t_del = tL[2];
            uint32_t f = fL;
            bool on_right = false;
            const bool any_f = __builtin_amdgcn_ballot_w64((fL | fR) != 0u) != 0;
            uint32_t plo = 0, phi = 0;
            auto close_word = [&](uint32_t w) {
                evl[(size_t)w * ev_stride] = (uint64_t)plo | ((uint64_t)phi << 32);
                n_ins += (uint32_t)__popc(plo & ~phi); n_del += (uint32_t)__popc(phi & ~plo);
                plo = 0; phi = 0;
            };
            // (the position counter is the same in every lane that is still drawing: kept wave-uniform, the lane's own
            // count is what it was when the lane left the loop)
            const uint32_t Lm1 = L32 - 1u;
            const uint32_t max_pos32 = max_pos > 0xffffffffULL ? 0xffffffffu : (uint32_t)max_pos;
            uint32_t upos = 0;
            while (cur < L32 && upos < max_pos32) {
                if (__builtin_amdgcn_ballot_w64(!on_right && cur >= split32)) {     // (the reference switches sides when cur reaches split_pos)
                    if (!on_right && cur >= split32) { t_none = tR[0]; t_ins = tR[1]; t_del = tR[2]; f = fR; on_right = true; }
                }
                const uint64_t x = rng();
                // same decision tree as the reference (src/hts_pacbio.h:296-314), written without divergent branches
                bool none = x >= t_none, lt_ins = x < t_ins, lt_del = x < t_del;
                if (any_f) { none = !(f & 1u) && none; lt_ins = (f & 2u) || lt_ins; lt_del = (f & 4u) || lt_del; }
                const bool ins = !none && lt_ins;
                const bool del = !none && !ins && lt_del;
                const bool sub = !none && !ins && !del;
                const bool ins_rec = ins && (cur < Lm1);              // an insertion at the last base is not recorded
                const bool del_rec = del && (extra > 0u);             // nor a deletion without spare chromosome
                extra = extra + (ins_rec ? 1u : 0u) - (del_rec ? 1u : 0u);
                cur += (ins_rec ? 1u : 0u) + (del ? 0u : 1u);
                const uint32_t bit = 1u << (upos & 31u);
                plo |= (ins_rec || sub) ? bit : 0u;        // code bit 0: insertion (1) or substitution (3)
                phi |= (del_rec || sub) ? bit : 0u;        // code bit 1: deletion (2) or substitution (3)
                upos++;
                if ((upos & 31u) == 0) close_word((upos >> 5) - 1u);
            }
            pos = upos;
            if (cur < L32) err |= JK_KERR_PB_TOO_LONG;
            else if (pos & 31u) close_word(pos >> 5);
        }
        if (err) break;
        uint64_t space = L + n_del - n_ins;
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
                    uint64_t* wp = evl + (scan >> 5) * ev_stride;
                    const uint64_t wv = *wp;
                    const uint32_t bit = (uint32_t)scan & 31u;         // deletion = code 2: plane 1 set, plane 0 clear
                    if (((wv >> (32u + bit)) & 1ULL) && !((wv >> bit) & 1ULL)) { *wp = wv & ~(1ULL << (32u + bit)); break; }
                }
                n_del--; space--;
            }
            if (space + read_start > chrom_len) give_up = true;
        }

        if (!give_up) {
            // ---- append_pool (src/hts_pacbio.cpp:350-414)
            const bool reverse = jk_runif_lt_half(rng());
            {
                uint32_t hdr_len = P.g.hdr_off[ci + 1] - P.g.hdr_off[ci];
                if ((uint64_t)o.pos + hdr_len + 24 + 2 * L + 8 > lane_cap) { err |= JK_KERR_POOL_OVERFLOW; break; }
                for (uint32_t h = P.g.hdr_off[ci]; h < P.g.hdr_off[ci + 1]; h++) ls_put(o, P.g.hdr_blob[h]);
                uint64_t v = read_start, packed_lo = 0, packed_hi = 0; uint32_t nd = 0;
                do {
                    const uint64_t q = v / 10, d = v - q * 10;
                    packed_hi = (packed_hi << 4) | (packed_lo >> 60);
                    packed_lo = (packed_lo << 4) | d;
                    v = q; nd++;
                } while (v);
                for (uint32_t d = 0; d < nd; d++) {
                    ls_put(o, '0' + (uint32_t)(packed_lo & 15u));
                    packed_lo = (packed_lo >> 4) | (packed_hi << 60); packed_hi >>= 4;
                }
                ls_put(o, '-');
                ls_put(o, reverse ? 'R' : 'F');
                ls_put(o, '\n');
            }
            // source walker: read[p] = forward chrom[start + p], reverse cmp(chrom[start + space - 1 - p])
            const uint8_t* const gseq = P.g.seq;
            const uint64_t chrom_off = P.g.chrom_off[HAP ? ci % P.g.n_chroms : ci];
            uint64_t gaddr = 0, gbuf = 0; uint32_t gcnt = 0;
            // reverse strand: the chunk is byte-swapped and its base codes complemented when it is loaded (code ^ 2 for
            // bytes 0..3 only: the flip is gated by bit 2 of each byte being clear, which also leaves 'N', 0xfc..0xff
            // and every byte that could turn INTO 'N' alone -- a flip never changes bit 2)
            auto rc8 = [](uint64_t v) -> uint64_t {
                v = __builtin_bswap64(v);
                return v ^ (((~v) >> 1) & 0x0202020202020202ULL);
            };
            auto src_init = [&](uint64_t a0) {
                const uint64_t ch = a0 & ~7ULL;
                uint64_t v = *reinterpret_cast<const uint64_t*>(gseq + ch);
                const uint32_t k = (uint32_t)a0 & 7u;
                if (reverse) { v = rc8(v); gbuf = v >> (8u * (7u - k)); gcnt = k + 1; gaddr = ch - 8; }
                else { gbuf = v >> (8u * k); gcnt = 8u - k; gaddr = ch + 8; }
            };
            auto src_next = [&]() -> uint32_t {
                const uint32_t c = (uint32_t)gbuf & 0xffu;
                gbuf >>= 8;
                if (--gcnt == 0) {
                    uint64_t v = *reinterpret_cast<const uint64_t*>(gseq + gaddr);
                    if (reverse) { v = rc8(v); gaddr -= 8; } else gaddr += 8;
                    gbuf = v; gcnt = 8;
                }
                return c;
            };
            uint64_t seg_end = ~0ULL;
            int64_t mcur = -1;
            auto seg_enter = [&](uint64_t p) {
                const uint64_t hpos = reverse ? (read_start + space - 1 - p) : (read_start + p);
                if (HAP) {
                    const HapSeg sg = hap_resolve(P.h, chrom_off, ci, mcur, hpos);
                    src_init(sg.addr);
                    const uint64_t avail = reverse ? (hpos - sg.begin + 1) : (sg.end - hpos);
                    seg_end = p + avail;
                } else src_init(chrom_off + hpos);
            };
            if (HAP) mcur = hap_search(P.h, ci, reverse ? (read_start + space - 1) : read_start);
            if (space > 0) seg_enter(0);
            uint64_t rd_pos = 0;           // the position src_next() delivers next (the word path below does not use it)
            // word path (reference genome): position p of the window is the byte at A + p, or the complement of the one at A - p
            const uint64_t A = chrom_off + (reverse ? read_start + space - 1 : read_start);
            const uint32_t rsel = reverse ? 0x04050607u : 0x03020100u;      // v_perm selectors: read order of an 8-byte chunk
            const uint32_t rcm = reverse ? 0x02020202u : 0u;                // complement of codes 0..3

            // ---- pass 2: emit bases
            // Positions are handled 32 at a time (one word of event codes).  Per word the lane first finds
            // how many of its positions it will process (the read ends when cur2 reaches L), makes the
            // draws of the insertions and substitutions among them -- a wave-uniform loop over "the j-th
            // event of my word", so the wave pays for max-over-lanes draws instead of one per position --
            // and parks each result (2 bits) at its position.  The per-position loop then only moves bytes.
            uint64_t cur2 = 0, p2 = 0;
            while (cur2 < L) {
                const uint64_t evw = (p2 < pos) ? evl[(p2 >> 5) * ev_stride] : 0;
                const uint32_t lo = (uint32_t)evw, hi = (uint32_t)(evw >> 32);        // bit k: code bit 0 / 1 of position k
                const uint32_t insm = lo & ~hi, delm = hi & ~lo;
                // smallest k in [0, 32] with cur2 + k + #ins(<k) - #del(<k) >= L (monotone in k): positions < k are processed
                uint32_t kcut = 32u;
                if (__builtin_amdgcn_ballot_w64(L - cur2 < 64u) != 0) {        // (far from the read's end every position of the word is processed)
                    uint32_t a = 0;                    // invariant: f(a) < L
                    const uint64_t need = L - cur2;    // > 0
#pragma unroll
                    for (uint32_t step = 16; step > 0; step >>= 1) {
                        const uint32_t k = a + step;
                        const uint32_t below = (1u << k) - 1u;          // k <= 31 here
                        const uint64_t f = (uint64_t)k + __popc(insm & below) - __popc(delm & below);
                        a = (f < need) ? k : a;
                    }
                    // a = largest k in [0, 31] with f(k) < need; position a is processed, a + 1 may be the cut
                    kcut = a + 1u;
                }
                const uint32_t proc = kcut >= 32u ? 0xffffffffu : ((1u << kcut) - 1u);
                // draws of this word's events, in position order (src/hts_pacbio.cpp:430-457)
                uint32_t evm = lo & proc;              // code 1 (insertion) or 3 (substitution)
                uint64_t res = 0; uint32_t nul = 0;
                while (__builtin_amdgcn_ballot_w64(evm != 0u)) {
                    if (evm) {
                        const uint32_t k = (uint32_t)__builtin_ctz(evm);
                        evm &= evm - 1u;
                        const uint64_t x = rng();
                        // substitution: mm_nucleos[nt][(uint64)(runif_01 * 3)]; insertion: jlp::bases[(uint64)(runif_01 * 4)]
                        const uint32_t nidx = ((hi >> k) & 1u) ? 3u : 4u;
                        const uint32_t code = runif_index32(x, nidx);
                        const bool is_nul = code >= nidx;                                   // index past the string: its NUL
                        nul |= is_nul ? (1u << k) : 0u;
                        res |= (uint64_t)(code & 3u) << (2u * k);
                    }
                }
                // ---- word path: all 32 positions of the word from registers.  Taken when, for every lane of the wave,
                // the positions it processes lie in its window, are TCAG, and no draw hit the NUL of the base strings.
                // The lane loads its 32 source bytes (four unaligned 8-byte loads, put in read order and complemented by
                // v_perm as in the Illumina kernel), maps codes to characters four at a time, patches the (rare)
                // substitutions, and then writes per position "base, inserted base" into its stage, advancing the
                // offset by 1 - deleted and by inserted: a byte that is not part of the read is overwritten by the next.
                if (!HAP) {
                    const bool word_ok = (p2 + kcut <= space) && nul == 0u;
                    if (__builtin_amdgcn_ballot_w64(!word_ok) == 0) {
                        uint32_t wq[8], bad = 0;
#pragma unroll
                        for (uint32_t g = 0; g < 4; g++) {
                            uint32_t v[2];
                            __builtin_memcpy(v, gseq + (reverse ? A - (p2 + 8u * g) - 7u : A + p2 + 8u * g), 8);
                            wq[2 * g] = __builtin_amdgcn_perm(v[1], v[0], rsel) ^ rcm;
                            wq[2 * g + 1] = __builtin_amdgcn_perm(v[1], v[0], rsel ^ 0x04040404u) ^ rcm;
                            bad |= (v[0] | v[1]) & 0xfcfcfcfcu;
                        }
                        // Non-TCAG bytes: 'N' is copied through on both strands (cmp_map keeps it), so a word may hold them
                        // (the N runs of real assemblies put one into most waves); any other byte goes the per-position way.
                        auto nzb = [](uint32_t v) -> uint32_t { return (((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v) & 0x80808080u; };   // 0x80 per non-zero byte
                        const bool any_n = __builtin_amdgcn_ballot_w64(bad != 0u) != 0;
                        bool chars_ok = true;
                        if (any_n) {
                            const uint32_t nkey = 0x4e4e4e4eu ^ rcm;        // what 'N' looks like after the strand flip of bit 1
                            uint32_t other = 0;
#pragma unroll
                            for (uint32_t q = 0; q < 8; q++) other |= nzb(wq[q] & 0xfcfcfcfcu) & nzb(wq[q] ^ nkey);
                            chars_ok = __builtin_amdgcn_ballot_w64(other != 0u) == 0;
                        }
                        if (chars_ok) {
                            if (space > buf_size) buf_size = space;
                            const uint32_t keep = proc & ~delm, insp = insm & proc;
                            uint32_t cw[8];
#pragma unroll
                            for (uint32_t q = 0; q < 8; q++) cw[q] = __builtin_amdgcn_perm(0u, 0x47414354u, wq[q]);
                            if (any_n) {
#pragma unroll
                                for (uint32_t q = 0; q < 8; q++) {
                                    const uint32_t m = (nzb(wq[q] & 0xfcfcfcfcu) >> 7) * 0xffu;      // 0xff per non-TCAG byte
                                    cw[q] = (cw[q] & ~m) | (0x4e4e4e4eu & m);
                                }
                            }
                            uint32_t subm = lo & hi & proc;
                            while (__builtin_amdgcn_ballot_w64(subm != 0u)) {
                                if (subm) {
                                    const uint32_t k = (uint32_t)__builtin_ctz(subm);
                                    subm &= subm - 1u;
                                    const uint32_t q = k >> 2, sh = 8u * (k & 3u);
                                    uint32_t wsel = wq[0];
#pragma unroll
                                    for (uint32_t i = 1; i < 8; i++) wsel = (q == i) ? wq[i] : wsel;
                                    const uint32_t nt = (wsel >> sh) & 3u;
                                    const uint32_t code = (uint32_t)(res >> (2u * k)) & 3u;
                                    // (a substitution on a non-TCAG base gives 'N': mm_nucleos[4] = "NNN", src/hts.h:46)
                                    const uint32_t sc = (((wsel >> sh) & 0xfcu) ? (uint32_t)'N' : base_char(code + (code >= nt ? 1u : 0u))) << sh;
                                    const uint32_t clr = ~(0xffu << sh);
#pragma unroll
                                    for (uint32_t i = 0; i < 8; i++) cw[i] = (q == i) ? ((cw[i] & clr) | sc) : cw[i];
                                }
                            }
                            uint32_t off = o.off;
#pragma unroll
                            for (uint32_t q = 0; q < 8; q++) {
                                if ((q & 1u) == 0 && __builtin_amdgcn_ballot_w64(off >= PB_FLUSH_AT)) { o.off = off; ls_flush(o); off = o.off; }
                                // the four inserted-base codes of this quad (2 bits each), spread to bytes, as characters
                                const uint32_t r8 = (uint32_t)(res >> (8u * q)) & 0xffu;
                                const uint32_t ic = __builtin_amdgcn_perm(0u, 0x47414354u, (r8 | (r8 << 6) | (r8 << 12) | (r8 << 18)) & 0x03030303u);
#pragma unroll
                                for (uint32_t j = 0; j < 4; j++) {
                                    const uint32_t k = 4u * q + j;
                                    o.lds[off] = (uint8_t)(cw[q] >> (8u * j));
                                    off += (keep >> k) & 1u;
                                    o.lds[off] = (uint8_t)(ic >> (8u * j));
                                    off += (insp >> k) & 1u;
                                }
                            }
                            const uint32_t nb = (uint32_t)__popc(keep) + (uint32_t)__popc(insp);
                            o.off = off; o.pos += nb; cur2 += nb;
                            p2 += kcut;
                            continue;
                        }
                    }
                }
                // bytes of a position whose buffer character is `ch` (code `nt` if it is one of TCAG): up to two, in the
                // low bytes, with their count in bits 16..17
                auto apply = [&](uint32_t k, uint32_t ch, bool is_nt, uint32_t nt) -> uint32_t {
                    const uint32_t b0bit = (lo >> k) & 1u, b1bit = (hi >> k) & 1u;
                    const bool is_ins = b0bit && !b1bit, is_del = !b0bit && b1bit, is_sub = b0bit && b1bit;
                    const uint32_t code = (uint32_t)(res >> (2u * k)) & 3u;
                    const bool is_nul = (nul >> k) & 1u;
                    const uint32_t ins_ch = is_nul ? 0u : base_char(code);
                    const uint32_t sub_ch = is_nul ? 0u : (is_nt ? base_char(code + (code >= nt ? 1u : 0u)) : (uint32_t)'N');
                    const uint32_t b0 = is_sub ? sub_ch : ch;
                    const uint32_t nb = is_del ? 0u : (is_ins ? 2u : 1u);
                    return (is_del ? 0u : (b0 | (is_ins ? (ins_ch << 8) : 0u))) | (nb << 16);
                };
                auto emit = [&](uint32_t k) -> uint32_t {
                    if (HAP && p2 + k >= seg_end) seg_enter(p2 + k);
                    const uint32_t c = src_next();
                    // character of read[p2 + k] as the reference sees it (cmp_map for the reverse strand)
                    const bool is_nt = c < 4u;
                    const uint32_t nt = is_nt ? c : 4u;            // (already complemented on the reverse strand)
                    const uint32_t ch = is_nt ? base_char(nt) : (reverse ? (c == 'N' ? (uint32_t)'N' : 0u) : jk_decode_other(c));
                    return apply(k, ch, is_nt, nt);
                };
                // position past this read's window: the character an earlier read left in the buffer (rare)
                auto emit_stale = [&](uint32_t k) -> uint32_t {
                    const uint64_t q = p2 + k;
                    uint32_t d = hdepth;
                    uint64_t e0 = 0, e1 = 0;
                    bool found = false;
                    while (d > 0) {                      // newest first: the first one long enough wrote position q last
                        d--;
                        e0 = hl[(2 * d) * hstride]; e1 = hl[(2 * d + 1) * hstride];
                        if ((e0 & 0xffffffffULL) > q) { found = true; break; }
                    }
                    if (!found) {
                        // read[size()] is the string's terminator (defined); anything further is outside the string
                        if (q == buf_size || P.undefined_as_nul) return apply(k, 0u, false, 4u);
                        err |= JK_KERR_PB_SPACE;
                        return 1u << 16;
                    }
                    const uint64_t sp_j = e0 & 0xffffffffULL;
                    const bool rev_j = (e0 >> 32) & 1ULL;
                    const uint32_t cell_j = (uint32_t)(e0 >> 33);
                    const uint64_t hpos = rev_j ? (e1 + sp_j - 1 - q) : (e1 + q);
                    uint64_t addr;
                    const uint64_t coff = P.g.chrom_off[HAP ? cell_j % P.g.n_chroms : cell_j];
                    if (HAP) {
                        int64_t m = hap_search(P.h, cell_j, hpos);
                        addr = hap_resolve(P.h, coff, cell_j, m, hpos).addr;
                    } else addr = coff + hpos;
                    const uint32_t c0 = gseq[addr];
                    const bool is_nt = c0 < 4u;
                    const uint32_t nt = is_nt ? (rev_j ? (c0 ^ 2u) : c0) : 4u;
                    const uint32_t ch = is_nt ? base_char(nt) : (rev_j ? (c0 == 'N' ? (uint32_t)'N' : 0u) : jk_decode_other(c0));
                    return apply(k, ch, is_nt, nt);
                };
                const uint32_t kmain = space > p2 ? (uint32_t)(space - p2 < kcut ? space - p2 : kcut) : 0u;
                if (space > buf_size) buf_size = space;       // fill_read grew the string before the walk
                if (!HAP && kmain > 0u && rd_pos != p2) seg_enter(p2);       // the word path moved on without the reader
                rd_pos = p2 + kmain;
                // two positions per ring check: they add at most 4 bytes to fewer than 4 pending ones, so one
                // word at most leaves the shift register (lanes near the end of their read run fewer positions)
                uint32_t k = 0;
                for (; k + 2u <= kmain; k += 2u) {
                    if (__builtin_amdgcn_ballot_w64(o.off >= PB_FLUSH_AT)) ls_flush(o);
                    const uint32_t r0 = emit(k), r1 = emit(k + 1u);
                    const uint32_t n0 = r0 >> 16, n1 = r1 >> 16;
                    ls_put2(o, r0 & 0xffffu, n0); ls_put2(o, r1 & 0xffffu, n1);
                    cur2 += n0 + n1;
                }
                for (; k < kcut; k++) {                  // at most one position of the window, then the stale ones
                    if (__builtin_amdgcn_ballot_w64(o.off >= PB_FLUSH_AT)) ls_flush(o);
                    const uint32_t r0 = k < kmain ? emit(k) : emit_stale(k);
                    ls_put2(o, r0 & 0xffffu, r0 >> 16);
                    cur2 += r0 >> 16;
                }
                p2 += kcut;
            }
            if (err) break;
            ls_put(o, '\n'); ls_put(o, '+'); ls_put(o, '\n');
            ls_fill(o, qual_left, split_pos < L ? split_pos : L);
            ls_fill(o, qual_right, split_pos < L ? L - split_pos : 0);
            ls_put(o, '\n');
            // this read's window now sits in the buffer: it hides every remembered read that was not longer
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
        }
        if (HAP) ccnt = ccnt > 0 ? ccnt - 1 : 0;     // n_reads_vc[hap][chr]-- (one_read) / if > 0 (re_read)

        // ---- ReadWriterOneThread::create_reads tail (src/hts.h:263-278), one read end
        made += 1; in_pool += 1;
        const uint64_t xd = rng();
        const bool dup = P.dup_all || xd < P.th_dup;
        if (dup && made < quota && in_pool < pool_size) is_dup = true;
        else { is_dup = false; if (in_pool >= pool_size || made >= quota) in_pool = 0; }
        } while (0);
    }
    if (valid) {
        ls_finish(o);
        P.lane_bytes[lane] = o.pos;
        if (o.pos > lane_cap) err |= JK_KERR_POOL_OVERFLOW;
        P.lane_made[lane] = made;
    }
    if (err) atomicOr(P.err, err);
}

// One wave that waits on the clock.  It heads the compaction chain of a PacBio launch: the generator has exactly as many
// workgroups as the device has slots for them, and the next launch and this chain become ready at the same moment -- a
// compaction workgroup that takes a slot first keeps a generator workgroup waiting until the first ones retire, 21 ms
// into a 29 ms launch (launches of 40 ms instead of 32, one or two per step at random; with RCCL in the process most
// steps: 21.7 instead of 24.2 M reads/s).  A millisecond later the generator's workgroups are all placed; the chain has
// twenty to spare.  `ticks` in units of wall_clock64 (100 MHz).
__global__ void __launch_bounds__(64) pb_delay_kernel(uint64_t ticks) {
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

}  // namespace jk
