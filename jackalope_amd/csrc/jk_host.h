// jk_host.h -- host-side set-up logic of the read-generation path (C++; no device code).
//
// Mirrors, for the GPU driver, what the reference does on the calling thread before its OpenMP
// region: read quotas (split_int, src/util.h:245-258), seed consumption order
// (mt_seeds then add_n_reads, src/hts.h:339,349-353), per-group binomial quotas (reads_per_group,
// src/hts.h:58-103), Vose alias tables (AliasSampler::construct, src/alias_sampler.h:68-106) and
// the quality -> mismatch-probability map (src/hts_illumina.h:182-187).  It then turns every
// floating-point comparison the per-read loop makes against a fixed probability into an exact
// 64-bit integer threshold on the raw pcg64 output (see jk_threshold_*), which is what the kernels
// consume.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <limits>
#include <numeric>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/jackalope_hip.h"
#include "jk_math.h"

namespace jk {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// ---- seed words -------------------------------------------------------------------------------
struct SeedReader {
    jk_seed_source src;
    uint64_t pos = 0;      // words consumed
    void take8(uint32_t* out) {
        if (src.words) {
            if (pos + 8 > src.n_words) throw Error(JK_ERR_SEEDS, "seed source exhausted: the path needs more 32-bit sub-seed words than were supplied");
            for (int i = 0; i < 8; i++) out[i] = src.words[pos + i];
        } else if (src.fn) {
            if (src.fn(src.user, out) != 0) throw Error(JK_ERR_SEEDS, "seed callback failed");
        } else {
            throw Error(JK_ERR_SEEDS, "no seed source given (args.seeds)");
        }
        pos += 8;
    }
};

// pcg64 as a C++ UniformRandomBitGenerator, for libstdc++'s binomial_distribution on the host.
struct HostPcg {
    typedef uint64_t result_type;
    jk_pcg64 e;
    static constexpr uint64_t min() { return 0; }
    static constexpr uint64_t max() { return ~uint64_t(0); }
    uint64_t operator()() { return jk_pcg_next(e); }
};

// LCG jump-ahead (engine::advance, pcg_random.hpp:419-429) as a table: map[k] carries a state 2^k steps
// forward, state -> mult * state + plus.  Stepping an arbitrary distance applies the maps of its set
// bits (they are powers of one map, so the order does not matter).
struct PcgMap { jk_u128 mult, plus; };
inline void pcg_advance_table(const jk_pcg64& e, PcgMap map[64]) {
    jk_u128 m = jk_mk128(JK_PCG_MULT_HI, JK_PCG_MULT_LO), c = jk_mk128(e.inc_hi, e.inc_lo);
    for (int k = 0; k < 64; k++) {
        map[k].mult = m; map[k].plus = c;
        c = (m + 1) * c;
        m = m * m;
    }
}
inline void pcg_advance(jk_pcg64& e, const PcgMap map[64], uint64_t steps) {
    jk_u128 st = jk_mk128(e.s_hi, e.s_lo);
    for (int k = 0; steps; k++, steps >>= 1)
        if (steps & 1u) st = map[k].mult * st + map[k].plus;
    e.s_hi = (uint64_t)(st >> 64); e.s_lo = (uint64_t)st;
}

inline std::vector<uint64_t> split_int(uint64_t x, uint64_t n) {
    std::vector<uint64_t> out(n, x / n);
    uint64_t extra = x - n * (x / n);          // the first `extra` chunks get one more
    for (uint64_t i = 0; i < extra; i++) out[i]++;
    return out;
}

// The probabilities reads_per_group() ends up using for its binomials: the loop
//     probs /= accumulate(probs); for g: ...; rest = 1 - probs[g]; probs[j > g] /= rest
// restated once (same operations in the same order), since nothing in it depends on the draws.
struct GroupChain {
    size_t G = 0;
    std::vector<double> p;        // [G-1] probability of group g's binomial
    std::vector<uint8_t> kind;    // [G-1] 0 = draw, 1 = skipped (probability 0), 2 = takes all remaining reads
    explicit GroupChain(std::vector<double> probs = {}) {
        G = probs.size();
        if (G == 0) return;
        const double total = std::accumulate(probs.begin(), probs.end(), 0.0);
        for (double& v : probs) v /= total;
        p.assign(G - 1, 0.0); kind.assign(G - 1, 1);
        for (size_t g = 0; g + 1 < G; g++) {
            if (probs[g] >= 1) { kind[g] = 2; break; }
            if (probs[g] == 0) continue;
            kind[g] = 0; p[g] = probs[g];
            const double rest = 1 - probs[g];
            for (size_t j = g + 1; j < G; j++) probs[j] /= rest;
        }
    }
};

// std::binomial_distribution<uint64_t>(t, p)(eng) as libstdc++ 11 computes it (random.tcc:1469-1680), restated:
//  * t * min(p, 1-p) < 8: the waiting-time algorithm -- a sum of -log(1 - u) / (t - x) until it exceeds -log(1 - p12);
//  * else Devroye's rejection algorithm, with param_type::_M_initialize's constants and the distribution's own
//    normal_distribution (polar method, second deviate kept) -- same expressions, same order, same libm calls, except
//    that lgamma() is called as lgamma_r(): the same function (glibc's lgamma is lgamma_r plus a store to the global
//    `signgam`), without the store, on which the host threads of the planner otherwise fight for one cache line
//    (2^21 lanes: 1.7 s on one thread, 0.6 s on 16, 0.9 s on 64 before).
// Checked draw for draw against std::binomial_distribution itself (tests/test_host_primitives.py) and, through every
// parity test, against the oracle, which calls libstdc++ as the reference does.
struct BinomDraw {
    struct Par { uint64_t t; double p; bool used; double d1, d2, s1, s2, c, a1, a123, s, lf, lp1p, q; Par() : t(0), p(0), used(false) {} };
    std::vector<Par> cache{512};           // the constants are a pure function of (t, p), and lanes repeat the same few hundred pairs
    jk_gamma_state nd{0.0, 0, 0};          // the distribution's normal_distribution member (_M_nd)
    void reset() { nd.saved = 0.0; nd.saved_available = 0; }
    static void init(Par& P, uint64_t t, double p) {       // param_type::_M_initialize, non-easy branch
        P.t = t; P.p = p; P.used = true;
        const double p12 = p <= 0.5 ? p : 1.0 - p;
        const double np = std::floor(t * p12);
        const double pa = np / t;
        const double _1p = 1 - pa;
        const double pi_4 = 0.7853981633974483096156608458198757L;
        const double d1x = std::sqrt(np * _1p * std::log(32 * np / (81 * pi_4 * _1p)));
        P.d1 = std::round(std::max<double>(1.0, d1x));
        const double d2x = std::sqrt(np * _1p * std::log(32 * t * _1p / (pi_4 * pa)));
        P.d2 = std::round(std::max<double>(1.0, d2x));
        const double spi_2 = 1.2533141373155002512078826424055226L;
        P.s1 = std::sqrt(np * _1p) * (1 + P.d1 / (4 * np));
        P.s2 = std::sqrt(np * _1p) * (1 + P.d2 / (4 * t * _1p));
        P.c = 2 * P.d1 / np;
        P.a1 = std::exp(P.c) * P.s1 * spi_2;
        const double a12 = P.a1 + P.s2 * spi_2;
        const double s1s = P.s1 * P.s1;
        P.a123 = a12 + (std::exp(P.d1 / (t * _1p)) * 2 * s1s / P.d1 * std::exp(-P.d1 * P.d1 / (2 * s1s)));
        const double s2s = P.s2 * P.s2;
        P.s = (P.a123 + 2 * s2s / P.d2 * std::exp(-P.d2 * P.d2 / (2 * s2s)));
        int sg = 0;
        P.lf = (::lgamma_r(np + 1, &sg) + ::lgamma_r(t - np + 1, &sg));
        P.lp1p = std::log(pa / _1p);
        P.q = -std::log(1 - (p12 - pa) / _1p);
    }
    static uint64_t waiting(HostPcg& eng, uint64_t t, double q) {      // _M_waiting
        uint64_t x = 0;
        double sum = 0.0;
        do {
            if (t == x) return x;
            const double e = -std::log(1.0 - jk_canonical(eng()));
            sum += e / (t - x);
            x += 1;
        } while (sum <= q);
        return x - 1;
    }
    uint64_t operator()(HostPcg& eng, uint64_t t, double p) {
        const double p12 = p <= 0.5 ? p : 1.0 - p;
        uint64_t ret;
        if (static_cast<double>(t) * p12 >= 8) {
            uint64_t pb; std::memcpy(&pb, &p, 8);
            Par& P = cache[(size_t)((t * 0x9E3779B97F4A7C15ULL) ^ (pb * 0xC2B2AE3D27D4EB4FULL)) >> 55];
            if (!P.used || P.t != t || P.p != p) init(P, t, p);
            double x;
            const double naf = (1 - std::numeric_limits<double>::epsilon()) / 2;
            const double thr = static_cast<double>(std::numeric_limits<uint64_t>::max()) + naf;
            const double np = std::floor(t * p12);
            const double spi_2 = 1.2533141373155002512078826424055226L;
            const double a1 = P.a1;
            const double a12 = a1 + P.s2 * spi_2;
            const double a123 = P.a123;
            const double s1s = P.s1 * P.s1;
            const double s2s = P.s2 * P.s2;
            bool reject;
            do {
                const double u = P.s * jk_canonical(eng());
                double v;
                if (u <= a1) {
                    const double n = jk_normal(nd, eng);
                    const double y = P.s1 * std::abs(n);
                    reject = y >= P.d1;
                    if (!reject) {
                        const double e = -std::log(1.0 - jk_canonical(eng()));
                        x = std::floor(y);
                        v = -e - n * n / 2 + P.c;
                    }
                } else if (u <= a12) {
                    const double n = jk_normal(nd, eng);
                    const double y = P.s2 * std::abs(n);
                    reject = y >= P.d2;
                    if (!reject) {
                        const double e = -std::log(1.0 - jk_canonical(eng()));
                        x = std::floor(-y);
                        v = -e - n * n / 2;
                    }
                } else if (u <= a123) {
                    const double e1 = -std::log(1.0 - jk_canonical(eng()));
                    const double e2 = -std::log(1.0 - jk_canonical(eng()));
                    const double y = P.d1 + 2 * s1s * e1 / P.d1;
                    x = std::floor(y);
                    v = (-e2 + P.d1 * (1 / (t - np) - y / (2 * s1s)));
                    reject = false;
                } else {
                    const double e1 = -std::log(1.0 - jk_canonical(eng()));
                    const double e2 = -std::log(1.0 - jk_canonical(eng()));
                    const double y = P.d2 + 2 * s2s * e1 / P.d2;
                    x = std::floor(-y);
                    v = -e2 - P.d2 * y / (2 * s2s);
                    reject = false;
                }
                reject = reject || x < -np || x > t - np;
                if (!reject) {
                    int sg = 0;
                    const double lfx = ::lgamma_r(np + x + 1, &sg) + ::lgamma_r(t - (np + x) + 1, &sg);
                    reject = v > P.lf - lfx + x * P.lp1p;
                }
                reject |= x + np >= thr;
            } while (reject);
            x += np + naf;
            const uint64_t z = waiting(eng, t - uint64_t(x), P.q);
            ret = uint64_t(x) + z;
        } else {
            ret = waiting(eng, t, -std::log(1 - p12));
        }
        if (p12 != p) ret = t - ret;
        return ret;
    }
};

// reads_per_group (src/hts.h:58-103) with the chain precomputed; out[g * out_stride], g < G.
template <typename Store>
inline void split_with_chain(uint64_t n_reads, const GroupChain& ch, const uint32_t* w, BinomDraw& bd, Store store) {
    const size_t G = ch.G;          // (the destination starts zeroed: only non-zero groups are stored)
    if (G == 0 || n_reads == 0) return;
    HostPcg eng{jk_pcg_seed(w)};
    bd.reset();                                       // a fresh distribution object per call, as in the reference
    for (size_t g = 0; g + 1 < G; g++) {
        if (ch.kind[g] == 2) { store(g, n_reads); return; }
        if (ch.kind[g] == 1) continue;
        const uint64_t k = bd(eng, n_reads, ch.p[g]);
        if (k) store(g, k);
        n_reads -= k;
        if (n_reads == 0) break;
    }
    if (n_reads) store(G - 1, n_reads);
}

// Sequential conditional binomials over groups (reference: src/hts.h:58-103) from a fresh engine seeded
// with the 8 words `w`.
inline std::vector<uint64_t> reads_per_group_w(uint64_t n_reads, const std::vector<double>& probs, const uint32_t* w) {
    std::vector<uint64_t> out(probs.size(), 0);
    GroupChain ch(probs);
    BinomDraw bd;
    split_with_chain(n_reads, ch, w, bd, [&](size_t g, uint64_t v) { out[g] = v; });
    return out;
}
// ... taking the words itself.  Consumes 8 seed words whenever n_reads > 0 and there is at least one
// group, even when no binomial is then drawn.
inline std::vector<uint64_t> reads_per_group(uint64_t n_reads, const std::vector<double>& probs, SeedReader& seeds) {
    if (n_reads == 0 || probs.empty()) return std::vector<uint64_t>(probs.size(), 0);
    uint32_t w[8];
    seeds.take8(w);
    return reads_per_group_w(n_reads, probs, w);
}

struct AliasTable {
    std::vector<double> prob;
    std::vector<uint64_t> alias;
};
inline AliasTable alias_build(std::vector<double> p) {
    const uint64_t n = p.size();
    AliasTable t;
    t.prob.assign(n, 0.0);
    t.alias.assign(n, 0);
    double s0 = 0, s1 = 0;
    uint64_t i = 0;
    for (; i + 1 < n; i += 2) { s0 += p[i]; s1 += p[i + 1]; }
    if (i < n) s0 += p[i];
    const double total = s0 + s1;
    for (double& v : p) v /= total;
    for (double& v : p) v *= static_cast<double>(n);
    std::deque<uint64_t> small, large;
    for (uint64_t k = 0; k < n; k++) (p[k] < 1 ? small : large).push_back(k);
    while (!small.empty() && !large.empty()) {
        const uint64_t l = small.front(), g = large.front();
        small.pop_front(); large.pop_front();
        t.prob[l] = p[l];
        t.alias[l] = g;
        p[g] = (p[g] + p[l]) - 1;
        (p[g] < 1 ? small : large).push_back(g);
    }
    for (uint64_t g : large) t.prob[g] = 1;
    for (uint64_t l : small) t.prob[l] = 1;
    return t;
}

// ---- exact integer thresholds -----------------------------------------------------------------
// u(x) = (double)runif_01 for raw engine output x is monotone non-decreasing in x, so for a fixed
// double c the sets {x : u(x) < c} and {x : u(x) <= c} are prefixes [0, th).  `all` marks th = 2^64.
struct Threshold { uint64_t th; bool all; };

template <typename Pred>   // pred(x) true on a prefix of [0, 2^64)
inline Threshold prefix_end(Pred pred) {
    if (!pred(0)) return {0, false};
    if (pred(~uint64_t(0))) return {0, true};
    uint64_t lo = 0, hi = ~uint64_t(0);         // pred(lo) true, pred(hi) false
    while (hi - lo > 1) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (pred(mid)) lo = mid; else hi = mid;
    }
    return {hi, false};
}
inline Threshold threshold_lt(double c) { return prefix_end([c](uint64_t x) { return jk_runif_double(x) < c; }); }
inline Threshold threshold_le(double c) { return prefix_end([c](uint64_t x) { return jk_runif_double(x) <= c; }); }

// ---- Illumina error-model tables, flattened for the kernel ------------------------------------
struct IlluminaTables {
    uint32_t read_length = 0, n_ends = 0;
    std::vector<uint32_t> info;     // [end][pos][nt] : first entry (24 bits) | n entries (8 bits)
    std::vector<uint64_t> thresh;   // per entry: draw x2 picks the entry itself iff x2 < thresh
    std::vector<uint16_t> quals;    // per entry: quality if picked | quality of its alias << 8
    std::vector<uint64_t> mm_thresh;  // [256] : mismatch iff x3 < mm_thresh[q]  (qual_prob_map, hts_illumina.h:182-187)
};

// The same tables in the form the kernel reads (one LDS/L2 access per step of a base, no unpacking arithmetic):
//   mm2   [256] u64, indexed by the quality CHARACTER c = (q + 33) & 255 (what fill_read_qual emits): mm_thresh[q]
//   tab   one blob of u32: info2 [end][pos][nt] {byte offset of the position's first alias entry in the blob,
//         number of entries}, then per alias entry {HIGH word of its cut point, 8*char if kept | 8*char of the alias << 16}
//         (8*char = byte offset of the character's cut point in mm2)
//   lo    per alias entry the LOW word of its cut point.  `u < Prob[i]` (src/alias_sampler.h:57) is `x < cut point` on
//         the raw 64-bit draw; the high words decide it unless they are equal (2^-32 per draw), and only then does the
//         kernel fetch the low word, from global memory, behind a wave-uniform branch.  8 bytes per entry instead of 12:
//         the pair of HiSeq 2500 / 125 bp profiles (20 218 entries) and GA II / 75 bp fit in LDS, which they did not.
struct IlluminaPacked {
    std::vector<uint64_t> mm2;
    std::vector<uint32_t> tab;
    std::vector<uint32_t> lo;
};
inline IlluminaPacked pack_illumina_tables(const IlluminaTables& T) {
    IlluminaPacked K;
    K.mm2.assign(256, 0);
    for (uint32_t q = 0; q < 256; q++) K.mm2[(q + 33u) & 255u] = T.mm_thresh[q];
    const size_t n_info = T.info.size(), n_ent = T.thresh.size();
    K.tab.resize(n_info * 2 + n_ent * 2);
    K.lo.resize(std::max<size_t>(n_ent, 1));
    for (size_t i = 0; i < n_info; i++) {
        K.tab[2 * i] = (uint32_t)(n_info * 8) + (T.info[i] & 0xffffffu) * 8u;
        K.tab[2 * i + 1] = T.info[i] >> 24;
    }
    uint32_t* ent = K.tab.data() + n_info * 2;
    for (size_t e = 0; e < n_ent; e++) {
        K.lo[e] = (uint32_t)T.thresh[e];
        ent[2 * e] = (uint32_t)(T.thresh[e] >> 32);
        const uint32_t c_self = ((T.quals[e] & 0xffu) + 33u) & 255u, c_alias = ((T.quals[e] >> 8) + 33u) & 255u;
        ent[2 * e + 1] = (c_self * 8u) | ((c_alias * 8u) << 16);
    }
    return K;
}

// The same with 6 bytes per entry, for LDS only (kernel flag E6): [info: {offset of the first cut-point word, entry count |
// offset of the first character pair << 8}][high words of the cut points][{character kept, character of the alias} pairs].
inline IlluminaPacked pack_illumina_tables6(const IlluminaTables& T) {
    IlluminaPacked K;
    K.mm2.assign(256, 0);
    for (uint32_t q = 0; q < 256; q++) K.mm2[(q + 33u) & 255u] = T.mm_thresh[q];
    const size_t n_info = T.info.size(), n_ent = T.thresh.size();
    K.tab.assign(n_info * 2 + n_ent + (n_ent + 1) / 2, 0u);
    K.lo.resize(std::max<size_t>(n_ent, 1));
    const uint32_t th0 = (uint32_t)(n_info * 8), qq0 = th0 + (uint32_t)(n_ent * 4);
    for (size_t i = 0; i < n_info; i++) {
        const uint32_t first = T.info[i] & 0xffffffu;
        K.tab[2 * i] = th0 + first * 4u;
        K.tab[2 * i + 1] = (T.info[i] >> 24) | ((qq0 + first * 2u) << 8);
    }
    uint16_t* qq = reinterpret_cast<uint16_t*>(K.tab.data() + n_info * 2 + n_ent);
    for (size_t e = 0; e < n_ent; e++) {
        K.lo[e] = (uint32_t)T.thresh[e];
        K.tab[n_info * 2 + e] = (uint32_t)(T.thresh[e] >> 32);
        const uint32_t c_self = ((T.quals[e] & 0xffu) + 33u) & 255u, c_alias = ((T.quals[e] >> 8) + 33u) & 255u;
        qq[e] = (uint16_t)(c_self | (c_alias << 8));
    }
    return K;
}

inline void add_profile(IlluminaTables& T, const jk_illumina_profile& pr) {
    const uint32_t L = pr.read_length;
    const size_t info0 = T.info.size();
    T.info.resize(info0 + (size_t)4 * L);
    if (!pr.n_quals || !pr.probs || !pr.quals) throw Error(JK_ERR_ARG, "profile arrays must not be NULL");
    uint64_t off = 0;
    for (uint32_t nt = 0; nt < 4; nt++) {
        for (uint32_t pos = 0; pos < L; pos++) {
            const uint32_t k = pr.n_quals[nt * L + pos];
            if (k == 0 || k > 255) throw Error(JK_ERR_ARG, "each profile position needs between 1 and 255 qualities");
            std::vector<double> p(pr.probs + off, pr.probs + off + k);
            AliasTable at = alias_build(p);
            const uint64_t first = T.thresh.size();
            if (first + k >= (1u << 24)) throw Error(JK_ERR_UNSUPPORTED, "quality profile too large");
            T.info[info0 + (size_t)pos * 4 + nt] = static_cast<uint32_t>(first) | (k << 24);
            for (uint32_t i = 0; i < k; i++) {
                Threshold th = threshold_lt(at.prob[i]);
                const uint8_t q_self = pr.quals[off + i];
                // when every draw picks the entry itself, make the alias point at it too
                const uint8_t q_alias = th.all ? q_self : pr.quals[off + at.alias[i]];
                T.thresh.push_back(th.all ? ~uint64_t(0) : th.th);
                T.quals.push_back(static_cast<uint16_t>(q_self | (q_alias << 8)));
            }
            off += k;
        }
    }
}

inline IlluminaTables build_illumina_tables(const jk_illumina_args& a) {
    IlluminaTables T;
    T.read_length = a.profile1.read_length;
    T.n_ends = a.paired ? 2 : 1;
    if (T.read_length == 0) throw Error(JK_ERR_ARG, "read length must be > 0");
    if (a.paired && a.profile2.read_length != a.profile1.read_length)
        throw Error(JK_ERR_ARG, "In IlluminaOneGenome constr., read lengths for R1 and R2 don't match.");
    add_profile(T, a.profile1);
    if (a.paired) add_profile(T, a.profile2);
    T.mm_thresh.assign(256, 0);
    for (uint32_t q = 0; q < 256; q++) {
        const double prob = (q == 0) ? 1.0 : std::pow(10, static_cast<double>(q) / -10.0);
        Threshold th = threshold_lt(prob);
        T.mm_thresh[q] = th.all ? ~uint64_t(0) : th.th;
    }
    return T;
}

}  // namespace jk
