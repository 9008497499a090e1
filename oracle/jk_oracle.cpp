// =====================================================================================
// TEST INFRASTRUCTURE ONLY.  CPU restatement ("oracle") of jackalope's HTS read-generation
// path.  Nothing under jackalope_amd/ may include, link or call this file: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker.
//
// What it restates (reference = /root/reference, jackalope v1.1.6; file:line cited at each
// function).  It deliberately keeps the reference's *arithmetic types*: x87 `long double`
// for runif_01 (src/pcg.h:99-105) and libstdc++'s own std::gamma_distribution /
// std::binomial_distribution (the reference calls exactly these: src/hts_illumina.h:301,
// src/hts.h:69) driven by a restated pcg64.  It is therefore only meaningful on x86-64 with
// libstdc++/glibc -- the platform the reference's bytes are defined on.
//
// Parity status: the pcg64 restatement is pinned against the reference's own PCG headers
// (oracle/_ref/libref_pcg.so, built by oracle/Makefile from /root/reference/inst/include).
// The rest of the reference path needs Rcpp, RcppArmadillo, RcppProgress and Rhtslib headers
// that this image lacks, so it is unbuildable here; those parts of the oracle are pinned by
// the reference's own known-answer tests (tests/testthat/test-sequencer.R:82-161,
// tests/testthat/test-vcf_IO.R:14-90), restated in tests/.  No golden FASTQ exists in the
// reference (its tests fix no seed), so byte-level parity of sampled quantities is
// "parity unpinned" beyond those tests -- see DESIGN.md.
//
// Third-party arithmetic that is NOT under /root/reference and is restated from its published
// algorithm: Armadillo `accu` (two interleaved accumulators, arrayops::accumulate; version
// unpinned in DESCRIPTION:33) used at src/alias_sampler.h:70.
// =====================================================================================
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <deque>
#include <random>
#include <string>
#include <vector>
#include <zlib.h>
#include <cerrno>
#include <stdexcept>
#include <algorithm>
#include <numeric>
#include <cfloat>
#include <memory>
#include <omp.h>

namespace orc {

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef int64_t s64;
typedef unsigned char u8;   // reference: uint8 = uint_fast8_t = unsigned char on Linux (src/jackalope_types.h:26)

// -------------------------------------------------------------------------------------
// pcg64 = setseq_xsl_rr_128_64 (inst/include/pcg/pcg_random.hpp:1675).
// engine ctor(state, stream): :471-477; bump :387-390; operator() post-advance for 128-bit
// state (output_previous = false) :405-411; XSL-RR output :971-998; default 128-bit
// multiplier :149-152; setseq increment = (stream << 1) | 1.
// -------------------------------------------------------------------------------------
struct Pcg64 {
    typedef u64 result_type;
    u128 state, inc;
    static constexpr u64 min() { return 0; }
    static constexpr u64 max() { return ~u64(0); }
    static u128 mult() { return ((u128)2549297995355413924ULL << 64) | 4865540595714422341ULL; }
    Pcg64() : state(0), inc(1) {}
    Pcg64(u128 seed1, u128 seed2) {
        inc = (seed2 << 1) | 1;
        state = (seed1 + inc) * mult() + inc;
    }
    u64 operator()() {
        state = state * mult() + inc;
        u64 hi = (u64)(state >> 64), lo = (u64)state;
        unsigned rot = (unsigned)(hi >> 58);
        u64 x = hi ^ lo;
        return (x >> rot) | (x << ((64 - rot) & 63));
    }
};

// src/pcg.h:48-61 fill_seeds + :73-84 seeded_pcg(sub_seeds)
static Pcg64 seeded_pcg(const uint32_t* w) {
    u128 a = ((u128)w[0] << 32) + w[1];
    u128 b = ((u128)w[2] << 32) + w[3];
    u128 c = ((u128)w[4] << 32) + w[5];
    u128 d = ((u128)w[6] << 32) + w[7];
    return Pcg64((a << 64) + b, (c << 64) + d);
}

// src/pcg.h:21,99-101.  pcg::max64 is a long double holding 2^64-1; "+ 2" rounds to 2^64 in
// the x87 64-bit significand, exactly as it does in the reference.
static const long double MAX64 = static_cast<long double>(Pcg64::max());
static inline long double runif_01(Pcg64& eng) {
    return (static_cast<long double>(eng()) + 1) / (MAX64 + 2);
}
static inline long double runif_01_from(u64 x) {
    return (static_cast<long double>(x) + 1) / (MAX64 + 2);
}

// Source of the 32-bit sub-seed words the reference pulls from R's RNG
// (Rcpp::runif(8, 0, 4294967296) truncated; src/pcg.h:37-46,63-71).
struct SeedSource {
    const uint32_t* w; u64 n; u64 pos;
    const uint32_t* take8() {
        if (pos + 8 > n) throw std::runtime_error("oracle: seed words exhausted");
        const uint32_t* r = w + pos; pos += 8; return r;
    }
};

// -------------------------------------------------------------------------------------
// AliasSampler (src/alias_sampler.h:36-106).  arma::accu restated as Armadillo's
// arrayops::accumulate (pairwise-interleaved two accumulators); `p /= s; p *= n` element-wise.
// -------------------------------------------------------------------------------------
static double arma_accu(const std::vector<double>& v) {
    double acc1 = 0, acc2 = 0;
    size_t n = v.size(), j, i = 0;
    for (j = 1; j < n; j += 2) { acc1 += v[i++]; acc2 += v[i++]; }
    if ((j - 1) < n) acc1 += v[i];
    return acc1 + acc2;
}

struct Alias {
    std::vector<double> Prob;
    std::vector<u64> Ali;
    u64 n;
    Alias() : n(0) {}
    explicit Alias(std::vector<double> p) : Prob(p.size()), Ali(p.size()), n(p.size()) {
        double s = arma_accu(p);
        for (double& x : p) x /= s;
        for (double& x : p) x *= static_cast<double>(n);
        std::deque<u64> Small, Large;
        for (u64 i = 0; i < n; i++) { if (p[i] < 1) Small.push_back(i); else Large.push_back(i); }
        while (!Small.empty() && !Large.empty()) {
            u64 l = Small.front(); Small.pop_front();
            u64 g = Large.front(); Large.pop_front();
            Prob[l] = p[l];
            Ali[l] = g;
            p[g] = (p[g] + p[l]) - 1;
            if (p[g] < 1) Small.push_back(g); else Large.push_back(g);
        }
        while (!Large.empty()) { Prob[Large.front()] = 1; Large.pop_front(); }
        while (!Small.empty()) { Prob[Small.front()] = 1; Small.pop_front(); }
    }
    // src/alias_sampler.h:53-60
    u64 sample(Pcg64& eng) const {
        u64 i = runif_01(eng) * n;
        double u = runif_01(eng);
        if (u < Prob[i]) return i;
        return Ali[i];
    }
};

// -------------------------------------------------------------------------------------
// Tables used by the sequencers (src/hts.h:36-46, src/jackalope_types.h:36,
// src/str_manip.h:58-72).
// -------------------------------------------------------------------------------------
static inline u8 nt_index(char c) {
    switch (c) { case 'T': return 0; case 'C': return 1; case 'A': return 2; case 'G': return 3; default: return 4; }
}
static const char* const MM_NUCLEOS[5] = {"CAG", "TAG", "TCG", "TCA", "NNN"};
static const std::string BASES = "TCAG";
static inline char cmp_char(char c) {
    switch (c) { case 'T': return 'A'; case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C';
                 case 'N': return 'N'; default: return 0; }
}
// src/str_manip.h:214-229
static void rev_comp(std::string& s) {
    u64 n = s.size(), half = n / 2;
    for (u64 j = 0; j < half; j++) {
        char tmp = cmp_char(s[j]);
        s[j] = cmp_char(s[n - j - 1]);
        s[n - j - 1] = tmp;
    }
    if (n & 1ULL) s[half] = cmp_char(s[half]);
}
// src/str_manip.h:234-248 (first n chars only; PacBio)
static void rev_comp_n(std::string& s, u64 n) {
    u64 half = n / 2;
    for (u64 j = 0; j < half; j++) {
        char tmp = cmp_char(s[j]);
        s[j] = cmp_char(s[n - j - 1]);
        s[n - j - 1] = tmp;
    }
    if (n & 1ULL) s[half] = cmp_char(s[half]);
}

// src/util.h:245-258
static std::vector<u64> split_int(u64 x, u64 n) {
    std::vector<u64> out(n, x / n);
    u64 sum = n * (x / n), i = 0;
    while (sum < x) { out[i]++; i++; sum++; }
    return out;
}

// src/hts.h:58-103.  A fresh engine is seeded from 8 sub-seed words whenever n_reads > 0 and
// probs is non-empty, even if it is then never used (probs.size() == 1).
static std::vector<u64> reads_per_group(u64 n_reads, std::vector<double> probs, SeedSource& seeds) {
    std::vector<u64> out(probs.size(), 0);
    if (n_reads == 0 || probs.size() == 0) return out;
    Pcg64 eng = seeded_pcg(seeds.take8());
    double sum_probs = std::accumulate(probs.begin(), probs.end(), 0.0);
    for (double& p : probs) p /= sum_probs;
    std::binomial_distribution<u64> distr(n_reads, 0.5);
    for (u64 i = 0; i < (probs.size() - 1); i++) {
        if (probs[i] >= 1) { out[i] = n_reads; return out; }
        if (probs[i] == 0) continue;
        distr.param(std::binomial_distribution<u64>::param_type(n_reads, probs[i]));
        out[i] = distr(eng);
        n_reads -= out[i];
        if (n_reads == 0) break;
        sum_probs = 1 - probs[i];
        for (u64 j = i + 1; j < probs.size(); j++) probs[j] /= sum_probs;
    }
    out.back() = n_reads;
    return out;
}

// -------------------------------------------------------------------------------------
// Illumina quality/mismatch model: IllQualPos + IlluminaQualityError
// (src/hts_illumina.h:93-277).
// -------------------------------------------------------------------------------------
struct Profile {              // one read end: [nt 0..3][pos][k]
    std::vector<std::vector<std::vector<double>>> probs;
    std::vector<std::vector<std::vector<u8>>> quals;
};

struct QualErr {
    std::vector<std::vector<Alias>> samplers;        // [nt][pos]
    std::vector<std::vector<std::vector<u8>>> quals;  // [nt][pos][k]
    std::vector<double> qual_prob_map;
    QualErr() {}
    explicit QualErr(const Profile& pr) : quals(pr.quals) {
        if (pr.probs.size() != 4 || pr.quals.size() != 4) throw std::runtime_error("profile must have 4 nucleotides");
        u64 L = pr.probs[0].size();
        u8 max_qual = 0;
        samplers.resize(4);
        for (int nt = 0; nt < 4; nt++) {
            if (pr.probs[nt].size() != L || pr.quals[nt].size() != L) throw std::runtime_error("profile lengths differ");
            for (u64 pos = 0; pos < L; pos++) samplers[nt].push_back(Alias(pr.probs[nt][pos]));
            for (const auto& qv : pr.quals[nt]) {
                u8 m = *std::max_element(qv.begin(), qv.end());
                if (m > max_qual) max_qual = m;
            }
        }
        // src/hts_illumina.h:182-187
        qual_prob_map.push_back(1);
        for (u64 q = 1; q < (static_cast<u64>(max_qual) + 1ULL); q++)
            qual_prob_map.push_back(std::pow(10, static_cast<double>(q) / -10.0));
    }

    // src/hts_illumina.h:202-259
    void fill_read_qual(std::string& read, std::string& qual, std::deque<u64>& insertions,
                        std::deque<u64>& deletions, Pcg64& eng) const {
        const u8 qual_start = static_cast<u8>('!');
        u64 chrom_pos = read.size() - 1ULL;
        while (!insertions.empty() || !deletions.empty()) {
            if (!insertions.empty() && chrom_pos == insertions.back()) {
                char c = BASES[static_cast<u64>(runif_01(eng) * 4.0)];
                read.insert(chrom_pos + 1, 1, c);
                insertions.pop_back();
            } else if (!deletions.empty() && chrom_pos == deletions.back()) {
                read.erase(chrom_pos, 1);
                deletions.pop_back();
            }
            if (chrom_pos == 0) break;
            chrom_pos--;
        }
        if (qual.size() != read.size()) qual.resize(read.size());
        for (u64 pos = 0; pos < read.size(); pos++) {
            char& nt = read[pos];
            u8 nt_ind = nt_index(nt);
            u8 qint;
            if (nt_ind > 3) {
                qint = runif_01(eng) * 10 + qual_start;
                qual[pos] = static_cast<char>(qint);
                nt = 'N';
                continue;
            }
            u64 k = samplers[nt_ind][pos].sample(eng);
            qint = quals[nt_ind][pos][k];
            double mis_prob = qual_prob_map[qint];
            qint += qual_start;
            qual[pos] = static_cast<char>(qint);
            double u = runif_01(eng);
            if (u < mis_prob) {
                const char* mm = MM_NUCLEOS[nt_ind];
                nt = mm[static_cast<u64>(runif_01(eng) * 3.0)];
            }
        }
    }
};

// -------------------------------------------------------------------------------------
// Genome views.  A "genome" for the sequencer is a name plus named chromosome strings
// (RefGenome: src/ref_classes.h:127-180; haplotype chromosomes are materialised by
// HapChrom::get_chrom_full first, exactly as IlluminaHaplotypes::one_read does,
// src/hts_illumina.cpp:527).
// -------------------------------------------------------------------------------------
struct Genome {
    std::string name;
    std::vector<std::string> chrom_names;
    std::vector<const std::string*> chroms;   // may be filled lazily for haplotypes
    std::vector<u64> chrom_sizes;
};

// src/ref_classes.h:102-116 / src/hts.h:109-130
static void fill_read(const std::string& chrom, std::string& read, u64 read_start, u64 chrom_start, u64 n_to_add) {
    if ((chrom_start + n_to_add - 1) >= chrom.size()) n_to_add = chrom.size() - chrom_start;
    if (read.size() < n_to_add + read_start) read.resize(n_to_add + read_start, 'N');
    for (u64 i = 0; i < n_to_add; i++) read[read_start + i] = chrom[chrom_start + i];
}

// src/hts_illumina.cpp:286-326
static void fill_fq_lines(std::vector<char>& pool, const std::string& name, const std::string& chrom_name,
                          const std::string& read, const std::string& qual, u64 i, u64 start, bool paired,
                          bool& reverse) {
    pool.push_back('@');
    for (char c : name) pool.push_back(c);
    pool.push_back('-');
    for (char c : chrom_name) pool.push_back(c);
    pool.push_back('-');
    for (char c : std::to_string(start)) pool.push_back(c);
    pool.push_back('-');
    pool.push_back(reverse ? 'R' : 'F');
    if (paired) { pool.push_back('/'); for (char c : std::to_string(i + 1)) pool.push_back(c); }
    pool.push_back('\n');
    for (char c : read) pool.push_back(c);
    pool.push_back('\n'); pool.push_back('+'); pool.push_back('\n');
    for (char c : qual) pool.push_back(c);
    pool.push_back('\n');
    reverse = !reverse;
}

// -------------------------------------------------------------------------------------
// IlluminaOneGenome<T> (src/hts_illumina.h:293-497, src/hts_illumina.cpp:33-482).
// One object per reference "thread" (= lane); the gamma distribution object (with libstdc++'s
// saved normal deviate) lives inside it, as in the reference.
// -------------------------------------------------------------------------------------
struct IlluminaParams {
    bool paired, matepair;
    double shape, scale;
    u64 frag_len_min, frag_len_max;
    Profile prof[2];
    double ins_prob[2], del_prob[2];
};

struct IlluminaOneGenome {
    // The reference copies these tables into every per-thread filler; they are never modified after
    // construction, so the copies share one instance here (1 M lanes x 300 KB would not fit in RAM).
    std::vector<std::shared_ptr<const QualErr>> qual_errors;
    std::gamma_distribution<double> frag_lengths;
    std::vector<u64> chrom_reads;
    const Genome* genome;
    u64 read_length;
    bool paired, matepair;
    std::vector<double> ins_probs, del_probs;
    std::vector<std::deque<u64>> insertions, deletions;
    u64 frag_len_min, frag_len_max;
    // IlluminaReadConstrInfo (src/hts_illumina.h:45-84)
    u64 chrom_ind, frag_len, frag_start;
    std::vector<std::string> reads, quals;
    std::vector<u64> read_chrom_spaces;
    std::string barcode;

    IlluminaOneGenome(const Genome& g, const IlluminaParams& p, const std::string& barcode_)
        : frag_lengths(p.shape, p.scale), genome(&g), read_length(p.prof[0].probs[0].size()),
          paired(p.paired), matepair(p.paired ? p.matepair : false),
          frag_len_min(p.frag_len_min), frag_len_max(p.frag_len_max),
          chrom_ind(0), frag_len(0), frag_start(0), barcode(barcode_) {
        u64 ne = paired ? 2 : 1;
        if (paired && p.prof[0].probs[0].size() != p.prof[1].probs[0].size())
            throw std::runtime_error("In IlluminaOneGenome constr., read lengths for R1 and R2 don't match.");
        for (u64 r = 0; r < ne; r++) {
            qual_errors.push_back(std::make_shared<const QualErr>(p.prof[r]));
            ins_probs.push_back(p.ins_prob[r]);
            del_probs.push_back(p.del_prob[r]);
        }
        insertions.resize(ne); deletions.resize(ne);
        reads.assign(ne, std::string(read_length, 'N'));
        quals.assign(ne, std::string());
        read_chrom_spaces.assign(ne, 0);
    }

    void reset_quota() { chrom_reads.clear(); }
    // src/hts_illumina.h:410-418
    void add_n_reads(u64 n_reads, SeedSource& seeds) {
        std::vector<double> probs_(genome->chrom_sizes.begin(), genome->chrom_sizes.end());
        if (paired) n_reads /= 2;
        chrom_reads = reads_per_group(n_reads, probs_, seeds);
        if (paired) for (u64& r : chrom_reads) r *= 2;
    }

    // src/hts_illumina.cpp:117-150
    void sample_indels(Pcg64& eng) {
        for (u64 r = 0; r < insertions.size(); r++) {
            u64 frag_pos = 0, length_now = 0;
            std::deque<u64>& ins = insertions[r];
            std::deque<u64>& del = deletions[r];
            const double ins_prob = ins_probs[r], del_prob = del_probs[r];
            ins.clear(); del.clear();
            while (length_now < read_length && frag_pos < frag_len) {
                double u = runif_01(eng);
                if (u > (ins_prob + del_prob)) {
                    length_now++;
                } else if (u > ins_prob) {
                    del.push_back(frag_pos);
                } else {
                    if (length_now == (read_length - 1)) length_now++;
                    else { ins.push_back(frag_pos); length_now += 2; }
                }
                frag_pos++;
            }
        }
    }
    // src/hts_illumina.cpp:154-184
    void adjust_chrom_spaces() {
        for (u64 r = 0; r < insertions.size(); r++) {
            s64 indel_effect = static_cast<s64>(deletions[r].size()) - static_cast<s64>(insertions[r].size());
            read_chrom_spaces[r] = std::min(read_length + indel_effect, frag_len);
            if (reads[r].size() != read_chrom_spaces[r]) reads[r].resize(read_chrom_spaces[r], 'N');
            read_chrom_spaces[r] -= barcode.size();
        }
    }
    // src/hts_illumina.cpp:236-265 (and the body of :192-226 after the chromosome pick)
    void indels_frag(Pcg64& eng) {
        u64 chrom_len = genome->chrom_sizes[chrom_ind];
        frag_len = static_cast<u64>(frag_lengths(eng));
        if (frag_len < frag_len_min) frag_len = frag_len_min;
        if (frag_len > frag_len_max) frag_len = frag_len_max;
        if (frag_len >= chrom_len) {
            frag_len = chrom_len;
            frag_start = 0;
        } else {
            double u = runif_01(eng);
            frag_start = static_cast<u64>(u * (chrom_len - frag_len + 1));
        }
        sample_indels(eng);
        adjust_chrom_spaces();
    }
    // src/hts_illumina.cpp:192-226
    void chrom_indels_frag(Pcg64& eng) {
        chrom_ind = 0;
        while (chrom_ind < chrom_reads.size() && chrom_reads[chrom_ind] == 0) chrom_ind++;
        if (chrom_ind == genome->chroms.size()) return;
        indels_frag(eng);
    }
    // src/hts_illumina.cpp:273-282
    void just_indels(Pcg64& eng) { sample_indels(eng); adjust_chrom_spaces(); }

    // src/hts_illumina.cpp:339-409 (ref_variant = true) and :414-482 (string overload)
    void append_pools(const std::string& chrom, std::vector<std::vector<char>>& pools, Pcg64& eng,
                      bool ref_variant) {
        u64 n_read_ends = ins_probs.size();
        if (pools.size() != n_read_ends) pools.resize(n_read_ends);
        bool reverse = runif_01(eng) < 0.5;
        for (u64 i = 0; i < n_read_ends; i++) {
            std::string& read = reads[i];
            std::string& qual = quals[i];
            u64 start;
            if ((!matepair && !reverse) || (matepair && reverse)) start = frag_start;
            else start = frag_start + frag_len - read_chrom_spaces[i];
            if (!reverse) {
                fill_read(chrom, read, barcode.size(), start, read_chrom_spaces[i]);
            } else {
                fill_read(chrom, read, 0, start, read_chrom_spaces[i]);
                rev_comp(read);
            }
            for (u64 b = 0; b < barcode.size(); b++) read[b] = barcode[b];
            qual_errors[i]->fill_read_qual(read, qual, insertions[i], deletions[i], eng);
            fill_fq_lines(pools[i], genome->name, genome->chrom_names[chrom_ind], read, qual, i, start,
                          paired, reverse);
        }
        if (ref_variant) {
            if (chrom_reads[chrom_ind] < n_read_ends) chrom_reads[chrom_ind] = 0;
            else chrom_reads[chrom_ind] -= n_read_ends;
        }
    }

    // src/hts_illumina.cpp:35-53 / :82-93
    void one_read(std::vector<std::vector<char>>& pools, bool& finished, Pcg64& eng) {
        chrom_indels_frag(eng);
        if (chrom_ind == genome->chroms.size()) { finished = true; return; }
        append_pools(*genome->chroms[chrom_ind], pools, eng, true);
    }
    void re_read(std::vector<std::vector<char>>& pools, bool& finished, Pcg64& eng) {
        (void)finished;
        just_indels(eng);
        append_pools(*genome->chroms[chrom_ind], pools, eng, true);
    }
    // string overloads used by the haplotype path: src/hts_illumina.cpp:58-74 / :96-110
    void one_read_str(const std::string& chrom, u64 chrom_i, std::vector<std::vector<char>>& pools, Pcg64& eng) {
        chrom_ind = chrom_i;
        indels_frag(eng);
        append_pools(chrom, pools, eng, false);
    }
    void re_read_str(const std::string& chrom, u64 chrom_i, std::vector<std::vector<char>>& pools, Pcg64& eng) {
        chrom_ind = chrom_i;
        just_indels(eng);
        append_pools(chrom, pools, eng, false);
    }
};

// -------------------------------------------------------------------------------------
// Haplotypes: mutation table read side (src/hap_classes.h:100-258,314-333,439-455;
// src/hap_classes.cpp:80-116).
// -------------------------------------------------------------------------------------
struct HapChrom {
    const std::string* ref;        // reference chromosome
    std::vector<u64> old_pos, new_pos;
    std::vector<std::string> nucleos;   // "" for a deletion (reference: nullptr)
    u64 chrom_size;
    std::string name;

    s64 size_modifier(u64 ind) const {
        s64 size_mod;
        if (ind < (new_pos.size() - 1)) size_mod = new_pos[ind + 1] - old_pos[ind + 1];
        else size_mod = chrom_size - ref->size();
        size_mod += static_cast<s64>(old_pos[ind] - new_pos[ind]);
        return size_mod;
    }
    char get_char_(u64 pos, u64 mut_i) const {
        u64 ind = pos - new_pos[mut_i];
        if (static_cast<s64>(ind) > size_modifier(mut_i)) {
            ind += (old_pos[mut_i] - size_modifier(mut_i));
            return (*ref)[ind];
        }
        if (nucleos[mut_i].empty()) throw std::runtime_error("mutations.nucleos[mut_i] == nullptr");
        return nucleos[mut_i][ind];
    }
    std::string get_chrom_full() const {
        if (new_pos.empty()) return *ref;
        u64 mut_i = 0, pos = 0;
        std::string out; out.reserve(chrom_size);
        while (pos < new_pos[mut_i]) { out.push_back((*ref)[pos]); ++pos; }
        u64 next_mut_i = mut_i + 1;
        while (next_mut_i < new_pos.size()) {
            while (pos < new_pos[next_mut_i]) { out.push_back(get_char_(pos, mut_i)); ++pos; }
            ++mut_i; ++next_mut_i;
        }
        while (pos < chrom_size) { out.push_back(get_char_(pos, mut_i)); ++pos; }
        return out;
    }
};

struct HapGenome { std::string name; std::vector<HapChrom> chroms; };

// Test-speed option (orc_set_chrom_cache): the reference materialises a haplotype chromosome with get_chrom_full()
// every time a thread's cursor enters a (haplotype, chromosome) cell.  The string is a pure function of the cell,
// so for genome-scale windows (24 x 125 Mbp x 4-8 haplotypes per thread) the cells are materialised once, in
// parallel, and the threads read the shared copies.  Same bytes, fewer repetitions.
struct ChromCache {
    u64 n_chroms = 0;
    std::vector<std::string> seq;                 // [hap * n_chroms + chr]
    void build(const std::vector<HapGenome>& haps) {
        n_chroms = haps.empty() ? 0 : haps[0].chroms.size();
        seq.assign(haps.size() * n_chroms, std::string());
        const long n = (long)seq.size();
#pragma omp parallel for schedule(dynamic, 1)
        for (long k = 0; k < n; k++) seq[k] = haps[k / n_chroms].chroms[k % n_chroms].get_chrom_full();
    }
};
static bool g_use_chrom_cache = false;
static const ChromCache* g_chrom_cache = nullptr;   // set for the duration of one orc_*_hap call

// IlluminaHaplotypes (src/hts_illumina.h:509-675, src/hts_illumina.cpp:495-558)
struct IlluminaHaplotypes {
    const std::vector<HapGenome>* haps;
    std::vector<Genome> genomes;                 // name/chrom-name/size views per haplotype
    std::vector<std::vector<u64>> n_reads_vc;
    std::vector<IlluminaOneGenome> read_makers;
    bool paired;
    std::vector<double> hap_probs;
    u64 hap, chr;
    std::string hap_chrom_seq;
    const std::string* cached_seq = nullptr;     // ChromCache entry standing in for hap_chrom_seq
    const std::string& cur_seq() const { return cached_seq ? *cached_seq : hap_chrom_seq; }

    IlluminaHaplotypes(const std::vector<HapGenome>& hs, const std::vector<double>& probs,
                       const IlluminaParams& p, std::vector<std::string> barcodes)
        : haps(&hs), paired(p.paired), hap_probs(probs), hap(0), chr(0) {
        if (barcodes.size() < hs.size()) barcodes.resize(hs.size(), "");
        genomes.resize(hs.size());
        for (u64 i = 0; i < hs.size(); i++) {
            genomes[i].name = hs[i].name;
            for (const HapChrom& hc : hs[i].chroms) {
                genomes[i].chrom_names.push_back(hc.name);
                genomes[i].chroms.push_back(nullptr);
                genomes[i].chrom_sizes.push_back(hc.chrom_size);
            }
        }
        read_makers.reserve(hs.size());
        for (u64 i = 0; i < hs.size(); i++) read_makers.push_back(IlluminaOneGenome(genomes[i], p, barcodes[i]));
    }
    // The copy made per thread must re-point each read maker at this object's genome views.
    IlluminaHaplotypes(const IlluminaHaplotypes& o)
        : haps(o.haps), genomes(o.genomes), n_reads_vc(o.n_reads_vc), read_makers(o.read_makers),
          paired(o.paired), hap_probs(o.hap_probs), hap(o.hap), chr(o.chr), hap_chrom_seq(o.hap_chrom_seq), cached_seq(o.cached_seq) {
        for (u64 i = 0; i < read_makers.size(); i++) read_makers[i].genome = &genomes[i];
    }

    void reset_quota() { n_reads_vc.clear(); for (auto& rm : read_makers) rm.reset_quota(); }
    // src/hts_illumina.h:620-644
    void add_n_reads(u64 n_reads, SeedSource& seeds) {
        u64 n_haps = haps->size();
        if (paired) n_reads /= 2;
        std::vector<u64> hap_reads = reads_per_group(n_reads, hap_probs, seeds);
        for (u64 v = 0; v < n_haps; v++) {
            std::vector<double> chrom_probs;
            for (const HapChrom& vc : (*haps)[v].chroms) chrom_probs.push_back(vc.chrom_size);
            n_reads_vc.push_back(reads_per_group(hap_reads[v], chrom_probs, seeds));
            if (paired) for (u64& r : n_reads_vc.back()) r *= 2;
        }
        for (u64 i = 0; i < n_haps; i++) read_makers[i].add_n_reads(hap_reads[i], seeds);
    }
    // src/hts_illumina.cpp:495-536
    void one_read(std::vector<std::vector<char>>& pools, bool& finished, Pcg64& eng) {
        if (hap == haps->size()) { finished = true; return; }
        if (n_reads_vc[hap][chr] == 0 || cur_seq().empty()) {
            u64 new_hap = hap, new_chr = chr;
            for (; new_hap < n_reads_vc.size(); new_hap++) {
                while (n_reads_vc[new_hap][new_chr] == 0) {
                    new_chr++;
                    if (new_chr == n_reads_vc[new_hap].size()) break;
                }
                if (new_chr < n_reads_vc[new_hap].size()) break;
                else new_chr = 0;
            }
            hap = new_hap; chr = new_chr;
            if (hap == haps->size()) { finished = true; return; }
            if (g_chrom_cache) cached_seq = &g_chrom_cache->seq[hap * g_chrom_cache->n_chroms + chr];
            else hap_chrom_seq = (*haps)[hap].chroms[chr].get_chrom_full();
        }
        read_makers[hap].one_read_str(cur_seq(), chr, pools, eng);
        n_reads_vc[hap][chr]--;
        if (paired && n_reads_vc[hap][chr] > 0) n_reads_vc[hap][chr]--;
    }
    // src/hts_illumina.cpp:542-558
    void re_read(std::vector<std::vector<char>>& pools, bool& finished, Pcg64& eng) {
        if (hap == haps->size()) { finished = true; return; }
        read_makers[hap].re_read_str(cur_seq(), chr, pools, eng);
        if (n_reads_vc[hap][chr] > 0) n_reads_vc[hap][chr]--;
        if (paired && n_reads_vc[hap][chr] > 0) n_reads_vc[hap][chr]--;
    }
};

// -------------------------------------------------------------------------------------
// Driver: ReadWriterOneThread::create_reads + write_reads_one_filetype_ (src/hts.h:193-429)
// restated for T sequential "threads".  Output = thread 0's pools, then thread 1's, ...
// (one of the interleavings the reference's `omp critical` flush can produce).
// -------------------------------------------------------------------------------------
struct RunOpts {
    u64 thread_begin = 0, thread_end = 0;   // only these threads generate (0,0 = all); seeds/quotas are
                                            // still derived for every thread, so the kept ones are unchanged
    std::vector<std::pair<u64, u64>> windows;   // several such windows in one run (orc_set_windows); output in window order
    bool discard = false;                   // count bytes only (null sink, for timing)
    std::vector<std::vector<u64>> thread_bytes;   // out: [end][thread]
};

template <typename Filler>
static void run_threads(const Filler& base, u64 n_reads, double prob_dup, u64 read_pool_size, u64 n_read_ends,
                        u64 n_threads, SeedSource& seeds, std::vector<std::vector<char>>& files, RunOpts& opts) {
    n_reads /= n_read_ends;
    std::vector<u64> reads_per_thread = split_int(n_reads, n_threads);
    for (u64& i : reads_per_thread) i *= n_read_ends;
    // mt_seeds first (src/hts.h:339), then one filler copy + add_n_reads per thread (:349-353)
    std::vector<const uint32_t*> tseeds(n_threads);
    for (u64 t = 0; t < n_threads; t++) tseeds[t] = seeds.take8();
    // Threads outside the generation window only need their seed words consumed: run add_n_reads on one
    // scratch copy for them instead of keeping a filler each.
    std::vector<std::pair<u64, u64>> wins = opts.windows;
    if (wins.empty()) wins.push_back({opts.thread_begin, opts.thread_end ? opts.thread_end : n_threads});
    std::vector<u64> gen;                    // the threads that generate, in output order
    for (auto& w : wins) {
        if (w.second > n_threads) w.second = n_threads;
        if (!gen.empty() && w.first < gen.back() + 1) throw std::runtime_error("thread windows must be increasing and disjoint");
        for (u64 t = w.first; t < w.second; t++) gen.push_back(t);
    }
    std::vector<Filler> fillers;
    fillers.reserve(gen.size());
    Filler scratch(base);
    {
        size_t gi = 0;
        for (u64 t = 0; t < n_threads; t++) {
            if (gi < gen.size() && gen[gi] == t) {
                fillers.push_back(base);
                fillers.back().add_n_reads(reads_per_thread[t], seeds);
                gi++;
            } else {
                scratch.reset_quota();
                scratch.add_n_reads(reads_per_thread[t], seeds);
            }
        }
    }
    files.assign(n_read_ends, std::vector<char>());
    // Threads are independent (own filler copy, own engine), so they may run concurrently; each
    // keeps its own output and the pieces are concatenated in thread order afterwards.
    std::vector<std::vector<std::vector<char>>> outs(gen.size(), std::vector<std::vector<char>>(n_read_ends));
    opts.thread_bytes.assign(n_read_ends, std::vector<u64>(n_threads, 0));
#pragma omp parallel for schedule(dynamic, 1)
    for (u64 gk = 0; gk < gen.size(); gk++) {
        const u64 t = gen[gk];
        std::vector<std::vector<char>>& files = outs[gk];
        Pcg64 eng = seeded_pcg(tseeds[t]);
        Filler& filler = fillers[gk];
        const u64 n = reads_per_thread[t];
        u64 reads_made = 0, reads_in_pool = 0;
        std::vector<std::vector<char>> pools(n_read_ends);
        while (reads_made < n) {
            bool do_write;
            bool finished = false;
            filler.one_read(pools, finished, eng);
            if (finished) { reads_made = n; do_write = true; }
            else {
                reads_made += n_read_ends; reads_in_pool += n_read_ends;
                double dup = runif_01(eng);
                bool fin2 = false;
                while (dup < prob_dup && reads_made < n && reads_in_pool < read_pool_size) {
                    filler.re_read(pools, finished, eng);
                    if (finished) { reads_made = n; fin2 = true; break; }
                    reads_made += n_read_ends; reads_in_pool += n_read_ends;
                    dup = runif_01(eng);
                }
                do_write = fin2 || reads_in_pool >= read_pool_size || reads_made >= n;
            }
            if (do_write) {
                for (u64 i = 0; i < pools.size(); i++) {
                    opts.thread_bytes[i][t] += pools[i].size();
                    if (!opts.discard) files[i].insert(files[i].end(), pools[i].begin(), pools[i].end());
                    pools[i].clear();
                }
                reads_in_pool = 0;
            }
        }
    }
    for (u64 i = 0; i < n_read_ends; i++) {
        size_t total = 0;
        for (u64 t = 0; t < outs.size(); t++) total += outs[t][i].size();
        files[i].reserve(total);
        for (u64 t = 0; t < outs.size(); t++) {
            files[i].insert(files[i].end(), outs[t][i].begin(), outs[t][i].end());
            std::vector<char>().swap(outs[t][i]);
        }
    }
}

// helpers to unflatten C inputs
static Profile make_profile(uint32_t L, const uint32_t* n_quals, const double* probs, const uint8_t* quals) {
    Profile p; p.probs.resize(4); p.quals.resize(4);
    u64 off = 0;
    for (int nt = 0; nt < 4; nt++) {
        p.probs[nt].resize(L); p.quals[nt].resize(L);
        for (uint32_t pos = 0; pos < L; pos++) {
            uint32_t k = n_quals[nt * L + pos];
            p.probs[nt][pos].assign(probs + off, probs + off + k);
            p.quals[nt][pos].assign(quals + off, quals + off + k);
            off += k;
        }
    }
    return p;
}

static void give(const std::vector<char>& v, char** out, uint64_t* len) {
    *len = v.size();
    *out = static_cast<char*>(std::malloc(v.size() ? v.size() : 1));
    if (!v.empty()) std::memcpy(*out, v.data(), v.size());
}

// -------------------------------------------------------------------------------------
// R nmath pieces used by the PacBio path (src/hts_pacbio.h:178,349,352).  R is NOT under
// /root/reference (DESCRIPTION:26 "R (>= 2.10)", unpinned) and is absent from this image: these are
// restatements of the published algorithms R implements -- qnorm5: Wichura (1988) AS 241 PPND16;
// pnorm5: Cody (1969) -- and their agreement with R itself is UNPINNED (checked numerically against
// scipy in tests, not bit for bit).
// -------------------------------------------------------------------------------------
static double qnorm(double p) {
    if (!(p > 0.0)) return -INFINITY;
    if (!(p < 1.0)) return INFINITY;
    const double q = p - 0.5;
    double r, val;
    if (std::fabs(q) <= 0.425) {
        r = .180625 - q * q;
        val = q * (((((((r * 2509.0809287301226727 + 33430.575583588128105) * r + 67265.770927008700853) * r
                      + 45921.953931549871457) * r + 13731.693765509461125) * r + 1971.5909503065514427) * r
                    + 133.14166789178437745) * r + 3.387132872796366608)
              / (((((((r * 5226.495278852545925 + 28729.085735721942674) * r + 39307.89580009271061) * r
                     + 21213.794301586595867) * r + 5394.1960214247511077) * r + 687.1870074920579083) * r
                   + 42.313330701600911252) * r + 1.);
        return val;
    }
    r = (q < 0) ? p : (0.5 - p + 0.5);
    r = std::sqrt(-std::log(r));
    if (r <= 5.) {
        r += -1.6;
        val = (((((((r * 7.7454501427834140764e-4 + .0227238449892691845833) * r + .24178072517745061177) * r
                   + 1.27045825245236838258) * r + 3.64784832476320460504) * r + 5.7694972214606914055) * r
                + 4.6303378461565452959) * r + 1.42343711074968357734)
              / (((((((r * 1.05075007164441684324e-9 + 5.475938084995344946e-4) * r + .0151986665636164571966) * r
                     + .14810397642748007459) * r + .68976733498510000455) * r + 1.6763848301838038494) * r
                  + 2.05319162663775882187) * r + 1.);
    } else {
        r += -5.;
        val = (((((((r * 2.01033439929228813265e-7 + 2.71155556874348757815e-5) * r + .0012426609473880784386) * r
                   + .026532189526576123093) * r + .29656057182850489123) * r + 1.7848265399172913358) * r
                + 5.4637849111641143699) * r + 6.6579046435011037772)
              / (((((((r * 2.04426310338993978564e-15 + 1.4215117583164458887e-7) * r + 1.8463183175100546818e-5) * r
                     + 7.868691311456132591e-4) * r + .0148753612908506148525) * r + .13692988092273580531) * r
                  + .59983220655588793769) * r + 1.);
    }
    return (q < 0.0) ? -val : val;
}

// pnorm5(x, 0, 1, lower, !log) after Cody (1969) and qchisq(p, df) by Newton on the regularised incomplete
// gamma function -- the same restatements the product's host set-up uses (jackalope_amd/csrc/jk_nmath.h);
// they only produce per-run tables.  Parity with R's nmath is UNPINNED (see above).
static double pnorm_std(double x) {          // P[N(0,1) <= x]
    static const double a[5] = {2.2352520354606839287, 161.02823106855587881, 1067.6894854603709582,
                                18154.981253343561249, 0.065682337918207449113};
    static const double b[4] = {47.20258190468824187, 976.09855173777669322, 10260.932208618978205,
                                45507.789335026729956};
    static const double c[9] = {0.39894151208813466764, 8.8831497943883759412, 93.506656132177855979,
                                597.27027639480026226, 2494.5375852903726711, 6848.1904505362823326,
                                11602.651437647350124, 9842.7148383839780218, 1.0765576773720192317e-8};
    static const double d[8] = {22.266688044328115691, 235.38790178262499861, 1519.377599407554805,
                                6485.558298266760755, 18615.571640885098091, 34900.952721145977266,
                                38912.003286093271411, 19685.429676859990727};
    static const double p[6] = {0.21589853405795699, 0.1274011611602473639, 0.022235277870649807,
                                0.001421619193227893466, 2.9112874951168792e-5, 0.02307344176494017303};
    static const double q[5] = {1.28426009614491121, 0.468238212480865118, 0.0659881378689285515,
                                0.00378239633202758244, 7.29751555083966205e-5};
    const double eps = DBL_EPSILON * 0.5, y = std::fabs(x);
    double xden, xnum, temp, del, xsq, cum, ccum;
    if (std::isnan(x)) return x;
    if (y <= 0.67448975) {
        if (y > eps) {
            xsq = x * x; xnum = a[4] * xsq; xden = xsq;
            for (int i = 0; i < 3; ++i) { xnum = (xnum + a[i]) * xsq; xden = (xden + b[i]) * xsq; }
        } else xnum = xden = 0.0;
        temp = x * (xnum + a[3]) / (xden + b[3]);
        return 0.5 + temp;
    }
    if (y <= 5.656854249492380195206754896838 /* sqrt(32) */) {
        xnum = c[8] * y; xden = y;
        for (int i = 0; i < 7; ++i) { xnum = (xnum + c[i]) * y; xden = (xden + d[i]) * y; }
        temp = (xnum + c[7]) / (xden + d[7]);
        xsq = std::trunc(y * 16) / 16; del = (y - xsq) * (y + xsq);
        cum = std::exp(-xsq * xsq * 0.5) * std::exp(-del * 0.5) * temp; ccum = 1.0 - cum;
        return x > 0. ? ccum : cum;
    }
    if (x > -37.5193 && x < 8.2924) {
        xsq = 1.0 / (x * x);
        xnum = p[5] * xsq; xden = xsq;
        for (int i = 0; i < 4; ++i) { xnum = (xnum + p[i]) * xsq; xden = (xden + q[i]) * xsq; }
        temp = xsq * (xnum + p[4]) / (xden + q[4]);
        temp = (0.398942280401432677939946059934 /* 1/sqrt(2 pi) */ - temp) / y;
        xsq = std::trunc(x * 16) / 16; del = (x - xsq) * (x + xsq);
        cum = std::exp(-xsq * xsq * 0.5) * std::exp(-del * 0.5) * temp; ccum = 1.0 - cum;
        return x > 0. ? ccum : cum;
    }
    return x > 0 ? 1.0 : 0.0;
}

// regularised lower incomplete gamma P(a, x), a > 0, x >= 0
static long double gamma_p(long double a, long double x) {
    if (x <= 0) return 0;
    const long double lg = lgammal(a);
    if (x < a + 1) {                                   // series
        long double ap = a, del = 1 / a, sum = del;
        for (int n = 0; n < 100000; n++) {
            ap += 1; del *= x / ap; sum += del;
            if (fabsl(del) < fabsl(sum) * 1e-20L) break;
        }
        return sum * expl(-x + a * logl(x) - lg);
    }
    const long double tiny = 1e-4000L;                 // Lentz continued fraction for Q(a, x)
    long double bb = x + 1 - a, cc = 1 / tiny, dd = 1 / bb, h = dd;
    for (int i = 1; i < 100000; i++) {
        const long double an = -i * (i - a);
        bb += 2;
        dd = an * dd + bb; if (fabsl(dd) < tiny) dd = tiny;
        cc = bb + an / cc; if (fabsl(cc) < tiny) cc = tiny;
        dd = 1 / dd;
        const long double dl = dd * cc;
        h *= dl;
        if (fabsl(dl - 1) < 1e-20L) break;
    }
    return 1 - expl(-x + a * logl(x) - lg) * h;
}

static double qchisq_upper_tail_point(double p, double df) {      // x with P[chi2_df <= x] = p
    const long double a = 0.5L * df;
    // Wilson-Hilferty start, then safeguarded Newton on P(a, x/2) - p
    long double lo = 0, hi = 1;
    while (gamma_p(a, hi / 2) < p) { lo = hi; hi *= 2; if (hi > 1e300L) return INFINITY; }
    long double x = 0.5L * (lo + hi);
    for (int it = 0; it < 200; it++) {
        const long double f = gamma_p(a, x / 2) - p;
        if (f > 0) hi = x; else lo = x;
        // density of chi2_df at x
        const long double dens = 0.5L * expl((a - 1) * logl(x / 2) - x / 2 - lgammal(a));
        long double xn = x - f / dens;
        if (!(xn > lo && xn < hi)) xn = 0.5L * (lo + hi);
        if (fabsl(xn - x) <= 1e-18L * fabsl(x)) { x = xn; break; }
        x = xn;
    }
    return static_cast<double>(x);
}


// -------------------------------------------------------------------------------------
// PacBio samplers (src/hts_pacbio.h:45-398, src/hts_pacbio.cpp:27-131)
// -------------------------------------------------------------------------------------
struct PacBioParams {
    double scale, sigma, loc, min_read_len;
    std::vector<double> read_probs; std::vector<u64> read_lens;
    u64 max_passes;
    std::vector<double> chi2_params_n, chi2_params_s, sqrt_params, norm_params;
    double prob_thresh, prob_ins, prob_del, prob_subst;
};

struct PacBioReadLenSampler {
    std::vector<u64> read_lens; Alias sampler; std::lognormal_distribution<double> distr;
    bool use_distr; double min_read_len, loc;
    PacBioReadLenSampler() : use_distr(true), min_read_len(1), loc(0) {}
    explicit PacBioReadLenSampler(const PacBioParams& p) {
        if (p.read_probs.empty()) {
            distr = std::lognormal_distribution<double>(std::log(p.scale), p.sigma);
            use_distr = true; min_read_len = std::ceil(p.min_read_len); loc = p.loc;
            if (min_read_len < 1) min_read_len = 1;
        } else {
            if (p.read_probs.size() != p.read_lens.size()) throw std::runtime_error("Probability and read lengths vector should be the same length.");
            read_lens = p.read_lens; sampler = Alias(p.read_probs); use_distr = false; min_read_len = 0; loc = 0;
        }
    }
    u64 sample(Pcg64& eng) {       // src/hts_pacbio.cpp:27-45
        u64 len_;
        if (use_distr) {
            double rnd = distr(eng) + loc;
            u64 iters = 0;
            while (rnd < min_read_len && iters < 10) { rnd = distr(eng) + loc; iters++; }
            if (rnd < min_read_len) rnd = min_read_len;
            len_ = static_cast<u64>(rnd);
        } else {
            u64 ind = sampler.sample(eng);
            len_ = read_lens[ind];
        }
        return len_;
    }
};

struct PacBioPassSampler {     // src/hts_pacbio.h:116-205
    u64 max_passes; std::vector<double> chi2_params_n, chi2_params_s;
    std::chi_squared_distribution<double> distr = std::chi_squared_distribution<double>(1);
    void sample(u64& split_pos, double& passes_left, double& passes_right, Pcg64& eng, const double& read_length) {
        double passes, prop_left;
        double n = chi2_params_n[0] * std::min(read_length, chi2_params_n[2]) + chi2_params_n[1];
        if (n < 0.001) n = 0.001;
        double s;
        if (read_length <= chi2_params_s[2]) {
            s = chi2_params_s[0] * read_length - chi2_params_s[1];
            if (s < 0.001) s = 0.001;
        } else s = chi2_params_s[3] / std::pow(read_length, chi2_params_s[4]);
        distr.param(std::chi_squared_distribution<double>::param_type(n));
        passes = distr(eng);
        double outlier_threshold = qchisq_upper_tail_point(0.9925, n);     // R::qchisq(0.9925, n, 1, 0)
        while (passes > outlier_threshold) passes = distr(eng);
        passes *= s;
        passes += 1;
        if (passes > max_passes) passes = max_passes;
        double fraction, wholes;
        fraction = std::modf(passes, &wholes);
        if ((static_cast<u64>(wholes) & 1ULL) == 0ULL) {
            prop_left = fraction;
            split_pos = std::round(static_cast<double>(read_length) * prop_left);
            passes_left = std::ceil(passes); passes_right = std::floor(passes);
        } else {
            prop_left = 1 - fraction;
            split_pos = std::round(static_cast<double>(read_length) * prop_left);
            passes_left = std::floor(passes); passes_right = std::ceil(passes);
        }
    }
};

static inline long double runif_ab(Pcg64& eng, const long double& a, const long double& b) {   // src/pcg.h:103-105
    return a + ((static_cast<long double>(eng()) + 1) / (MAX64 + 2)) * (b - a);
}

struct PacBioQualityError {    // src/hts_pacbio.h:213-398, src/hts_pacbio.cpp:50-131
    std::vector<double> sqrt_params, norm_params;
    double prob_thresh, prob_ins, prob_del, prob_subst, min_exp;
    std::vector<double> cum_probs_left = std::vector<double>(3), cum_probs_right = std::vector<double>(3);
    PacBioQualityError() {}
    explicit PacBioQualityError(const PacBioParams& p)
        : sqrt_params(p.sqrt_params), norm_params(p.norm_params), prob_thresh(p.prob_thresh), prob_ins(p.prob_ins),
          prob_del(p.prob_del), prob_subst(p.prob_subst), min_exp(calc_min_exp()) {}
    double calc_min_exp() {
        double min_exp_ = 1;
        double total = std::pow(prob_ins, min_exp_) + std::pow(prob_del, min_exp_) + std::pow(prob_subst, min_exp_);
        double left, right;
        if (total < prob_thresh) {
            while (total < prob_thresh) {
                min_exp_ /= 2;
                total = std::pow(prob_ins, min_exp_) + std::pow(prob_del, min_exp_) + std::pow(prob_subst, min_exp_);
            }
            left = min_exp_; right = min_exp_ * 2;
        } else {
            while (total > prob_thresh) {
                min_exp_ *= 2;
                total = std::pow(prob_ins, min_exp_) + std::pow(prob_del, min_exp_) + std::pow(prob_subst, min_exp_);
            }
            left = min_exp_ / 2; right = min_exp_;
        }
        for (u64 i = 0; i < 15; i++) {
            double m = (left + right) / 2;
            total = std::pow(prob_ins, m) + std::pow(prob_del, m) + std::pow(prob_subst, m);
            if (total == prob_thresh) { min_exp_ = m; break; }
            else if (total > prob_thresh) { left = m; min_exp_ = (m + right) / 2; }
            else { right = m; min_exp_ = (left + m) / 2; }
        }
        return min_exp_;
    }
    inline double sigmoid(const double& x) { return 1 / (1 + std::pow(2, (-2.5 / 3 * x + 6.5 / 3))); }
    double trunc_norm(const double& lower_thresh, Pcg64& eng) {
        double rnd;
        double a_bar = (lower_thresh - norm_params[0]) / norm_params[1];
        if (lower_thresh < (norm_params[0] + 5 * norm_params[1])) {
            double p = pnorm_std(a_bar);            // R::pnorm5(a_bar, 0, 1, 1, 0)
            double u = runif_ab(eng, p, 1);
            double x = qnorm(u);                    // R::qnorm5(u, 0, 1, 1, 0)
            rnd = x * norm_params[1] + norm_params[0];
        } else {
            double u, x_bar, v;
            u = runif_01(eng);
            x_bar = std::sqrt(a_bar * a_bar - 2 * std::log(1 - u));
            v = runif_01(eng);
            while (v > (x_bar / a_bar)) {
                u = runif_01(eng);
                x_bar = std::sqrt(a_bar * a_bar - 2 * std::log(1 - u));
                v = runif_01(eng);
            }
            rnd = norm_params[1] * x_bar + norm_params[0];
        }
        return rnd;
    }
    void update_probs(Pcg64& eng, const double& passes_left, const double& passes_right) {
        double left_thresh = (min_exp - (std::sqrt(passes_left + sqrt_params[0]) - sqrt_params[1])) / sigmoid(passes_left);
        double right_thresh = (min_exp - (std::sqrt(passes_right + sqrt_params[0]) - sqrt_params[1])) / sigmoid(passes_right);
        double incr_quals_l = trunc_norm(left_thresh, eng);
        double incr_quals_r = trunc_norm(right_thresh, eng);
        double exponent_left = incr_quals_l * sigmoid(passes_left) + std::sqrt(passes_left + sqrt_params[0]) - sqrt_params[1];
        double exponent_right = incr_quals_r * sigmoid(passes_right) + std::sqrt(passes_right + sqrt_params[0]) - sqrt_params[1];
        if (exponent_left < 0.6) exponent_left = 0.6;
        if (exponent_right < 0.6) exponent_right = 0.6;
        cum_probs_left[0] = std::pow(prob_ins, exponent_left);
        cum_probs_left[1] = std::pow(prob_del, exponent_left) + cum_probs_left[0];
        cum_probs_left[2] = std::pow(prob_subst, exponent_left) + cum_probs_left[1];
        cum_probs_right[0] = std::pow(prob_ins, exponent_right);
        cum_probs_right[1] = std::pow(prob_del, exponent_right) + cum_probs_right[0];
        cum_probs_right[2] = std::pow(prob_subst, exponent_right) + cum_probs_right[1];
    }
    void fill_quals(char& qual_left, char& qual_right) {
        const u64 max_qual = 93, qual_start = static_cast<u64>('!');
        u64 tmp_l = std::round(-10.0 * std::log10(cum_probs_left.back()));
        u64 tmp_r = std::round(-10.0 * std::log10(cum_probs_right.back()));
        if (tmp_l > max_qual) tmp_l = max_qual;
        if (tmp_r > max_qual) tmp_r = max_qual;
        qual_left = static_cast<char>(tmp_l + qual_start);
        qual_right = static_cast<char>(tmp_r + qual_start);
    }
    void sample(Pcg64& eng, char& qual_left, char& qual_right, std::deque<u64>& insertions, std::deque<u64>& deletions,
                std::deque<u64>& substitutions, const u64& chrom_len, const u64& read_length, const u64& split_pos,
                const double& passes_left, const double& passes_right) {
        insertions.clear(); deletions.clear(); substitutions.clear();
        update_probs(eng, passes_left, passes_right);
        fill_quals(qual_left, qual_right);
        u64 current_length = 0, chrom_pos = 0;
        u64 extra_space = chrom_len - read_length;
        double u;
        std::vector<double>* cum_probs = &cum_probs_left;
        while (current_length < read_length) {
            if (current_length == split_pos) cum_probs = &cum_probs_right;
            u = runif_01(eng);
            if (u > cum_probs->at(2)) {
                current_length++;
            } else if (u < cum_probs->at(0)) {
                if (current_length < (read_length - 1)) {
                    insertions.push_back(chrom_pos);
                    current_length++;
                    extra_space++;
                    if (current_length == split_pos) cum_probs = &cum_probs_right;
                }
                current_length++;
            } else if (u < cum_probs->at(1)) {
                if (extra_space > 0) { deletions.push_back(chrom_pos); extra_space--; }
            } else {
                substitutions.push_back(chrom_pos);
                current_length++;
            }
            chrom_pos++;
        }
    }
};

// PacBioOneGenome<T> (src/hts_pacbio.h:420-560, src/hts_pacbio.cpp:136-485).  The copy made per thread
// copies the samplers and quotas only; the read buffer and event deques start fresh (the reference's copy
// constructor does not copy them).
struct PacBioOneGenome {
    PacBioReadLenSampler len_sampler; PacBioPassSampler pass_sampler; PacBioQualityError qe_sampler;
    std::vector<u64> chrom_reads;
    const Genome* genome;
    u64 split_pos = 0; double passes_left = 0, passes_right = 0;
    char qual_left = '!', qual_right = '!';
    u64 read_chrom_space = 1;
    std::string read = std::string(1000, 'N');
    std::deque<u64> insertions, deletions, substitutions;
    u64 chrom_ind = 0, read_length = 0, read_start = 0;

    PacBioOneGenome(const Genome& g, const PacBioParams& p) : len_sampler(p), qe_sampler(p), genome(&g) {
        pass_sampler.max_passes = p.max_passes; pass_sampler.chi2_params_n = p.chi2_params_n; pass_sampler.chi2_params_s = p.chi2_params_s;
    }
    PacBioOneGenome(const PacBioOneGenome& o)
        : len_sampler(o.len_sampler), pass_sampler(o.pass_sampler), qe_sampler(o.qe_sampler), chrom_reads(o.chrom_reads), genome(o.genome) {}
    void reset_quota() { chrom_reads.clear(); }
    void add_n_reads(u64 n_reads, SeedSource& seeds) {
        std::vector<double> probs_(genome->chrom_sizes.begin(), genome->chrom_sizes.end());
        chrom_reads = reads_per_group(n_reads, probs_, seeds);
    }
    void sample_read_info(u64 chrom_len, Pcg64& eng) {     // shared middle of one_read (both overloads)
        read_length = len_sampler.sample(eng);
        if (read_length >= chrom_len) read_length = chrom_len;
        pass_sampler.sample(split_pos, passes_left, passes_right, eng, read_length);
        qe_sampler.sample(eng, qual_left, qual_right, insertions, deletions, substitutions, chrom_len, read_length,
                          split_pos, passes_left, passes_right);
        read_chrom_space = read_length + deletions.size() - insertions.size();
        if (read_chrom_space < chrom_len) {
            double u = runif_01(eng);
            read_start = static_cast<u64>(u * (chrom_len - read_chrom_space + 1));
        } else if (read_chrom_space == chrom_len) read_start = 0;
        else throw std::runtime_error("read_chrom_space should never exceed the chromosome length.");
    }
    bool resample_for_duplicate(u64 chrom_len, Pcg64& eng) {   // shared middle of re_read; false = give up
        pass_sampler.sample(split_pos, passes_left, passes_right, eng, read_length);
        qe_sampler.sample(eng, qual_left, qual_right, insertions, deletions, substitutions, chrom_len, read_length,
                          split_pos, passes_left, passes_right);
        read_chrom_space = read_length + deletions.size() - insertions.size();
        while ((read_chrom_space + read_start) > chrom_len) {
            if (deletions.empty()) break;
            deletions.pop_back();
            read_chrom_space--;
        }
        return !((read_chrom_space + read_start) > chrom_len);
    }
    // src/hts_pacbio.cpp:350-414 / :417-485
    void append_pool(const std::string& chrom, std::vector<char>& pool, Pcg64& eng) {
        bool reverse = runif_01(eng) < 0.5;
        pool.push_back('@');
        for (char c : genome->name) pool.push_back(c);
        pool.push_back('-');
        for (char c : genome->chrom_names[chrom_ind]) pool.push_back(c);
        pool.push_back('-');
        for (char c : std::to_string(read_start)) pool.push_back(c);
        pool.push_back('-');
        pool.push_back(reverse ? 'R' : 'F');
        pool.push_back('\n');
        fill_read(chrom, read, 0, read_start, read_chrom_space);
        if (reverse) rev_comp_n(read, read_chrom_space);
        u64 read_pos = 0, current_length = 0, rndi;
        while (current_length < read_length) {
            if (!insertions.empty() && read_pos == insertions.front()) {
                rndi = static_cast<u64>(runif_01(eng) * 4);
                pool.push_back(read[read_pos]);
                pool.push_back(BASES[rndi]);
                insertions.pop_front();
                current_length += 2;
            } else if (!deletions.empty() && read_pos == deletions.front()) {
                deletions.pop_front();
            } else if (!substitutions.empty() && read_pos == substitutions.front()) {
                rndi = static_cast<u64>(runif_01(eng) * 3);
                pool.push_back(MM_NUCLEOS[nt_index(read[read_pos])][rndi]);
                substitutions.pop_front();
                current_length++;
            } else {
                pool.push_back(read[read_pos]);
                current_length++;
            }
            read_pos++;
        }
        pool.push_back('\n'); pool.push_back('+'); pool.push_back('\n');
        for (u64 i = 0; i < split_pos; i++) pool.push_back(qual_left);
        for (u64 i = split_pos; i < read_length; i++) pool.push_back(qual_right);
        pool.push_back('\n');
    }
    void one_read(std::vector<std::vector<char>>& pools, bool& finished, Pcg64& eng) {
        chrom_ind = 0;
        while (chrom_ind < chrom_reads.size() && chrom_reads[chrom_ind] == 0) chrom_ind++;
        if (chrom_ind == chrom_reads.size()) { finished = true; return; }
        sample_read_info(genome->chrom_sizes[chrom_ind], eng);
        append_pool(*genome->chroms[chrom_ind], pools[0], eng);
        // NB: the reference never decrements chrom_reads on this path (src/hts_pacbio.cpp:136-188 only reads
        // it, :147), so every read of a thread comes from its first chromosome with a non-zero quota.
    }
    void re_read(std::vector<std::vector<char>>& pools, bool& finished, Pcg64& eng) {
        (void)finished;
        if (!resample_for_duplicate(genome->chrom_sizes[chrom_ind], eng)) return;
        append_pool(*genome->chroms[chrom_ind], pools[0], eng);
    }
    void one_read_str(const std::string& chrom, u64 chrom_i, std::vector<std::vector<char>>& pools, Pcg64& eng) {
        chrom_ind = chrom_i;
        sample_read_info(genome->chrom_sizes[chrom_ind], eng);
        append_pool(chrom, pools[0], eng);
    }
    void re_read_str(const std::string& chrom, u64 chrom_i, std::vector<std::vector<char>>& pools, Pcg64& eng) {
        chrom_ind = chrom_i;
        if (!resample_for_duplicate(genome->chrom_sizes[chrom_ind], eng)) return;
        append_pool(chrom, pools[0], eng);
    }
};

// PacBioHaplotypes (src/hts_pacbio.h:569-715, src/hts_pacbio.cpp:488-551)
struct PacBioHaplotypes {
    const std::vector<HapGenome>* haps;
    std::vector<Genome> genomes;
    std::vector<std::vector<u64>> n_reads_vc;
    std::vector<PacBioOneGenome> read_makers;
    std::vector<double> hap_probs;
    u64 hap, chr;
    std::string hap_chrom_seq;
    const std::string* cached_seq = nullptr;     // ChromCache entry standing in for hap_chrom_seq
    const std::string& cur_seq() const { return cached_seq ? *cached_seq : hap_chrom_seq; }
    PacBioHaplotypes(const std::vector<HapGenome>& hs, const std::vector<double>& probs, const PacBioParams& p)
        : haps(&hs), hap_probs(probs), hap(0), chr(0) {
        genomes.resize(hs.size());
        for (u64 i = 0; i < hs.size(); i++) {
            genomes[i].name = hs[i].name;
            for (const HapChrom& hc : hs[i].chroms) {
                genomes[i].chrom_names.push_back(hc.name);
                genomes[i].chroms.push_back(nullptr);
                genomes[i].chrom_sizes.push_back(hc.chrom_size);
            }
        }
        read_makers.reserve(hs.size());
        for (u64 i = 0; i < hs.size(); i++) read_makers.push_back(PacBioOneGenome(genomes[i], p));
    }
    PacBioHaplotypes(const PacBioHaplotypes& o)
        : haps(o.haps), genomes(o.genomes), n_reads_vc(o.n_reads_vc), read_makers(o.read_makers), hap_probs(o.hap_probs),
          hap(o.hap), chr(o.chr), hap_chrom_seq(o.hap_chrom_seq), cached_seq(o.cached_seq) {
        for (u64 i = 0; i < read_makers.size(); i++) read_makers[i].genome = &genomes[i];
    }
    void reset_quota() { n_reads_vc.clear(); for (auto& rm : read_makers) rm.reset_quota(); }
    void add_n_reads(u64 n_reads, SeedSource& seeds) {
        u64 n_haps = haps->size();
        std::vector<u64> hap_reads = reads_per_group(n_reads, hap_probs, seeds);
        for (u64 v = 0; v < n_haps; v++) {
            std::vector<double> chrom_probs;
            for (const HapChrom& vc : (*haps)[v].chroms) chrom_probs.push_back(vc.chrom_size);
            n_reads_vc.push_back(reads_per_group(hap_reads[v], chrom_probs, seeds));
        }
        for (u64 i = 0; i < n_haps; i++) read_makers[i].add_n_reads(hap_reads[i], seeds);
    }
    void one_read(std::vector<std::vector<char>>& pools, bool& finished, Pcg64& eng) {
        if (hap == haps->size()) { finished = true; return; }
        if (n_reads_vc[hap][chr] == 0 || cur_seq().empty()) {
            u64 new_hap = hap, new_chr = chr;
            for (; new_hap < n_reads_vc.size(); new_hap++) {
                while (n_reads_vc[new_hap][new_chr] == 0) {
                    new_chr++;
                    if (new_chr == n_reads_vc[new_hap].size()) break;
                }
                if (new_chr < n_reads_vc[new_hap].size()) break;
                else new_chr = 0;
            }
            hap = new_hap; chr = new_chr;
            if (hap == haps->size()) { finished = true; return; }
            if (g_chrom_cache) cached_seq = &g_chrom_cache->seq[hap * g_chrom_cache->n_chroms + chr];
            else hap_chrom_seq = (*haps)[hap].chroms[chr].get_chrom_full();
        }
        read_makers[hap].one_read_str(cur_seq(), chr, pools, eng);
        n_reads_vc[hap][chr]--;
    }
    void re_read(std::vector<std::vector<char>>& pools, bool& finished, Pcg64& eng) {
        if (hap == haps->size()) { finished = true; return; }
        read_makers[hap].re_read_str(cur_seq(), chr, pools, eng);
        if (n_reads_vc[hap][chr] > 0) n_reads_vc[hap][chr]--;
    }
};

static thread_local std::string g_err;

}  // namespace orc

using namespace orc;

extern "C" {

const char* orc_last_error(void) { return g_err.c_str(); }
void orc_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
// Options for the following orc_illumina_* / orc_pacbio_* calls (genome-scale tests):
// windows = n pairs {begin, end} of thread indices, increasing and disjoint: only those threads generate (n = 0: back
// to the args' thread_begin/thread_end); chrom cache: see ChromCache.
static std::vector<std::pair<u64, u64>> g_windows;
void orc_set_windows(const uint64_t* bounds, uint64_t n) {
    g_windows.clear();
    for (uint64_t i = 0; i < n; i++) g_windows.push_back({bounds[2 * i], bounds[2 * i + 1]});
}
void orc_set_chrom_cache(int on) { g_use_chrom_cache = on != 0; }
int orc_max_threads(void) { return omp_get_max_threads(); }
void orc_free(void* p) { std::free(p); }

void orc_pcg64_outputs(const uint32_t* sub_seeds, uint64_t n, uint64_t* out) {
    Pcg64 e = seeded_pcg(sub_seeds);
    for (uint64_t i = 0; i < n; i++) out[i] = e();
}

// Elementary conversions, evaluated with the reference's x87 expressions.
uint64_t orc_index(uint64_t x, uint64_t n) { return static_cast<uint64_t>(runif_01_from(x) * n); }         // alias_sampler.h:55
uint64_t orc_index_d(uint64_t x, double n) { return static_cast<uint64_t>(runif_01_from(x) * n); }          // hts_illumina.h:216,254
double orc_u01_double(uint64_t x) { double u = runif_01_from(x); return u; }                                   // alias_sampler.h:57
uint8_t orc_nqual(uint64_t x) { u8 q = runif_01_from(x) * 10 + static_cast<u8>('!'); return q; }              // hts_illumina.h:238
int orc_lt_half(uint64_t x) { return runif_01_from(x) < 0.5; }                                                 // hts_illumina.cpp:352
double orc_canonical(uint64_t x) {   // libstdc++ generate_canonical<double,53> over a 64-bit URBG (random.tcc:3348-3380)
    struct One { typedef uint64_t result_type; uint64_t v; static constexpr uint64_t min() { return 0; }
                 static constexpr uint64_t max() { return ~uint64_t(0); } uint64_t operator()() { return v; } } g{x};
    return std::generate_canonical<double, 53>(g);
}
uint64_t orc_frag_start(uint64_t x, uint64_t span) { double u = runif_01_from(x); return static_cast<uint64_t>(u * span); }  // hts_illumina.cpp:215-216
double orc_log(double x) { return std::log(x); }
double orc_qual_prob(uint32_t q) { return q == 0 ? 1.0 : std::pow(10, static_cast<double>(q) / -10.0); }

// Vectorised form of the primitives above with the operation codes of include/jackalope_hip.h
// (JK_OP_*), so tests can compare millions of inputs without per-call ctypes overhead.
void orc_eval_many(int what, const uint64_t* in, uint64_t n, uint64_t aux, uint64_t* out) {
    auto bits = [](double d) { uint64_t u; std::memcpy(&u, &d, 8); return u; };
    auto dbl = [](uint64_t u) { double d; std::memcpy(&d, &u, 8); return d; };
    for (uint64_t i = 0; i < n; i++) {
        switch (what) {
            case 0: { uint32_t w[8]; for (int k = 0; k < 8; k++) w[k] = (uint32_t)in[i * 8 + k];
                      Pcg64 e = seeded_pcg(w); for (uint64_t k = 0; k < aux; k++) out[i * aux + k] = e(); break; }
            case 1: out[i] = orc_index(in[i], aux); break;
            case 2: out[i] = bits(orc_u01_double(in[i])); break;
            case 3: out[i] = bits(orc_canonical(in[i])); break;
            case 4: out[i] = orc_nqual(in[i]); break;
            case 5: out[i] = orc_lt_half(in[i]); break;
            case 6: out[i] = orc_frag_start(in[i], aux); break;
            case 7: out[i] = bits(std::log(dbl(in[i]))); break;
            case 8: out[i] = bits(std::sqrt(dbl(in[i]))); break;
            case 10: out[i] = bits(std::exp(dbl(in[i]))); break;
            case 11: out[i] = bits(std::pow(dbl(in[2 * i]), dbl(in[2 * i + 1]))); break;
            case 12: out[i] = bits(std::log10(dbl(in[i]))); break;
            case 13: out[i] = bits(orc::qnorm(dbl(in[i]))); break;
            case 14: {   // (double) runif_ab(eng, a, 1): a + runif_01 * (b - a) in long double (src/pcg.h:103-105)
                const long double a = dbl(in[4 * i + 1]), b = 1.0L;
                const double u = a + runif_01_from(in[4 * i]) * (b - a);
                out[i] = bits(u); break; }
            default: break;
        }
    }
}
static double g_gamma_shape = 16.0, g_gamma_scale = 25.0;
void orc_set_gamma(double shape, double scale) { g_gamma_shape = shape; g_gamma_scale = scale; }
// JK_OP_GAMMA_STREAM: `aux` std::gamma_distribution draws per seeded engine (hts_illumina.cpp:206)
void orc_gamma_streams(const uint64_t* in, uint64_t n, uint64_t aux, uint64_t* out) {
    for (uint64_t i = 0; i < n; i++) {
        uint32_t w[8]; for (int k = 0; k < 8; k++) w[k] = (uint32_t)in[i * 8 + k];
        Pcg64 e = seeded_pcg(w);
        std::gamma_distribution<double> g(g_gamma_shape, g_gamma_scale);
        for (uint64_t k = 0; k < aux; k++) { double d = g(e); std::memcpy(&out[i * aux + k], &d, 8); }
    }
}

void orc_alias_build(const double* probs, uint64_t n, double* Prob, uint64_t* Ali) {
    Alias a(std::vector<double>(probs, probs + n));
    for (uint64_t i = 0; i < n; i++) { Prob[i] = a.Prob[i]; Ali[i] = a.Ali[i]; }
}

// n gamma fragment lengths + the engine outputs consumed, for one seeded engine (hts_illumina.cpp:206)
void orc_gamma_draws(const uint32_t* sub_seeds, double shape, double scale, uint64_t n, double* out) {
    Pcg64 e = seeded_pcg(sub_seeds);
    std::gamma_distribution<double> g(shape, scale);
    for (uint64_t i = 0; i < n; i++) out[i] = g(e);
}

void orc_split_int(uint64_t x, uint64_t n, uint64_t* out) {
    std::vector<u64> v = split_int(x, n);
    for (uint64_t i = 0; i < n; i++) out[i] = v[i];
}
int orc_reads_per_group(uint64_t n_reads, const double* probs, uint64_t n, const uint32_t* seed_words,
                        uint64_t n_seed_words, uint64_t* out, uint64_t* words_used) {
    try {
        SeedSource s{seed_words, n_seed_words, 0};
        std::vector<u64> v = reads_per_group(n_reads, std::vector<double>(probs, probs + n), s);
        for (uint64_t i = 0; i < n; i++) out[i] = v[i];
        *words_used = s.pos;
        return 0;
    } catch (std::exception& e) { g_err = e.what(); return 1; }
}

void orc_rev_comp(char* s, uint64_t n) { std::string t(s, n); rev_comp(t); std::memcpy(s, t.data(), n); }

struct orc_illumina_args {
    int32_t paired, matepair;
    uint64_t n_reads;
    double prob_dup;
    uint64_t n_threads, read_pool_size;
    double frag_len_shape, frag_len_scale;
    uint64_t frag_len_min, frag_len_max;
    uint32_t read_length;
    const uint32_t* n_quals1; const double* probs1; const uint8_t* quals1; double ins_prob1, del_prob1;
    const uint32_t* n_quals2; const double* probs2; const uint8_t* quals2; double ins_prob2, del_prob2;
    const uint32_t* seed_words; uint64_t n_seed_words;
    uint64_t thread_begin, thread_end;   // generate only these threads (0,0 = all)
    int32_t discard;                     // 1 = null sink (bytes are counted, not kept)
    uint64_t* thread_bytes1; uint64_t* thread_bytes2;   // optional out: [n_threads] bytes per thread and end
};

static void export_thread_bytes(const orc_illumina_args* a, const RunOpts& o) {
    if (a->thread_bytes1) for (u64 t = 0; t < a->n_threads; t++) a->thread_bytes1[t] = o.thread_bytes[0][t];
    if (a->thread_bytes2 && a->paired) for (u64 t = 0; t < a->n_threads; t++) a->thread_bytes2[t] = o.thread_bytes[1][t];
}

static IlluminaParams to_params(const orc_illumina_args* a) {
    IlluminaParams p;
    p.paired = a->paired; p.matepair = a->matepair;
    p.shape = a->frag_len_shape; p.scale = a->frag_len_scale;
    p.frag_len_min = a->frag_len_min; p.frag_len_max = a->frag_len_max;
    p.prof[0] = make_profile(a->read_length, a->n_quals1, a->probs1, a->quals1);
    p.ins_prob[0] = a->ins_prob1; p.del_prob[0] = a->del_prob1;
    if (a->paired) {
        p.prof[1] = make_profile(a->read_length, a->n_quals2, a->probs2, a->quals2);
        p.ins_prob[1] = a->ins_prob2; p.del_prob[1] = a->del_prob2;
    }
    return p;
}

// illumina_ref_cpp (src/hts_illumina.cpp:589-649) with uncompressed sinks returned in memory.
// Filler wrapper so run_threads' add_n_reads(n, seeds) signature is shared.
int orc_illumina_ref(uint64_t n_chroms, const char* const* chrom_names, const char* const* chrom_seqs,
                     const uint64_t* chrom_lens, const orc_illumina_args* a, const char* barcode,
                     char** out1, uint64_t* len1, char** out2, uint64_t* len2, uint64_t* seed_words_used) {
    try {
        std::vector<std::string> seqs(n_chroms);
        Genome g; g.name = "REF";
        for (uint64_t i = 0; i < n_chroms; i++) {
            seqs[i].assign(chrom_seqs[i], chrom_lens[i]);
            g.chrom_names.push_back(chrom_names[i]);
            g.chrom_sizes.push_back(chrom_lens[i]);
        }
        for (uint64_t i = 0; i < n_chroms; i++) g.chroms.push_back(&seqs[i]);
        IlluminaParams p = to_params(a);
        IlluminaOneGenome base(g, p, barcode ? barcode : "");
        SeedSource seeds{a->seed_words, a->n_seed_words, 0};
        std::vector<std::vector<char>> files;
        RunOpts opts; opts.thread_begin = a->thread_begin; opts.thread_end = a->thread_end; opts.discard = a->discard != 0; opts.windows = g_windows;
        run_threads(base, a->n_reads, a->prob_dup, a->read_pool_size, a->paired ? 2 : 1, a->n_threads, seeds, files, opts);
        export_thread_bytes(a, opts);
        give(files[0], out1, len1);
        if (a->paired) give(files[1], out2, len2); else { *out2 = nullptr; *len2 = 0; }
        if (seed_words_used) *seed_words_used = seeds.pos;
        return 0;
    } catch (std::exception& e) { g_err = e.what(); return 1; }
}

// Haplotype set handed over as flat arrays: for haplotype h, chromosome c (index h*n_chroms+c):
// chrom_size, n_mut, then mutation arrays concatenated in (h,c) order; nucleos as one blob with
// per-mutation offsets (n_mut_total+1 entries; equal consecutive offsets = deletion).
struct orc_hap_set {
    uint64_t n_haps, n_chroms;
    const char* const* hap_names;
    const char* const* chrom_names; const char* const* ref_seqs; const uint64_t* ref_lens;
    const uint64_t* chrom_size; const uint64_t* n_mut;
    const uint64_t* old_pos; const uint64_t* new_pos; const uint64_t* nuc_off; const char* nuc_blob;
};

static void build_haps(const orc_hap_set* hs, std::vector<std::string>& refs, std::vector<HapGenome>& haps) {
    refs.resize(hs->n_chroms);
    for (uint64_t c = 0; c < hs->n_chroms; c++) refs[c].assign(hs->ref_seqs[c], hs->ref_lens[c]);
    haps.resize(hs->n_haps);
    uint64_t m = 0;
    for (uint64_t h = 0; h < hs->n_haps; h++) {
        haps[h].name = hs->hap_names[h];
        haps[h].chroms.resize(hs->n_chroms);
        for (uint64_t c = 0; c < hs->n_chroms; c++) {
            HapChrom& hc = haps[h].chroms[c];
            uint64_t idx = h * hs->n_chroms + c;
            hc.ref = &refs[c]; hc.name = hs->chrom_names[c]; hc.chrom_size = hs->chrom_size[idx];
            for (uint64_t j = 0; j < hs->n_mut[idx]; j++, m++) {
                hc.old_pos.push_back(hs->old_pos[m]); hc.new_pos.push_back(hs->new_pos[m]);
                hc.nucleos.push_back(std::string(hs->nuc_blob + hs->nuc_off[m], hs->nuc_off[m + 1] - hs->nuc_off[m]));
            }
        }
    }
}

// HapChrom::get_chrom_full for (hap, chrom) -- src/hap_classes.cpp:80-116
int orc_hap_chrom_full(const orc_hap_set* hs, uint64_t hap, uint64_t chrom, char** out, uint64_t* len) {
    try {
        std::vector<std::string> refs; std::vector<HapGenome> haps;
        build_haps(hs, refs, haps);
        std::string s = haps[hap].chroms[chrom].get_chrom_full();
        *len = s.size(); *out = static_cast<char*>(std::malloc(s.size() ? s.size() : 1));
        std::memcpy(*out, s.data(), s.size());
        return 0;
    } catch (std::exception& e) { g_err = e.what(); return 1; }
}

// illumina_hap_cpp (src/hts_illumina.cpp:662-739), sep_files = FALSE, uncompressed.
int orc_illumina_hap(const orc_hap_set* hs, const double* hap_probs, const orc_illumina_args* a,
                     const char* const* barcodes, uint64_t n_barcodes,
                     char** out1, uint64_t* len1, char** out2, uint64_t* len2, uint64_t* seed_words_used) {
    try {
        std::vector<std::string> refs; std::vector<HapGenome> haps;
        build_haps(hs, refs, haps);
        IlluminaParams p = to_params(a);
        std::vector<std::string> bcs;
        for (uint64_t i = 0; i < n_barcodes; i++) bcs.push_back(barcodes[i]);
        IlluminaHaplotypes base(haps, std::vector<double>(hap_probs, hap_probs + hs->n_haps), p, bcs);
        ChromCache cache;
        struct CacheScope { ~CacheScope() { g_chrom_cache = nullptr; } } scope;
        if (g_use_chrom_cache) { cache.build(haps); g_chrom_cache = &cache; }
        SeedSource seeds{a->seed_words, a->n_seed_words, 0};
        std::vector<std::vector<char>> files;
        RunOpts opts; opts.thread_begin = a->thread_begin; opts.thread_end = a->thread_end; opts.discard = a->discard != 0; opts.windows = g_windows;
        run_threads(base, a->n_reads, a->prob_dup, a->read_pool_size, a->paired ? 2 : 1, a->n_threads, seeds, files, opts);
        export_thread_bytes(a, opts);
        give(files[0], out1, len1);
        if (a->paired) give(files[1], out2, len2); else { *out2 = nullptr; *len2 = 0; }
        if (seed_words_used) *seed_words_used = seeds.pos;
        return 0;
    } catch (std::exception& e) { g_err = e.what(); return 1; }
}

}  // extern "C" (reopened below)

struct orc_pacbio_args {
    uint64_t n_reads, n_threads, read_pool_size;
    double prob_dup;
    double scale, sigma, loc, min_read_len;
    const double* read_probs; const uint64_t* read_lens; uint64_t n_read_lens;
    uint64_t max_passes;
    const double* chi2_params_n;   // [3]
    const double* chi2_params_s;   // [5]
    const double* sqrt_params;     // [2]
    const double* norm_params;     // [2]
    double prob_thresh, prob_ins, prob_del, prob_subst;
    const uint32_t* seed_words; uint64_t n_seed_words;
    uint64_t thread_begin, thread_end; int32_t discard;
    uint64_t* thread_bytes;
};

static PacBioParams to_pb_params(const orc_pacbio_args* a) {
    PacBioParams p;
    p.scale = a->scale; p.sigma = a->sigma; p.loc = a->loc; p.min_read_len = a->min_read_len;
    if (a->n_read_lens) { p.read_probs.assign(a->read_probs, a->read_probs + a->n_read_lens); p.read_lens.assign(a->read_lens, a->read_lens + a->n_read_lens); }
    p.max_passes = a->max_passes;
    p.chi2_params_n.assign(a->chi2_params_n, a->chi2_params_n + 3);
    p.chi2_params_s.assign(a->chi2_params_s, a->chi2_params_s + 5);
    p.sqrt_params.assign(a->sqrt_params, a->sqrt_params + 2);
    p.norm_params.assign(a->norm_params, a->norm_params + 2);
    p.prob_thresh = a->prob_thresh; p.prob_ins = a->prob_ins; p.prob_del = a->prob_del; p.prob_subst = a->prob_subst;
    return p;
}

template <typename Filler>
static void run_pacbio(const Filler& base, const orc_pacbio_args* a, char** out, uint64_t* len, uint64_t* used) {
    SeedSource seeds{a->seed_words, a->n_seed_words, 0};
    std::vector<std::vector<char>> files;
    RunOpts opts; opts.thread_begin = a->thread_begin; opts.thread_end = a->thread_end; opts.discard = a->discard != 0; opts.windows = g_windows;
    run_threads(base, a->n_reads, a->prob_dup, a->read_pool_size, 1, a->n_threads, seeds, files, opts);
    if (a->thread_bytes) for (u64 t = 0; t < a->n_threads; t++) a->thread_bytes[t] = opts.thread_bytes[0][t];
    give(files[0], out, len);
    if (used) *used = seeds.pos;
}

extern "C" {

// pacbio_ref_cpp (src/hts_pacbio.cpp:579-640), uncompressed sink in memory
int orc_pacbio_ref(uint64_t n_chroms, const char* const* chrom_names, const char* const* chrom_seqs,
                   const uint64_t* chrom_lens, const orc_pacbio_args* a, char** out, uint64_t* len, uint64_t* seed_words_used) {
    try {
        std::vector<std::string> seqs(n_chroms);
        Genome g; g.name = "REF";
        for (uint64_t i = 0; i < n_chroms; i++) {
            seqs[i].assign(chrom_seqs[i], chrom_lens[i]);
            g.chrom_names.push_back(chrom_names[i]);
            g.chrom_sizes.push_back(chrom_lens[i]);
        }
        for (uint64_t i = 0; i < n_chroms; i++) g.chroms.push_back(&seqs[i]);
        PacBioOneGenome base(g, to_pb_params(a));
        run_pacbio(base, a, out, len, seed_words_used);
        return 0;
    } catch (std::exception& e) { g_err = e.what(); return 1; }
}

// pacbio_hap_cpp (src/hts_pacbio.cpp:646-715), sep_files = FALSE
int orc_pacbio_hap(const orc_hap_set* hs, const double* hap_probs, const orc_pacbio_args* a,
                   char** out, uint64_t* len, uint64_t* seed_words_used) {
    try {
        std::vector<std::string> refs; std::vector<HapGenome> haps;
        build_haps(hs, refs, haps);
        PacBioHaplotypes base(haps, std::vector<double>(hap_probs, hap_probs + hs->n_haps), to_pb_params(a));
        ChromCache cache;
        struct CacheScope { ~CacheScope() { g_chrom_cache = nullptr; } } scope;
        if (g_use_chrom_cache) { cache.build(haps); g_chrom_cache = &cache; }
        run_pacbio(base, a, out, len, seed_words_used);
        return 0;
    } catch (std::exception& e) { g_err = e.what(); return 1; }
}

double orc_pnorm(double x) { return pnorm_std(x); }
double orc_qchisq(double p, double df) { return qchisq_upper_tail_point(p, df); }

// create_chromosomes_ / create_genome_cpp (src/create_sequences.cpp:59-169): mt_seeds for n_threads engines,
// chromosomes dealt to threads by `omp for schedule(static)` (GCC/LLVM: contiguous blocks, the first
// n_chroms % n_threads threads get one more), per chromosome one gamma length draw (when len_sd > 0) and
// one alias draw per base over pi_tcag.  Run here thread after thread; the result does not depend on
// how the threads interleave.  lens[n_chroms]; *blob = concatenated chromosomes (orc_free).
int orc_create_genome(uint64_t n_chroms, double len_mean, double len_sd, const double* pi_tcag, uint64_t n_threads,
                      const uint32_t* seed_words, uint64_t n_seed_words, uint64_t* lens, char** blob,
                      uint64_t* blob_len, uint64_t* seed_words_used) {
    try {
        SeedSource seeds{seed_words, n_seed_words, 0};
        std::vector<const uint32_t*> tseeds(n_threads);
        for (u64 t = 0; t < n_threads; t++) tseeds[t] = seeds.take8();
        const Alias sampler(std::vector<double>(pi_tcag, pi_tcag + 4));
        const double gamma_shape = (len_mean * len_mean) / (len_sd * len_sd);
        const double gamma_scale = (len_sd * len_sd) / len_mean;
        const std::string bases = "TCAG";
        std::vector<std::string> chroms(n_chroms);
        const u64 q = n_chroms / n_threads, r = n_chroms % n_threads;
        for (u64 t = 0; t < n_threads; t++) {
            const u64 i0 = t < r ? (q + 1) * t : q * t + r;
            const u64 i1 = i0 + (t < r ? q + 1 : q);
            Pcg64 engine = seeded_pcg(tseeds[t]);
            std::gamma_distribution<double> distr;
            if (len_sd > 0) distr = std::gamma_distribution<double>(gamma_shape, gamma_scale);
            for (u64 i = i0; i < i1; i++) {
                u64 len;
                if (len_sd > 0) {
                    len = static_cast<u64>(distr(engine));
                    if (len < 1) len = 1;
                } else len = static_cast<u64>(len_mean);
                std::string& chrom = chroms[i];
                chrom.reserve(len);
                for (u64 j = 0; j < len; j++) chrom.push_back(bases[sampler.sample(engine)]);
            }
        }
        u64 total = 0;
        for (u64 i = 0; i < n_chroms; i++) { lens[i] = chroms[i].size(); total += lens[i]; }
        char* out = static_cast<char*>(std::malloc(total ? total : 1));
        if (!out) throw std::runtime_error("oracle: out of memory");
        u64 at = 0;
        for (u64 i = 0; i < n_chroms; i++) { std::memcpy(out + at, chroms[i].data(), lens[i]); at += lens[i]; }
        *blob = out; *blob_len = total; *seed_words_used = seeds.pos;
        return 0;
    } catch (std::exception& e) { g_err = e.what(); return 1; }
}

// ---- FASTA reader (src/io_fasta.cpp:41-169 non-indexed, :183-408 indexed) restated with the reference's own
// buffering: 4095-byte gzread pieces turned into C strings (so a NUL byte in the file truncates the piece),
// lines split on '\n' with one trailing '\r' dropped, a line containing '>' anywhere starts a chromosome,
// every sequence byte mapped through the filter tables of src/str_manip.h:24-56 (anything but
// TCAGN/tcagn becomes a zero byte).
}  // extern "C" (helpers below are C++)

namespace {

struct FaChrom { std::string name, nucleos; };

char fa_filter(char c, bool upper) {
    switch (c) {
        case 'T': case 'C': case 'A': case 'G': case 'N': return c;
        case 't': case 'c': case 'a': case 'g': case 'n': return upper ? static_cast<char>(c - 32) : c;
        default: return 0;
    }
}

std::vector<std::string> split_newline(const std::string& in) {          // cpp_str_split_newline, src/str_manip.h:142-171
    std::vector<std::string> out(1, "");
    std::string::size_type i0 = 0, i = in.find('\n');
    while (i != std::string::npos) {
        out.back().append(in, i0, i - i0);
        if (!out.back().empty() && out.back().back() == '\r') out.back().pop_back();
        i0 = i + 1;
        i = in.find('\n', i0);
        out.push_back("");
    }
    out.back().append(in, i0, std::string::npos);
    return out;
}

void fa_parse_line(const std::string& line, bool cut_names, std::vector<FaChrom>& ref) {   // parse_fasta_line, :43-65
    if (line.find(">") != std::string::npos) {
        std::string name;
        if (cut_names) {
            std::string::size_type spc = line.find(' ', 2);
            if (spc == std::string::npos) spc = line.size();
            name = line.substr(1, spc);
            name.erase(std::remove_if(name.begin(), name.end(), ::isspace), name.end());
        } else name = line.substr(1, line.size());
        ref.push_back(FaChrom{name, ""});
    } else {
        if (ref.empty()) throw std::runtime_error("oracle: sequence line before the first '>' line");
        ref.back().nucleos += line;
    }
}

void fa_append_noind(std::vector<FaChrom>& ref, const std::string& fn, bool cut_names, bool upper) {   // append_ref_noind, :74-136
    gzFile file = gzopen(fn.c_str(), "rb");
    if (!file) throw std::runtime_error("gzopen of " + fn + " failed: " + strerror(errno) + ".\n");
    const size_t first_new = ref.size();
    std::string lastline;
    std::vector<char> buffer(0x1000);
    for (;;) {
        const int bytes_read = gzread(file, buffer.data(), 0x1000 - 1);
        if (bytes_read < 0) { gzclose(file); throw std::runtime_error("gzread failed"); }
        buffer[bytes_read] = '\0';
        std::string mystring = lastline + std::string(buffer.data());
        std::vector<std::string> svec = split_newline(mystring);
        for (size_t i = 0; i + 1 < svec.size(); i++) fa_parse_line(svec[i], cut_names, ref);
        lastline = svec.back();
        if (bytes_read < 0x1000 - 1) {
            if (gzeof(file)) { fa_parse_line(lastline, cut_names, ref); break; }
        }
    }
    gzclose(file);
    (void)first_new;
    for (FaChrom& c : ref) for (char& ch : c.nucleos) ch = fa_filter(ch, upper);     // all chromosomes, again (:130-133)
}

void fa_append_ind(std::vector<FaChrom>& ref, const std::string& fn, const std::string& fai, bool upper) {   // :270-370
    std::vector<u64> offsets, lengths, line_lens;
    std::vector<std::string> names;
    {
        gzFile f = gzopen(fai.c_str(), "rb");
        if (!f) throw std::runtime_error("gzopen of " + fai + " failed: " + strerror(errno) + ".\n");
        std::string all; char buf[4096]; int n;
        while ((n = gzread(f, buf, sizeof buf)) > 0) all.append(buf, n);
        gzclose(f);
        for (const std::string& line : split_newline(all)) {
            if (line.empty()) continue;
            std::vector<std::string> cols(1, "");
            for (char ch : line) { if (ch == '\t') cols.push_back(""); else cols.back() += ch; }
            if (cols.size() < 4) throw std::runtime_error("oracle: short fai line");
            names.push_back(cols[0]); lengths.push_back(std::stoull(cols[1]));
            offsets.push_back(std::stoull(cols[2])); line_lens.push_back(std::stoul(cols[3]));
        }
    }
    gzFile file = gzopen(fn.c_str(), "rb");
    if (!file) throw std::runtime_error("gzopen of " + fn + " failed: " + strerror(errno) + ".\n");
    const u64 LIMIT = 4194304;
    for (size_t i = 0; i < offsets.size(); i++) {
        FaChrom rs; rs.name = names[i];
        const u64 len = lengths[i] + lengths[i] / line_lens[i] + 1;
        for (u64 j = 0; j < len; j += (LIMIT - 1)) {
            gzseek(file, offsets[i] + j, SEEK_SET);
            u64 partial_len = LIMIT;
            if (len - j < LIMIT) partial_len = len - j;
            std::vector<char> buffer(partial_len);
            const long bytes_read = gzread(file, buffer.data(), partial_len - 1);
            buffer[bytes_read > 0 ? bytes_read : 0] = '\0';
            std::string chrom_str(buffer.data());
            chrom_str.erase(std::remove(chrom_str.begin(), chrom_str.end(), '\n'), chrom_str.end());
            for (char& ch : chrom_str) ch = fa_filter(ch, upper);
            rs.nucleos += chrom_str;
            if (bytes_read < static_cast<long>(partial_len) && gzeof(file)) break;      // "fai file lengths appear incorrect"
        }
        ref.push_back(rs);
    }
    gzclose(file);
}

}  // namespace

extern "C" {

// read_fasta_noind / read_fasta_ind.  fai_files may be NULL.  Returns names joined by '\n', lens[], and the
// concatenated sequences (all malloc'd; orc_free).
int orc_read_fasta(const char* const* fasta_files, const char* const* fai_files, uint64_t n_files, int cut_names,
                   int remove_soft_mask, uint64_t* n_chroms, char** names, uint64_t* names_len, uint64_t** lens,
                   char** seqs, uint64_t* seqs_len) {
    try {
        std::vector<FaChrom> ref;
        for (uint64_t i = 0; i < n_files; i++) {
            if (fai_files) fa_append_ind(ref, fasta_files[i], fai_files[i], remove_soft_mask != 0);
            else fa_append_noind(ref, fasta_files[i], cut_names != 0, remove_soft_mask != 0);
        }
        std::string nm, sq;
        uint64_t* L = static_cast<uint64_t*>(std::malloc(sizeof(uint64_t) * (ref.size() + 1)));
        for (size_t i = 0; i < ref.size(); i++) {
            if (i) nm += '\n';
            nm += ref[i].name; sq += ref[i].nucleos; L[i] = ref[i].nucleos.size();
        }
        *n_chroms = ref.size();
        *names = static_cast<char*>(std::malloc(nm.size() + 1)); std::memcpy(*names, nm.data(), nm.size()); *names_len = nm.size();
        *seqs = static_cast<char*>(std::malloc(sq.size() + 1)); std::memcpy(*seqs, sq.data(), sq.size()); *seqs_len = sq.size();
        *lens = L;
        return 0;
    } catch (std::exception& e) { g_err = e.what(); return 1; }
}

}  // extern "C"
