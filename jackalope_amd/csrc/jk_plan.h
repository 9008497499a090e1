// jk_plan.h -- per-lane read quotas of a run (host; no device code).
//
// What the reference does per OpenMP thread before its parallel region (src/hts.h:349-353): every thread's
// filler copy gets add_n_reads(), which splits the thread's reads over haplotypes and chromosomes with
// reads_per_group() (src/hts.h:58-103), each call seeding a fresh pcg64 from the next 8 words of R's RNG.
// With ~10^6 lanes that is ~10^8 binomial draws, and the position of a lane's words in the seed stream
// depends on how many words the lanes before it took (a group with no reads takes none).
//
// LanePlanner reproduces that word for word, in O(own lanes) expensive work per process:
//  * the renormalised group probabilities of reads_per_group do not depend on the draws, so they are
//    computed once per probability vector (GroupChain) instead of once per lane (an O(G^2) chain of divisions);
//  * small binomials (t*p < 8: libstdc++'s waiting-time branch) are restated inline -- the same expression
//    over the same libm log, without the parameter set-up; larger ones go through std::binomial_distribution
//    itself, whose normal-deviate state persists across the groups of a call as in the reference;
//  * when the seed words are given as an array, a lane's word offset is speculated ("every haplotype gets
//    reads"), lanes are planned in parallel on host threads, and the offsets are verified and corrected by a
//    fix-point pass (exact in all cases; the common case needs one pass);
//  * with a seed callback (R's RNG: sequential by nature) the words are pulled in the reference's order on the
//    calling thread, the haplotype-level split is done there (it decides how many words follow), and the
//    chromosome-level splits of this process's lanes run on host threads.
#pragma once
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "jk_host.h"

namespace jk {

// JK_TIMING=1: wall-clock of the planner's phases on stderr
struct PlanTimer {
    const char* what; std::chrono::steady_clock::time_point t0; bool on;
    explicit PlanTimer(const char* w) : what(w), t0(std::chrono::steady_clock::now()), on(std::getenv("JK_TIMING") != nullptr) {}
    void lap(const char* next) {
        const auto t1 = std::chrono::steady_clock::now();
        if (on) std::fprintf(stderr, "[jk timing]   plan: %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        what = next; t0 = t1;
    }
    ~PlanTimer() { lap(""); }
};

// Number of host threads for planning work.
inline unsigned plan_threads(size_t n_items, size_t min_per_thread) {
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    if (const char* e = std::getenv("JK_HOST_THREADS")) { const int v = std::atoi(e); if (v >= 1) hw = (unsigned)v; }
    const size_t by_items = n_items / (min_per_thread ? min_per_thread : 1) + 1;
    return (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)hw, (size_t)64, by_items}));
}

template <typename F>   // f(begin, end, thread_index)
inline void parallel_for(size_t n, size_t min_per_thread, F f) {
    const unsigned nt = plan_threads(n, min_per_thread);
    if (nt <= 1 || n == 0) { f((size_t)0, n, 0u); return; }
    std::vector<std::thread> pool;
    std::vector<std::string> errs(nt);
    std::vector<int> codes(nt, 0);
    for (unsigned k = 0; k < nt; k++)
        pool.emplace_back([&, k] {
            try { f(n * k / nt, n * (k + 1) / nt, k); }
            catch (const Error& e) { errs[k] = e.what(); codes[k] = e.code; }
            catch (const std::exception& e) { errs[k] = e.what(); codes[k] = JK_ERR_ARG; }
        });
    for (std::thread& th : pool) th.join();
    for (unsigned k = 0; k < nt; k++) if (codes[k]) throw Error(codes[k], errs[k]);
}

// Description of a run's quota structure.
struct QuotaModel {
    bool hap = false;                      // haplotype run: hap-level split, then one chromosome split per haplotype
    uint32_t n_ends = 1;                   // read ends per "read" of the API (the driver counts pairs as 2)
    bool maker_halves = false;             // Illumina haplotypes, paired: each read maker's own add_n_reads() sees hap_reads / 2
    GroupChain hap_chain;                  // over haplotype_probs
    std::vector<GroupChain> chrom_chain;   // [n_haps] (ref: one entry) over chromosome sizes
    uint64_t n_haps = 1, n_chroms = 0;
};

// Zero-initialised array whose pages are first touched by the threads that fill it (calloc: no serial memset of the
// ~1 GB quota table of a 2^21-lane haplotype run).
template <typename T>
struct ZeroArray {
    T* p = nullptr; size_t n = 0;
    ZeroArray() {}
    ZeroArray(const ZeroArray&) = delete;
    ZeroArray& operator=(const ZeroArray&) = delete;
    ZeroArray(ZeroArray&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    ZeroArray& operator=(ZeroArray&& o) noexcept { if (this != &o) { std::free(p); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
    ~ZeroArray() { std::free(p); }
    void assign_zero(size_t count) {
        std::free(p); p = nullptr; n = count;
        if (count) { p = static_cast<T*>(std::calloc(count, sizeof(T))); if (!p) throw std::bad_alloc(); }
    }
    T* data() { return p; }
    const T* data() const { return p; }
    size_t size() const { return n; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
};

struct LanePlan {
    std::vector<uint32_t> lane_seeds;      // [n_shard * 8]
    std::vector<uint64_t> lane_reads;      // [n_shard] reads (all ends)
    ZeroArray<uint32_t> quotas;            // [cell][lane of the shard]
    uint64_t words_used = 0;               // position in the seed stream after the last lane that was planned
    uint64_t shard_begin_word = 0, shard_end_word = 0;   // add_n_reads words of this shard's lanes: [begin, end) in the stream
    // Deferred chromosome-level splits (`defer` of plan_lane_quotas): instead of `quotas`, the list of reads_per_group
    // calls that make it -- per call its 8 seed words, its read count, its lane and haplotype -- in lane order.  The
    // sessions run them on the device (chrom_split_kernel: 10^8 waiting-time binomials are nothing there, and the
    // 0.8 GB quota table of a 2^21-lane run never crosses the host link).
    bool deferred = false;
    ZeroArray<uint32_t> task_words, task_n, task_lane, task_hap;       // (filled by host threads: first touch is theirs)
    uint64_t n_tasks() const { return task_n.size(); }
};

// Lanes [begin, end) of a plan made for lanes [0, T): what one device of a multi-device run takes.
inline LanePlan slice_plan(const LanePlan& full, uint64_t T, uint64_t begin, uint64_t end) {
    LanePlan P;
    const uint64_t n = end - begin, n_cells = (T && !full.deferred) ? full.quotas.size() / T : 0;
    P.lane_seeds.assign(full.lane_seeds.begin() + begin * 8, full.lane_seeds.begin() + end * 8);
    P.lane_reads.assign(full.lane_reads.begin() + begin, full.lane_reads.begin() + end);
    if (full.deferred) {
        P.deferred = true;
        const uint32_t* tl = full.task_lane.data();
        const size_t nt = full.task_lane.size();
        const size_t a = std::lower_bound(tl, tl + nt, (uint32_t)begin) - tl;
        const size_t b = end > 0xffffffffULL ? nt : (size_t)(std::lower_bound(tl, tl + nt, (uint32_t)end) - tl);
        P.task_words.assign_zero((b - a) * 8); P.task_n.assign_zero(b - a); P.task_hap.assign_zero(b - a); P.task_lane.assign_zero(b - a);
        if (b > a) {
            std::memcpy(P.task_words.data(), full.task_words.data() + a * 8, (b - a) * 32);
            std::memcpy(P.task_n.data(), full.task_n.data() + a, (b - a) * 4);
            std::memcpy(P.task_hap.data(), full.task_hap.data() + a, (b - a) * 4);
        }
        for (size_t k = a; k < b; k++) P.task_lane[k - a] = tl[k] - (uint32_t)begin;
    } else {
        P.quotas.assign_zero((size_t)n_cells * n);
        for (uint64_t c = 0; c < n_cells; c++) std::memcpy(P.quotas.data() + c * n, full.quotas.data() + c * T + begin, n * 4);
    }
    P.words_used = full.words_used; P.shard_begin_word = full.shard_begin_word; P.shard_end_word = full.shard_end_word;
    return P;
}

namespace plan_detail {

// words a lane takes after its hap-level split (`hr` = reads per haplotype): one chromosome split per haplotype
// with reads, then each read maker's own add_n_reads (src/hts_illumina.h:639-641, src/hts_pacbio.h:697-699)
inline uint32_t words_after_hap_split(const QuotaModel& M, const uint64_t* hr) {
    uint32_t n = 0;
    for (uint64_t h = 0; h < M.n_haps; h++) {
        if (hr[h] > 0) n += 8;
        if ((M.maker_halves ? hr[h] / 2 : hr[h]) > 0) n += 8;
    }
    return n;
}

}  // namespace plan_detail

// pairs[t] = the lane's count as add_n_reads sees it (reads / n_ends); per_lane[t] = reads of lane t.
inline LanePlan plan_lane_quotas(const QuotaModel& M, const std::vector<uint64_t>& per_lane, uint64_t lane_begin, uint64_t lane_end,
                                 SeedReader& seeds, bool offset_given, uint64_t offset_words, bool defer = false) {
    using namespace plan_detail;
    const uint64_t T = per_lane.size(), n_shard = lane_end - lane_begin;
    const uint64_t nh = M.hap ? M.n_haps : 1, nc = M.n_chroms, n_cells = nh * nc;
    LanePlan P;
    P.lane_seeds.assign(n_shard * 8, 0);
    P.lane_reads.assign(n_shard, 0);
    if (n_shard > 0xffffffffULL) defer = false;
    P.deferred = defer;
    if (!defer) P.quotas.assign_zero((size_t)n_cells * n_shard);
    for (uint64_t l = 0; l < n_shard; l++) P.lane_reads[l] = per_lane[lane_begin + l];
    auto pairs_of = [&](uint64_t t) { return per_lane[t] / M.n_ends; };
    auto store_cell = [&](uint64_t cell0, uint64_t l) {
        return [&P, cell0, l, n_shard, mult = M.n_ends](size_t c, uint64_t v) { P.quotas[(cell0 + c) * n_shard + l] = (uint32_t)(v * mult); };
    };

    if (!seeds.src.words) {
        // ---- callback source: strictly sequential word consumption on this thread
        if (offset_given) throw Error(JK_ERR_ARG, "a seed-word offset needs the seed words as an array");
        uint32_t w[8];
        for (uint64_t t = 0; t < T; t++) {           // mt_seeds (src/pcg.h:37-46)
            seeds.take8(w);
            if (t >= lane_begin && t < lane_end) std::memcpy(&P.lane_seeds[(t - lane_begin) * 8], w, sizeof(w));
        }
        struct Task { uint64_t n; uint32_t w[8]; uint32_t h; uint64_t l; };
        std::vector<Task> tasks;
        std::vector<uint32_t> cb_words, cb_n, cb_lane, cb_hap;      // deferred tasks, grown as the callback delivers
        auto run_tasks = [&]() {                   // the chromosome-level splits collected so far, on host threads
            if (defer) {
                for (const Task& k : tasks) {
                    cb_words.insert(cb_words.end(), k.w, k.w + 8);
                    cb_n.push_back((uint32_t)k.n); cb_lane.push_back((uint32_t)k.l); cb_hap.push_back(k.h);
                }
                tasks.clear();
                return;
            }
            parallel_for(tasks.size(), 2048, [&](size_t a, size_t b, unsigned) {
                BinomDraw bd2;
                for (size_t i = a; i < b; i++) {
                    const Task& k = tasks[i];
                    split_with_chain(k.n, M.chrom_chain[k.h], k.w, bd2, store_cell((uint64_t)k.h * nc, k.l));
                }
            });
            tasks.clear();
        };
        BinomDraw bd;
        std::vector<uint64_t> hr(nh);
        for (uint64_t t = 0; t < T; t++) {
            const uint64_t n = pairs_of(t);
            const bool mine = t >= lane_begin && t < lane_end;
            if (t == lane_begin) P.shard_begin_word = seeds.pos;
            if (t == lane_end) P.shard_end_word = seeds.pos;
            if (n == 0) continue;
            if (tasks.size() >= (1u << 20)) run_tasks();
            if (!M.hap) {
                if (nc == 0) continue;
                seeds.take8(w);
                if (mine) { Task k; k.n = n; std::memcpy(k.w, w, 32); k.h = 0; k.l = t - lane_begin; tasks.push_back(k); }
                continue;
            }
            seeds.take8(w);
            std::fill(hr.begin(), hr.end(), (uint64_t)0);
            split_with_chain(n, M.hap_chain, w, bd, [&](size_t h, uint64_t v) { hr[h] = v; });
            for (uint64_t h = 0; h < nh; h++) {
                if (hr[h] == 0 || nc == 0) continue;
                seeds.take8(w);
                if (mine) { Task k; k.n = hr[h]; std::memcpy(k.w, w, 32); k.h = (uint32_t)h; k.l = t - lane_begin; tasks.push_back(k); }
            }
            for (uint64_t h = 0; h < nh; h++)
                if ((M.maker_halves ? hr[h] / 2 : hr[h]) > 0 && nc > 0) seeds.take8(w);
        }
        run_tasks();
        if (defer) {
            auto take = [](ZeroArray<uint32_t>& dst, const std::vector<uint32_t>& src) { dst.assign_zero(src.size()); if (!src.empty()) std::memcpy(dst.data(), src.data(), src.size() * 4); };
            take(P.task_words, cb_words); take(P.task_n, cb_n); take(P.task_lane, cb_lane); take(P.task_hap, cb_hap);
        }
        P.words_used = seeds.pos;
        if (lane_begin >= T) P.shard_begin_word = seeds.pos;
        if (lane_end >= T) P.shard_end_word = seeds.pos;
        return P;
    }

    // ---- array source: random access, lanes planned in parallel
    const uint32_t* W = seeds.src.words;
    const uint64_t NW = seeds.src.n_words;
    const uint64_t base = seeds.pos;                  // (words the caller already took, e.g. the per-file split of sep_files)
    if (base + T * 8 > NW) throw Error(JK_ERR_SEEDS, "seed source exhausted: the path needs more 32-bit sub-seed words than were supplied");
    for (uint64_t l = 0; l < n_shard; l++) std::memcpy(&P.lane_seeds[l * 8], W + base + (lane_begin + l) * 8, 32);
    const uint64_t after_mt = base + T * 8;
    static const uint32_t zero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto words_at = [&](uint64_t off) -> const uint32_t* { return off + 8 <= NW ? W + off : zero8; };   // (out of range: caught after the offsets are final)

    // lanes with reads are a prefix [0, n_active) (split_int gives the first x % n lanes one more)
    uint64_t n_active = 0;
    while (n_active < T && pairs_of(n_active) > 0) n_active++;
    if (nc == 0) { P.words_used = P.shard_begin_word = P.shard_end_word = after_mt; seeds.pos = after_mt; return P; }

    if (!M.hap) {
        // reference genome: every lane with reads takes exactly 8 words
        auto off_of = [&](uint64_t t) { return after_mt + 8 * std::min(t, n_active); };
        if (offset_given && offset_words != off_of(lane_begin)) throw Error(JK_ERR_ARG, "seed-word offset does not match the lanes before this shard");
        const uint64_t end_off = off_of(T);
        if (end_off > NW) throw Error(JK_ERR_SEEDS, "seed source exhausted: the path needs more 32-bit sub-seed words than were supplied");
        if (defer) {
            uint64_t nt = 0;
            for (uint64_t l = 0; l < n_shard; l++) nt += pairs_of(lane_begin + l) > 0;
            P.task_words.assign_zero(nt * 8); P.task_n.assign_zero(nt); P.task_lane.assign_zero(nt); P.task_hap.assign_zero(nt);
            uint64_t k = 0;
            for (uint64_t l = 0; l < n_shard; l++) {
                const uint64_t t = lane_begin + l, n = pairs_of(t);
                if (n == 0) continue;
                std::memcpy(&P.task_words[k * 8], W + off_of(t), 32);
                P.task_n[k] = (uint32_t)n; P.task_lane[k] = (uint32_t)l; k++;
            }
        } else
        parallel_for(n_shard, 4096, [&](size_t a, size_t b, unsigned) {
            BinomDraw bd;
            for (size_t l = a; l < b; l++) {
                const uint64_t t = lane_begin + l, n = pairs_of(t);
                if (n == 0) continue;
                split_with_chain(n, M.chrom_chain[0], W + off_of(t), bd, store_cell(0, l));
            }
        });
        P.words_used = offset_given ? off_of(lane_end) : end_off;
        P.shard_begin_word = off_of(lane_begin); P.shard_end_word = off_of(lane_end);
        seeds.pos = P.words_used;
        return P;
    }

    PlanTimer tm("set-up");
    // haplotypes.  need[t] = words lane t takes / 8; off[t] = its first word.  Planned range: [first, last).
    const uint64_t first = offset_given ? lane_begin : 0;
    const uint64_t last = offset_given ? lane_end : T;
    const uint64_t R = last - first;
    if (1 + 2 * nh > 0xffff) throw Error(JK_ERR_UNSUPPORTED, "too many haplotypes");
    std::vector<uint16_t> need(R, 0);
    std::vector<uint64_t> off(R + 1, 0);
    ZeroArray<uint32_t> hr_own;                        // [own lane][hap]: the hap-level split at the lane's final offset
    hr_own.assign_zero((size_t)n_shard * nh);
    auto keep_hr = [&](uint64_t t, const uint64_t* hr) {
        if (t >= lane_begin && t < lane_end) for (uint64_t h = 0; h < nh; h++) hr_own[(t - lane_begin) * nh + h] = (uint32_t)hr[h];
    };
    // speculation: every haplotype that can get reads gets enough for both of its seed draws.  The expected number of
    // lanes for which that is wrong decides whether the speculative passes are tried at all.
    uint16_t spec = 1;
    double expect_bad = 0;
    {
        const GroupChain& H = M.hap_chain;
        const uint64_t n_min = n_active ? pairs_of(n_active - 1) : 0;
        double rest = 1.0;
        for (size_t g = 0; g < H.G; g++) {
            double pg = rest;
            if (g + 1 < H.G) {
                if (H.kind[g] == 0) { pg = rest * H.p[g]; rest *= 1.0 - H.p[g]; }
                else if (H.kind[g] == 2) { pg = rest; rest = 0.0; }
                else pg = 0.0;
            }
            if (pg <= 0.0) continue;
            spec += 2;
            double miss = std::pow(1.0 - pg, (double)n_min);
            if (M.maker_halves && n_min > 0 && pg < 1.0) miss += (double)n_min * pg * std::pow(1.0 - pg, (double)(n_min - 1));
            expect_bad += miss * (double)n_active;
        }
    }
    for (uint64_t i = 0; i < R; i++) need[i] = pairs_of(first + i) > 0 ? spec : 0;
    const uint64_t start = offset_given ? offset_words : after_mt;
    auto rebuild = [&](uint64_t from) {                // off[i] for i > from, from need[]
        if (from == 0) off[0] = start;
        for (uint64_t i = from; i < R; i++) off[i + 1] = off[i] + 8ull * need[i];
    };
    rebuild(0);
    tm.lap("hap-level splits");
    // fix-point: compute every lane's real need at its current offset; a lane whose need differs moves all later
    // offsets.  Lanes before the first difference are final, so each pass makes progress; the usual case is one pass.
    uint64_t stable = 0;                               // lanes [0, stable) are final
    bool go_sequential = expect_bad > 4.0;
    for (int pass = 0; stable < R; pass++) {
        if (pass >= 6 || go_sequential) {
            // few reads per lane and haplotype (a haplotype without reads is then common): every such lane moves all
            // later offsets and a parallel pass only gets as far as the first of them.  The chain off[t+1] = off[t] +
            // need(n_t, words at off[t]) is a walk over the 8-word slots of the seed array whose step depends on the
            // slot and on n_t only -- and n_t takes two values (split_int: q+1 on the first lanes, q on the rest).  So
            // the slot space is cut into ranges, each walked on its own thread with the n its lanes are expected to
            // have; the one sequential walk that follows then finds nearly every slot it lands on already evaluated
            // and evaluates the others itself (around the q+1 -> q change, past the estimated end).
            BinomDraw bd;
            std::vector<uint64_t> hr(nh);
            auto need_of = [&](BinomDraw& d, std::vector<uint64_t>& h, uint64_t n, uint64_t word_off) -> uint16_t {
                std::fill(h.begin(), h.end(), (uint64_t)0);
                split_with_chain(n, M.hap_chain, words_at(word_off), d, [&](size_t k, uint64_t v) { h[k] = v; });
                return (uint16_t)(1 + words_after_hap_split(M, h.data()) / 8);
            };
            const uint64_t act = n_active > first ? std::min(R, n_active - first) : 0;      // lanes [0, act) of the range have reads
            const uint64_t i0 = stable;
            uint64_t i = i0;
            const uint64_t n_sample = std::min<uint64_t>(act > i ? act - i : 0, 4096);    // mean step, from the first lanes
            for (const uint64_t e = i + n_sample; i < e; i++) {
                need[i] = need_of(bd, hr, pairs_of(first + i), off[i]);
                off[i + 1] = off[i] + 8ull * need[i];
            }
            if (i < act) {
                const double mean = std::max(1.0, double(off[i] - off[i0]) / 8.0 / double(n_sample));
                const uint64_t w0 = off[i];                                               // slot 0
                const uint64_t nA = pairs_of(first + i), nB = pairs_of(first + act - 1);
                uint64_t isw = i;                                                         // first lane with nB reads
                if (nA != nB) { uint64_t lo = i, hi = act - 1; while (lo < hi) { uint64_t m = (lo + hi) / 2; if (pairs_of(first + m) == nB) hi = m; else lo = m + 1; } isw = lo; }
                else isw = i;
                const uint64_t RANGE = 16384;
                uint64_t n_slots = (uint64_t)(double(act - i) * mean * 1.02) + 1024;
                if (w0 < NW) n_slots = std::min(n_slots, (NW - w0) / 8); else n_slots = 0;
                const uint64_t n_ranges = (n_slots + RANGE - 1) / RANGE;
                std::vector<uint16_t> need_at(n_slots, 0);                               // 0: not evaluated
                std::vector<uint8_t> range_is_B(n_ranges, 0);
                std::vector<uint64_t> exit_slot(n_ranges, 0);                            // where range r's walk enters range r + 1
                for (uint64_t r = 0; r < n_ranges; r++) range_is_B[r] = nA == nB || i + (uint64_t)(double(r * RANGE) / mean) >= isw;
                // walk 1: every range from its first slot.  Walk 2: every range again from the slot its predecessor's
                // walk left at, until it lands on a slot of walk 1 (with steps of mostly 1+2*n_haps slots two walks need
                // some tens of lanes to fall into step) -- after which the chain from slot 0 is evaluated throughout,
                // unless some range's two walks never met.
                for (int walk = 0; walk < 2; walk++)
                    parallel_for(n_ranges, 1, [&](size_t a, size_t b, unsigned) {
                        BinomDraw d;
                        std::vector<uint64_t> h(nh);
                        for (size_t r = a; r < b; r++) {
                            if (walk == 1 && r == 0) continue;
                            const uint64_t end = std::min(n_slots, (r + 1) * RANGE);
                            uint64_t sl = walk == 0 ? r * RANGE : exit_slot[r - 1];
                            while (sl < end && (walk == 0 || need_at[sl] == 0)) {
                                const uint16_t nd = need_of(d, h, range_is_B[r] ? nB : nA, w0 + 8 * sl);
                                need_at[sl] = nd;
                                sl += nd;
                            }
                            if (walk == 0) exit_slot[r] = sl;
                        }
                    });
                uint64_t sl = 0, n_own_eval = 0;
                const uint64_t i_walk = i;
                for (; i < act; i++) {
                    const uint64_t n = pairs_of(first + i);
                    uint16_t nd = 0;
                    if (sl < n_slots && need_at[sl] != 0 && (range_is_B[sl / RANGE] ? nB : nA) == n) nd = need_at[sl];
                    else { nd = need_of(bd, hr, n, off[i]); n_own_eval++; }
                    need[i] = nd;
                    off[i + 1] = off[i] + 8ull * nd;
                    sl += nd;
                }
                if (tm.on) std::fprintf(stderr, "[jk timing]   plan: slot walk over %llu lanes in %llu ranges, %llu evaluated by the walk itself\n",
                                        (unsigned long long)(act - i_walk), (unsigned long long)n_ranges, (unsigned long long)n_own_eval);
            }
            for (; i < R; i++) { need[i] = 0; off[i + 1] = off[i]; }
            // the hap-level split of this shard's lanes, at their final offsets
            const uint64_t own_lo = std::max(first + i0, lane_begin), own_hi = std::min(first + act, lane_end);
            if (own_hi > own_lo)
                parallel_for(own_hi - own_lo, 2048, [&](size_t a, size_t b, unsigned) {
                    BinomDraw d;
                    std::vector<uint64_t> h(nh);
                    for (size_t k = a; k < b; k++) {
                        const uint64_t t = own_lo + k;
                        need_of(d, h, pairs_of(t), off[t - first]);
                        keep_hr(t, h.data());
                    }
                });
            stable = R;
            break;
        }
        std::atomic<uint64_t> first_bad{R};
        std::atomic<uint64_t> n_bad{0};
        const uint64_t s0 = stable;
        parallel_for(R - s0, 2048, [&](size_t a, size_t b, unsigned) {
            BinomDraw bd;
            std::vector<uint64_t> hr(nh);
            for (size_t k = a; k < b; k++) {
                const uint64_t i = s0 + k, n = pairs_of(first + i);
                uint16_t nd = 0;
                if (n > 0) {
                    std::fill(hr.begin(), hr.end(), (uint64_t)0);
                    split_with_chain(n, M.hap_chain, words_at(off[i]), bd, [&](size_t h, uint64_t v) { hr[h] = v; });
                    nd = (uint16_t)(1 + words_after_hap_split(M, hr.data()) / 8);
                    keep_hr(first + i, hr.data());
                }
                if (nd != need[i]) {
                    need[i] = nd;
                    n_bad.fetch_add(1, std::memory_order_relaxed);
                    uint64_t cur = first_bad.load();
                    while (i < cur && !first_bad.compare_exchange_weak(cur, i)) {}
                }
            }
        });
        const uint64_t fb = first_bad.load();
        if (fb == R) { stable = R; break; }
        rebuild(fb);                                   // lane fb itself was computed at a final offset
        stable = fb + 1;
        if (n_bad.load() > 4) go_sequential = true;    // (each further pass would cost a sweep over all remaining lanes)
    }
    if (off[R] > NW) throw Error(JK_ERR_SEEDS, "seed source exhausted: the path needs more 32-bit sub-seed words than were supplied");
    tm.lap("chromosome-level tasks");
    // chromosome-level splits of this shard's lanes
    if (defer) {
        std::vector<uint64_t> first_task(n_shard + 1, 0);
        for (uint64_t l = 0; l < n_shard; l++) {
            uint64_t c = 0;
            for (uint64_t h = 0; h < nh; h++) c += hr_own[l * nh + h] > 0;
            first_task[l + 1] = first_task[l] + c;
        }
        const uint64_t nt = first_task[n_shard];
        P.task_words.assign_zero(nt * 8); P.task_n.assign_zero(nt); P.task_lane.assign_zero(nt); P.task_hap.assign_zero(nt);
        parallel_for(n_shard, 4096, [&](size_t a, size_t b, unsigned) {
            for (size_t l = a; l < b; l++) {
                uint64_t o = off[lane_begin + l - first] + 8, k = first_task[l];
                for (uint64_t h = 0; h < nh; h++) {
                    const uint32_t n = hr_own[l * nh + h];
                    if (n == 0) continue;
                    std::memcpy(&P.task_words[k * 8], W + o, 32);
                    P.task_n[k] = n; P.task_lane[k] = (uint32_t)l; P.task_hap[k] = (uint32_t)h;
                    o += 8; k++;
                }
            }
        });
    } else
    parallel_for(n_shard, 512, [&](size_t a, size_t b, unsigned) {
        BinomDraw bd;
        for (size_t l = a; l < b; l++) {
            const uint64_t t = lane_begin + l, n = pairs_of(t);
            if (n == 0) continue;
            uint64_t o = off[t - first] + 8;           // (the hap-level split was kept by the fix-point pass)
            for (uint64_t h = 0; h < nh; h++) {
                const uint64_t k = hr_own[l * nh + h];
                if (k == 0) continue;
                split_with_chain(k, M.chrom_chain[h], W + o, bd, store_cell(h * nc, l));
                o += 8;
            }
        }
    });
    P.words_used = off[R];
    P.shard_begin_word = off[lane_begin - first]; P.shard_end_word = off[lane_end - first];
    seeds.pos = P.words_used;
    return P;
}

// Run (a subset of) deferred tasks on the host: quotas[(hap * n_chroms + g) * stride + lane].  `only` = task indices
// (nullptr: all).  Used for the tasks the device kernel hands back (a binomial outside libstdc++'s waiting-time branch).
inline void run_tasks_on_host(const QuotaModel& M, const LanePlan& P, const uint64_t* only, uint64_t n_only, uint32_t* out /* [n][n_chroms] */) {
    const uint64_t nc = M.n_chroms;
    const uint64_t n = only ? n_only : P.n_tasks();
    parallel_for(n, 256, [&](size_t a, size_t b, unsigned) {
        BinomDraw bd;
        for (size_t i = a; i < b; i++) {
            const uint64_t k = only ? only[i] : i;
            uint32_t* dst = out + i * nc;
            for (uint64_t g = 0; g < nc; g++) dst[g] = 0;
            split_with_chain(P.task_n[k], M.chrom_chain[M.hap ? P.task_hap[k] : 0], &P.task_words[k * 8], bd,
                             [dst, mult = M.n_ends](size_t g, uint64_t v) { dst[g] = (uint32_t)(v * mult); });
        }
    });
}

}  // namespace jk
