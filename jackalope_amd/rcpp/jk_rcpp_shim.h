// jk_rcpp_shim.h -- what the four Rcpp entry points of the sequencers have in common when their bodies forward to
// libjackalope_hip.so (include/jackalope_hip.h).  Lives in jackalope's src/ next to hts_illumina_hip.cpp and
// hts_pacbio_hip.cpp; RcppExports.cpp/.R, NAMESPACE and the R functions stay untouched.
//
// Everything R-specific stays on R's main thread, as in the reference: XPtr access, expand_path (src/io.h:37-46),
// the R RNG (Rcpp::runif inside the generated wrapper's RNGScope, src/pcg.h:37-71), the progress bar and the
// interrupt check (src/hts.h:396-399,414).  Only jk_job_run -- which makes no R call -- runs on a worker thread.
#ifndef JK_RCPP_SHIM_H
#define JK_RCPP_SHIM_H

#include <RcppArmadillo.h>
#include <progress.hpp>        // RcppProgress
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "jackalope_types.h"   // uint64, uint8
#include "ref_classes.h"       // RefGenome, RefChrom
#include "hap_classes.h"       // HapSet, HapGenome, HapChrom, AllMutations
#include "io.h"                // expand_path
#include "jackalope_hip.h"     // the C ABI

namespace jk_shim {

// 8 sub-seed words as mt_seeds()/seeded_pcg() draw them: runif(8, 0, 2^32) truncated (src/pcg.h:37-46,63-71).
// Called by jk_job_plan_next on the thread that called it -- R's main thread.
inline int r_seed_words(void*, uint32_t* out8) {
    Rcpp::NumericVector v = Rcpp::runif(8, 0, 4294967296.0);
    for (int i = 0; i < 8; i++) out8[i] = static_cast<uint32_t>(static_cast<uint64>(v[i]));
    return 0;
}

inline void check(int rc) { if (rc != JK_OK) throw Rcpp::exception(jk_last_error(), false); }

// qual_probs / quals ([nt][pos][k]) -> jk_illumina_profile, with the checks of the IlluminaQualityError constructor
// (src/hts_illumina.h:154-181)
struct FlatProfile {
    std::vector<uint32_t> n_quals; std::vector<double> probs; std::vector<uint8_t> quals;
    jk_illumina_profile view;
    FlatProfile(const std::vector<std::vector<std::vector<double>>>& p,
                const std::vector<std::vector<std::vector<uint8>>>& q) {
        std::memset(&view, 0, sizeof(view));
        if (p.size() != 4 || q.size() != 4) Rcpp::stop("All probs and quals for IlluminaQualityError must be of length 4");
        const size_t L = p[0].size();
        for (size_t nt = 0; nt < 4; nt++) {
            if (p[nt].size() != L || q[nt].size() != L) Rcpp::stop("In IlluminaQualityError construct, all probs' lengths not equal");
            for (size_t pos = 0; pos < L; pos++) {
                if (p[nt][pos].size() != q[nt][pos].size()) Rcpp::stop("Probability and quality vector should be the same length.");
                n_quals.push_back(static_cast<uint32_t>(p[nt][pos].size()));
                probs.insert(probs.end(), p[nt][pos].begin(), p[nt][pos].end());
                quals.insert(quals.end(), q[nt][pos].begin(), q[nt][pos].end());
            }
        }
        view.read_length = static_cast<uint32_t>(L);
        view.n_quals = n_quals.data(); view.probs = probs.data(); view.quals = quals.data();
    }
};

// RefGenome (src/ref_classes.h:127-180) -> jk_ref_genome; borrows the chromosome strings
struct RefView {
    std::vector<const char*> names, seqs; std::vector<uint64_t> lens; jk_ref_genome view;
    explicit RefView(const RefGenome& g) {
        std::memset(&view, 0, sizeof(view));
        for (uint64 i = 0; i < g.size(); i++) {
            names.push_back(g[i].name.c_str()); seqs.push_back(g[i].nucleos.data()); lens.push_back(g[i].size());
        }
        view.n_chroms = g.size(); view.chrom_names = names.data(); view.chrom_seqs = seqs.data();
        view.chrom_lens = lens.data(); view.name = g.name.c_str();
    }
};

// HapSet (src/hap_classes.h:500-611) -> jk_hap_set: per (haplotype, chromosome) cell chrom_size and the three
// AllMutations deques (src/hap_classes.h:100-104; nucleos[m] == nullptr is a deletion), concatenated in cell order
struct HapView {
    RefView ref;
    std::vector<const char*> hap_names;
    std::vector<uint64_t> chrom_size, n_mut, old_pos, new_pos, nuc_off;
    std::string blob;
    jk_hap_set view;
    explicit HapView(const HapSet& hs) : ref(*hs.reference) {
        std::memset(&view, 0, sizeof(view));
        nuc_off.push_back(0);
        for (uint64 h = 0; h < hs.size(); h++) {
            hap_names.push_back(hs[h].name.c_str());
            for (uint64 c = 0; c < hs[h].size(); c++) {
                const HapChrom& hc = hs[h][c];
                const AllMutations& mu = hc.mutations;
                chrom_size.push_back(hc.chrom_size);
                n_mut.push_back(mu.size());
                for (uint64 m = 0; m < mu.size(); m++) {
                    old_pos.push_back(mu.old_pos[m]); new_pos.push_back(mu.new_pos[m]);
                    if (mu.nucleos[m] != nullptr) blob.append(mu.nucleos[m]);
                    nuc_off.push_back(blob.size());
                }
            }
        }
        view.n_haps = hs.size(); view.n_chroms = hs.reference->size();
        view.hap_names = hap_names.data(); view.ref = ref.view;
        view.chrom_size = chrom_size.data(); view.n_mut = n_mut.data();
        view.old_pos = old_pos.data(); view.new_pos = new_pos.data(); view.nuc_off = nuc_off.data();
        view.nuc_blob = blob.data();
    }
};

// Which GPUs a call uses, and with how many generator lanes.  By default every visible device and the lanes R asked
// for (n_threads: the files then equal the reference's for that thread count).  Environment overrides:
//   JACKALOPE_HIP_DEVICES="0,2,3"   the devices
//   JACKALOPE_HIP_LANES=1048576     lanes instead of n_threads (GPU throughput needs 2^18..2^20 lanes per device;
//                                   the reads then are those of a reference run with that many threads)
struct Placement {
    std::vector<int32_t> devices; uint64_t lanes;
    explicit Placement(uint64_t n_threads) : lanes(n_threads) {
        if (const char* e = std::getenv("JACKALOPE_HIP_DEVICES")) {
            for (const char* p = e; *p;) { char* end; long v = std::strtol(p, &end, 10); if (end == p) break; devices.push_back(static_cast<int32_t>(v)); p = (*end == ',') ? end + 1 : end; }
        }
        if (devices.empty()) { const int n = jk_device_count(); for (int i = 0; i < n; i++) devices.push_back(i); }
        if (devices.empty()) Rcpp::stop("no MI355X device is visible to this R session");
        if (const char* e = std::getenv("JACKALOPE_HIP_LANES")) { const long long v = std::atoll(e); if (v >= 1) lanes = static_cast<uint64_t>(v); }
        else if (lanes < 4096) {
            // The trade-off is the user's: n_threads lanes reproduce the reference's files for that thread count, but an
            // Illumina lane is ONE GPU thread -- a handful of them is slower than the CPU path (PacBio lanes are worked on by
            // a whole wave each and do not mind).  Say so once per call instead of running slowly in silence.
            Rcpp::warning("jackalope (HIP): n_threads = %d gives %d generator lane(s) -- the output equals the reference's for that "
                          "thread count, but the GPU needs 2^16..2^20 lanes for Illumina reads. Set the environment variable "
                          "JACKALOPE_HIP_LANES (e.g. 262144): the reads are then those of a reference run with that many threads.",
                          static_cast<int>(lanes), static_cast<int>(lanes));
        }
    }
};

// The reference's tail of each entry point (progress bar sized n_reads [+ n_reads / 2 for the compression pass],
// write_reads_cpp_ / write_reads_cpp_sep_files_): per output file set, plan on this thread, run on a worker, and
// keep the bar and the interrupt check going here.
inline void run_job(jk_job* job, uint64 n_reads, int compress, uint64 n_threads, bool show_progress, volatile int32_t* abort_flag) {
    struct Free { jk_job* j; ~Free() { jk_job_free(j); } } guard{job};
    uint64 prog_n = n_reads;
    if (compress > 0 && n_threads > 1) prog_n += (n_reads / 2);      // (kept: the bar of the reference has this size)
    Progress prog_bar(prog_n, show_progress);
    uint64_t shown = 0;
    const uint32_t n_files = jk_job_n_files(job);
    for (uint32_t f = 0; f < n_files; f++) {
        if (Progress::check_abort()) break;                            // src/hts.h:536
        check(jk_job_plan_next(job));                                  // draws from R's RNG, here
        std::atomic<int> rc(-1);
        std::string err;
        std::thread worker([&] { int r = jk_job_run(job); if (r != JK_OK) err = jk_last_error(); rc.store(r); });
        while (rc.load() < 0) {
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
            if (Progress::check_abort()) *abort_flag = 1;              // src/hts.h:396-399
            uint64_t done = 0, total = 0;
            if (jk_job_progress(job, &done, &total) == JK_OK && done > shown) { prog_bar.increment(done - shown); shown = done; }
        }
        worker.join();
        if (rc.load() == JK_ERR_ABORTED) break;                        // the reference leaves its loops the same way
        if (rc.load() != JK_OK) throw Rcpp::exception(err.c_str(), false);
    }
    if (shown < prog_n && !Progress::check_abort()) prog_bar.increment(prog_n - shown);
}

}  // namespace jk_shim
#endif
