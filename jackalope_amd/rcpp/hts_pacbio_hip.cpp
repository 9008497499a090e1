// Rcpp shim: the reference's two PacBio entry points with their bodies forwarding to libjackalope_hip.so.
// Replaces the bodies of pacbio_ref_cpp / pacbio_hap_cpp at the end of src/hts_pacbio.cpp (:579-715); signatures,
// RcppExports.cpp/.R and R/hts_pacbio.R are unchanged.  See hts_illumina_hip.cpp.
#include "jk_rcpp_shim.h"

using namespace Rcpp;
using namespace jk_shim;

namespace {

struct PacbioCall {
    std::string prefix; Placement place;
    volatile int32_t abort_flag;
    jk_pacbio_args a;
    PacbioCall(const std::string& out_prefix, const int& compress, const std::string& comp_method, const uint64& n_reads,
               const uint64& n_threads, const bool& show_progress, const uint64& read_pool_size, const double& prob_dup,
               const double& scale, const double& sigma, const double& loc, const double& min_read_len,
               const std::vector<double>& read_probs, const std::vector<uint64>& read_lens, const uint64& max_passes,
               const std::vector<double>& chi2_params_n, const std::vector<double>& chi2_params_s,
               const std::vector<double>& sqrt_params, const std::vector<double>& norm_params,
               const double& prob_thresh, const double& prob_ins, const double& prob_del, const double& prob_subst)
        : prefix(out_prefix), place(n_threads), abort_flag(0) {
        if (read_probs.size() != read_lens.size()) stop("Probability and read lengths vector should be the same length.");   // src/hts_pacbio.h:73-75
        // (R/hts_pacbio.R:8-122 has checked these lengths already; the C ABI reads 3, 5, 2 and 2 values)
        if (chi2_params_n.size() != 3 || chi2_params_s.size() != 5 || sqrt_params.size() != 2 || norm_params.size() != 2)
            stop("chi2_params_n, chi2_params_s, sqrt_params and norm_params must have 3, 5, 2 and 2 values");
        expand_path(prefix);
        std::memset(&a, 0, sizeof(a));
        a.out_prefix = prefix.c_str(); a.compress = compress; a.comp_method = comp_method.c_str();
        a.n_reads = n_reads; a.n_threads = place.lanes; a.show_progress = show_progress;
        a.read_pool_size = read_pool_size; a.prob_dup = prob_dup;
        a.scale = scale; a.sigma = sigma; a.loc = loc; a.min_read_len = min_read_len;
        a.read_probs = read_probs.data();
        static_assert(sizeof(uint64) == sizeof(uint64_t), "uint64 is uint_fast64_t (src/jackalope_types.h:27)");
        a.read_lens = reinterpret_cast<const uint64_t*>(read_lens.data());
        a.n_read_lens = read_probs.size();
        a.max_passes = max_passes;
        a.chi2_params_n = chi2_params_n.data(); a.chi2_params_s = chi2_params_s.data();
        a.sqrt_params = sqrt_params.data(); a.norm_params = norm_params.data();
        a.prob_thresh = prob_thresh; a.prob_ins = prob_ins; a.prob_del = prob_del; a.prob_subst = prob_subst;
        a.seeds.fn = r_seed_words;
        a.abort_flag = &abort_flag;
        a.devices = place.devices.data(); a.n_devices = static_cast<uint32_t>(place.devices.size());
    }
};

}  // namespace

//[[Rcpp::export]]
void pacbio_ref_cpp(SEXP ref_genome_ptr, const std::string& out_prefix, const int& compress, const std::string& comp_method,
                    const uint64& n_reads, const uint64& n_threads, const bool& show_progress, const uint64& read_pool_size,
                    const double& prob_dup, const double& scale, const double& sigma, const double& loc,
                    const double& min_read_len, const std::vector<double>& read_probs, const std::vector<uint64>& read_lens,
                    const uint64& max_passes, const std::vector<double>& chi2_params_n, const std::vector<double>& chi2_params_s,
                    const std::vector<double>& sqrt_params, const std::vector<double>& norm_params,
                    const double& prob_thresh, const double& prob_ins, const double& prob_del, const double& prob_subst) {
    XPtr<RefGenome> ref_genome(ref_genome_ptr);
    RefView g(*ref_genome);
    PacbioCall c(out_prefix, compress, comp_method, n_reads, n_threads, show_progress, read_pool_size, prob_dup, scale, sigma,
                 loc, min_read_len, read_probs, read_lens, max_passes, chi2_params_n, chi2_params_s, sqrt_params, norm_params,
                 prob_thresh, prob_ins, prob_del, prob_subst);
    jk_job* job = nullptr;
    check(jk_pacbio_ref_job(&g.view, &c.a, &job));
    run_job(job, n_reads, compress, n_threads, show_progress, &c.abort_flag);
}

//[[Rcpp::export]]
void pacbio_hap_cpp(SEXP hap_set_ptr, const std::string& out_prefix, const bool& sep_files, const int& compress,
                    const std::string& comp_method, const uint64& n_reads, const uint64& n_threads, const bool& show_progress,
                    const uint64& read_pool_size, const std::vector<double>& haplotype_probs, const double& prob_dup,
                    const double& scale, const double& sigma, const double& loc, const double& min_read_len,
                    const std::vector<double>& read_probs, const std::vector<uint64>& read_lens, const uint64& max_passes,
                    const std::vector<double>& chi2_params_n, const std::vector<double>& chi2_params_s,
                    const std::vector<double>& sqrt_params, const std::vector<double>& norm_params,
                    const double& prob_thresh, const double& prob_ins, const double& prob_del, const double& prob_subst) {
    XPtr<HapSet> hap_set(hap_set_ptr);
    HapView hv(*hap_set);
    if (haplotype_probs.size() != hap_set->size()) stop("haplotype_probs must have one entry per haplotype");
    PacbioCall c(out_prefix, compress, comp_method, n_reads, n_threads, show_progress, read_pool_size, prob_dup, scale, sigma,
                 loc, min_read_len, read_probs, read_lens, max_passes, chi2_params_n, chi2_params_s, sqrt_params, norm_params,
                 prob_thresh, prob_ins, prob_del, prob_subst);
    c.a.sep_files = sep_files;
    c.a.haplotype_probs = haplotype_probs.data();
    jk_job* job = nullptr;
    check(jk_pacbio_hap_job(&hv.view, &c.a, &job));
    run_job(job, n_reads, compress, n_threads, show_progress, &c.abort_flag);
}
