// Rcpp shim: the reference's two Illumina entry points with their bodies forwarding to libjackalope_hip.so.
// Replaces the bodies of illumina_ref_cpp / illumina_hap_cpp at the end of src/hts_illumina.cpp (:589-739); signatures,
// RcppExports.cpp/.R and R/hts_illumina.R are unchanged.  Needs R + Rcpp, so it is not built in this repo's image;
// tests/rcpp_stubs/ holds just enough declarations to compile and drive it in the tests.
#include "jk_rcpp_shim.h"

using namespace Rcpp;
using namespace jk_shim;

namespace {

struct IlluminaCall {
    FlatProfile p1; std::unique_ptr<FlatProfile> p2;
    std::string prefix; std::vector<const char*> bcs; Placement place;
    volatile int32_t abort_flag;
    jk_illumina_args a;
    IlluminaCall(const bool& paired, const bool& matepair, const std::string& out_prefix, const int& compress,
                 const std::string& comp_method, const uint64& n_reads, const double& prob_dup, const uint64& n_threads,
                 const bool& show_progress, const uint64& read_pool_size, const double& frag_len_shape,
                 const double& frag_len_scale, const uint64& frag_len_min, const uint64& frag_len_max,
                 const std::vector<std::vector<std::vector<double>>>& qual_probs1,
                 const std::vector<std::vector<std::vector<uint8>>>& quals1, const double& ins_prob1, const double& del_prob1,
                 const std::vector<std::vector<std::vector<double>>>& qual_probs2,
                 const std::vector<std::vector<std::vector<uint8>>>& quals2, const double& ins_prob2, const double& del_prob2,
                 const std::vector<std::string>& barcodes)
        : p1(qual_probs1, quals1), prefix(out_prefix), place(n_threads), abort_flag(0) {
        expand_path(prefix);                                            // write_reads_cpp_, src/hts.h:451
        for (const std::string& b : barcodes) bcs.push_back(b.c_str());
        std::memset(&a, 0, sizeof(a));
        a.paired = paired; a.matepair = matepair; a.out_prefix = prefix.c_str();
        a.compress = compress; a.comp_method = comp_method.c_str();
        a.n_reads = n_reads; a.prob_dup = prob_dup; a.n_threads = place.lanes; a.show_progress = show_progress;
        a.read_pool_size = read_pool_size;
        a.frag_len_shape = frag_len_shape; a.frag_len_scale = frag_len_scale;
        a.frag_len_min = frag_len_min; a.frag_len_max = frag_len_max;
        a.profile1 = p1.view; a.ins_prob1 = ins_prob1; a.del_prob1 = del_prob1;
        if (paired) { p2.reset(new FlatProfile(qual_probs2, quals2)); a.profile2 = p2->view; }
        a.ins_prob2 = ins_prob2; a.del_prob2 = del_prob2;
        a.barcodes = bcs.data(); a.n_barcodes = bcs.size();
        a.seeds.fn = r_seed_words;                                      // R's RNG, in the reference's order
        a.abort_flag = &abort_flag;
        a.devices = place.devices.data(); a.n_devices = static_cast<uint32_t>(place.devices.size());
    }
};

}  // namespace

//[[Rcpp::export]]
void illumina_ref_cpp(SEXP ref_genome_ptr, const bool& paired, const bool& matepair, const std::string& out_prefix,
                      const int& compress, const std::string& comp_method, const uint64& n_reads,
                      const double& prob_dup, const uint64& n_threads, const bool& show_progress,
                      const uint64& read_pool_size, const double& frag_len_shape, const double& frag_len_scale,
                      const uint64& frag_len_min, const uint64& frag_len_max,
                      const std::vector<std::vector<std::vector<double>>>& qual_probs1,
                      const std::vector<std::vector<std::vector<uint8>>>& quals1,
                      const double& ins_prob1, const double& del_prob1,
                      const std::vector<std::vector<std::vector<double>>>& qual_probs2,
                      const std::vector<std::vector<std::vector<uint8>>>& quals2,
                      const double& ins_prob2, const double& del_prob2,
                      const std::vector<std::string>& barcodes) {
    XPtr<RefGenome> ref_genome(ref_genome_ptr);
    RefView g(*ref_genome);
    IlluminaCall c(paired, matepair, out_prefix, compress, comp_method, n_reads, prob_dup, n_threads, show_progress,
                   read_pool_size, frag_len_shape, frag_len_scale, frag_len_min, frag_len_max, qual_probs1, quals1,
                   ins_prob1, del_prob1, qual_probs2, quals2, ins_prob2, del_prob2, barcodes);
    jk_job* job = nullptr;
    check(jk_illumina_ref_job(&g.view, &c.a, &job));
    run_job(job, n_reads, compress, n_threads, show_progress, &c.abort_flag);
}

//[[Rcpp::export]]
void illumina_hap_cpp(SEXP hap_set_ptr, const bool& paired, const bool& matepair, const std::string& out_prefix,
                      const bool& sep_files, const int& compress, const std::string& comp_method, const uint64& n_reads,
                      const double& prob_dup, const uint64& n_threads, const bool& show_progress,
                      const uint64& read_pool_size, const std::vector<double>& haplotype_probs,
                      const double& frag_len_shape, const double& frag_len_scale,
                      const uint64& frag_len_min, const uint64& frag_len_max,
                      const std::vector<std::vector<std::vector<double>>>& qual_probs1,
                      const std::vector<std::vector<std::vector<uint8>>>& quals1,
                      const double& ins_prob1, const double& del_prob1,
                      const std::vector<std::vector<std::vector<double>>>& qual_probs2,
                      const std::vector<std::vector<std::vector<uint8>>>& quals2,
                      const double& ins_prob2, const double& del_prob2,
                      const std::vector<std::string>& barcodes) {
    XPtr<HapSet> hap_set(hap_set_ptr);
    HapView hv(*hap_set);
    if (haplotype_probs.size() != hap_set->size()) stop("haplotype_probs must have one entry per haplotype");
    IlluminaCall c(paired, matepair, out_prefix, compress, comp_method, n_reads, prob_dup, n_threads, show_progress,
                   read_pool_size, frag_len_shape, frag_len_scale, frag_len_min, frag_len_max, qual_probs1, quals1,
                   ins_prob1, del_prob1, qual_probs2, quals2, ins_prob2, del_prob2, barcodes);
    c.a.sep_files = sep_files;
    c.a.haplotype_probs = haplotype_probs.data();
    jk_job* job = nullptr;
    check(jk_illumina_hap_job(&hv.view, &c.a, &job));
    run_job(job, n_reads, compress, n_threads, show_progress, &c.abort_flag);
}
