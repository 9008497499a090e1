// Rcpp shim: keeps the reference's RcppExports entry points for Illumina sequencing and forwards
// them to libjackalope_hip.so.  Drop this file into jackalope's src/ IN PLACE OF the bodies of
// illumina_ref_cpp / illumina_hap_cpp (src/hts_illumina.cpp:589-739); RcppExports.cpp/.R and the R
// function illumina() stay untouched.  It cannot be compiled in this repo's image (no R / Rcpp);
// INTEGRATION.md describes the build.
//
// Everything R-specific stays on the calling thread, as in the reference: XPtr access, expand_path
// (src/io.h:37-46) and the R RNG (Rcpp::runif inside the wrapper's RNGScope, src/pcg.h:37-71).
#include <RcppArmadillo.h>
#include <string>
#include <vector>

#include "jackalope_types.h"   // uint64, uint8
#include "ref_classes.h"       // RefGenome
#include "hap_classes.h"       // HapSet
#include "io.h"                // expand_path
#include "jackalope_hip.h"     // C ABI

using namespace Rcpp;

namespace {

// 8 sub-seed words exactly as fill-ins for mt_seeds()/seeded_pcg(): runif(8, 0, 2^32) truncated.
int r_seed_words(void*, uint32_t* out8) {
    NumericVector v = Rcpp::runif(8, 0, 4294967296.0);
    for (int i = 0; i < 8; i++) out8[i] = static_cast<uint32_t>(static_cast<uint64>(v[i]));
    return 0;
}

struct FlatProfile {
    std::vector<uint32_t> n_quals; std::vector<double> probs; std::vector<uint8_t> quals;
    jk_illumina_profile view{};
    FlatProfile(const std::vector<std::vector<std::vector<double>>>& p,
                const std::vector<std::vector<std::vector<uint8>>>& q) {
        if (p.size() != 4 || q.size() != 4) stop("All probs and quals for IlluminaQualityError must be of length 4");
        const size_t L = p[0].size();
        for (size_t nt = 0; nt < 4; nt++) {
            if (p[nt].size() != L || q[nt].size() != L) stop("In IlluminaQualityError construct, all probs' lengths not equal");
            for (size_t pos = 0; pos < L; pos++) {
                n_quals.push_back(p[nt][pos].size());
                probs.insert(probs.end(), p[nt][pos].begin(), p[nt][pos].end());
                quals.insert(quals.end(), q[nt][pos].begin(), q[nt][pos].end());
            }
        }
        view.read_length = L; view.n_quals = n_quals.data(); view.probs = probs.data(); view.quals = quals.data();
    }
};

struct RefView {
    std::vector<const char*> names, seqs; std::vector<uint64_t> lens; jk_ref_genome view{};
    explicit RefView(const RefGenome& g) {
        for (uint64 i = 0; i < g.size(); i++) {
            names.push_back(g[i].name.c_str()); seqs.push_back(g[i].nucleos.data()); lens.push_back(g[i].size());
        }
        view.n_chroms = g.size(); view.chrom_names = names.data(); view.chrom_seqs = seqs.data();
        view.chrom_lens = lens.data(); view.name = g.name.c_str();
    }
};

void check(int rc) { if (rc != JK_OK) throw Rcpp::exception(jk_last_error(), false); }

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
    FlatProfile p1(qual_probs1, quals1);
    std::string prefix = out_prefix;
    expand_path(prefix);
    std::vector<const char*> bcs;
    for (const std::string& b : barcodes) bcs.push_back(b.c_str());

    jk_illumina_args a{};
    a.paired = paired; a.matepair = matepair; a.out_prefix = prefix.c_str();
    a.compress = compress; a.comp_method = comp_method.c_str();
    a.n_reads = n_reads; a.prob_dup = prob_dup; a.n_threads = n_threads; a.show_progress = show_progress;
    a.read_pool_size = read_pool_size;
    a.frag_len_shape = frag_len_shape; a.frag_len_scale = frag_len_scale;
    a.frag_len_min = frag_len_min; a.frag_len_max = frag_len_max;
    a.profile1 = p1.view; a.ins_prob1 = ins_prob1; a.del_prob1 = del_prob1;
    std::unique_ptr<FlatProfile> p2;
    if (paired) { p2.reset(new FlatProfile(qual_probs2, quals2)); a.profile2 = p2->view; }
    a.ins_prob2 = ins_prob2; a.del_prob2 = del_prob2;
    a.barcodes = bcs.data(); a.n_barcodes = bcs.size();
    a.seeds.fn = r_seed_words;            // pulls from R's RNG in the reference's order
    check(jk_illumina_ref(&g.view, &a));
}
// illumina_hap_cpp forwards the same way with jk_hap_set built from XPtr<HapSet>
// (per (haplotype, chromosome): chrom_size and the AllMutations deques old_pos/new_pos/nucleos,
// src/hap_classes.h:100-104) and `sep_files`, `haplotype_probs` copied into the args.
