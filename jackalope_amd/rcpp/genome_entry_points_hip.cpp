// Rcpp shim for the entry points on either side of the sequencers: the bodies of
//   make_hap_set, add_substitution, add_insertion, add_deletion   (src/ref_hap_access.cpp:127-132, :816-865)
//   create_genome_cpp                                             (src/create_sequences.cpp:151-169)
//   read_fasta_noind, read_fasta_ind                              (src/io_fasta.cpp:153-169, :389-408)
// forwarded to libjackalope_hip.so.  Signatures, RcppExports.cpp/.R and the R functions stay untouched.
// Like hts_illumina_hip.cpp this file cannot be compiled in this repo's image (no R / Rcpp); INTEGRATION.md
// describes the build.
//
// With these in place a genome made by create_genome()/read_fasta() is an XPtr<jk_genome> that lives in GPU
// memory; the sequencer shims obtain their jk_ref_genome from jk_genome_view (device pointers,
// seqs_on_device = 1) instead of walking a RefGenome, and a haplotype set is an XPtr<jk_hap_builder>
// whose tables jk_hap_builder_view flattens for jk_illumina_hap / jk_pacbio_hap.  R-level accessors that
// need bases on the host (ref$chrom(i), haps$chrom(h, i)) go through jk_genome_fetch / jk_hap_chrom_full.
#include <RcppArmadillo.h>
#include <string>
#include <vector>

#include "jackalope_types.h"   // uint64
#include "io.h"                // expand_path
#include "jackalope_hip.h"     // C ABI

using namespace Rcpp;

namespace {

int r_seed_words(void*, uint32_t* out8) {          // Rcpp::runif(8, 0, 2^32) truncated, src/pcg.h:37-46
    NumericVector v = Rcpp::runif(8, 0, 4294967296.0);
    for (int i = 0; i < 8; i++) out8[i] = static_cast<uint32_t>(static_cast<uint64>(v[i]));
    return 0;
}
void check(int rc) { if (rc != JK_OK) throw Rcpp::exception(jk_last_error(), false); }
void free_genome(jk_genome* g) { jk_genome_free(g); }
void free_builder(jk_hap_builder* b) { jk_hap_builder_free(b); }
typedef XPtr<jk_genome, PreserveStorage, free_genome> GenomePtr;
typedef XPtr<jk_hap_builder, PreserveStorage, free_builder> BuilderPtr;

}  // namespace

//[[Rcpp::export]]
SEXP create_genome_cpp(const uint64& n_chroms, const double& len_mean, const double& len_sd,
                       std::vector<double> pi_tcag, const uint64& n_threads) {
    jk_seed_source seeds{};
    seeds.fn = r_seed_words;                         // mt_seeds(n_threads): 8 words per thread, in order
    jk_genome* g = nullptr;
    check(jk_create_genome(n_chroms, len_mean, len_sd, pi_tcag.data(), n_threads, &seeds, 0, &g));
    return GenomePtr(g, true);
}

//[[Rcpp::export]]
SEXP read_fasta_noind(const std::vector<std::string>& fasta_files, const bool& cut_names, const bool& remove_soft_mask) {
    std::vector<std::string> paths(fasta_files);
    std::vector<const char*> p;
    for (std::string& f : paths) { expand_path(f); p.push_back(f.c_str()); }
    jk_genome* g = nullptr;
    check(jk_read_fasta(p.data(), nullptr, p.size(), cut_names, remove_soft_mask, 0, &g));
    return GenomePtr(g, true);
}

//[[Rcpp::export]]
SEXP read_fasta_ind(const std::vector<std::string>& fasta_files, const std::vector<std::string>& fai_files,
                    const bool& remove_soft_mask) {
    if (fasta_files.size() != fai_files.size())
        stop("\nThe vector of fasta index files must be the same length as the vector of fasta files.");
    std::vector<std::string> fa(fasta_files), fai(fai_files);
    std::vector<const char*> p, q;
    for (std::string& f : fa) { expand_path(f); p.push_back(f.c_str()); }
    for (std::string& f : fai) { expand_path(f); q.push_back(f.c_str()); }
    jk_genome* g = nullptr;
    check(jk_read_fasta(p.data(), q.data(), p.size(), /*cut_names=*/0, remove_soft_mask, 0, &g));
    return GenomePtr(g, true);
}

// The builder edits tables on the host, so it needs the reference bases there: a device-resident genome is
// fetched once (jk_genome_fetch) into vectors that live as long as the builder (kept in an R attribute).
//[[Rcpp::export]]
SEXP make_hap_set(SEXP ref_genome_ptr, const uint64& n_haps) {
    GenomePtr g(ref_genome_ptr);
    jk_ref_genome dev{};
    check(jk_genome_view(g.get(), &dev));
    std::vector<std::string>* bases = new std::vector<std::string>(dev.n_chroms);
    std::vector<const char*> seqs(dev.n_chroms);
    for (uint64 i = 0; i < dev.n_chroms; i++) {
        (*bases)[i].resize(dev.chrom_lens[i]);
        check(jk_genome_fetch(g.get(), i, &(*bases)[i][0], dev.chrom_lens[i]));
        seqs[i] = (*bases)[i].data();
    }
    jk_ref_genome host = dev;
    host.chrom_seqs = seqs.data();
    host.seqs_on_device = 0;
    jk_hap_builder* b = nullptr;
    check(jk_hap_builder_new(&host, n_haps, &b));
    BuilderPtr out(b, true);
    out.attr("host_bases") = XPtr<std::vector<std::string>>(bases, true);
    return out;
}

//[[Rcpp::export]]
void add_substitution(SEXP hap_set_ptr, const uint64& hap_ind, const uint64& chrom_ind, const char& nucleo_, const uint64& new_pos_) {
    BuilderPtr b(hap_set_ptr);
    check(jk_add_substitution(b.get(), hap_ind, chrom_ind, nucleo_, new_pos_));
}

//[[Rcpp::export]]
void add_insertion(SEXP hap_set_ptr, const uint64& hap_ind, const uint64& chrom_ind, const std::string& nucleos_, const uint64& new_pos_) {
    BuilderPtr b(hap_set_ptr);
    check(jk_add_insertion(b.get(), hap_ind, chrom_ind, nucleos_.c_str(), new_pos_));
}

//[[Rcpp::export]]
void add_deletion(SEXP hap_set_ptr, const uint64& hap_ind, const uint64& chrom_ind, const uint64& size_, const uint64& new_pos_) {
    BuilderPtr b(hap_set_ptr);
    check(jk_add_deletion(b.get(), hap_ind, chrom_ind, size_, new_pos_));
}
