/*
 * jackalope_hip.h -- C ABI of libjackalope_hip.so: jackalope's HTS read-generation hot path
 * (illumina() / pacbio()) on AMD MI355X (gfx950).
 *
 * This is the drop-in boundary.  Each entry point replaces one C++ function of the reference
 * (reference = lucasnell/jackalope v1.1.6; file:line relative to its root) and is what an Rcpp
 * shim keeping the reference's RcppExports signatures would bind (INTEGRATION.md shows that shim).
 * Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 *   reference function                                  replaced by
 *   --------------------------------------------------  -----------------------------------------
 *   illumina_ref_cpp   src/hts_illumina.cpp:589-649     jk_illumina_ref
 *   illumina_hap_cpp   src/hts_illumina.cpp:662-739     jk_illumina_hap
 *   pacbio_ref_cpp     src/hts_pacbio.cpp:579-640       jk_pacbio_ref
 *   pacbio_hap_cpp     src/hts_pacbio.cpp:646-715       jk_pacbio_hap
 *   write_reads_cpp_ / write_reads_one_filetype_        jk_session_* (plan, generate, fetch, write)
 *                      src/hts.h:323-500
 *   mt_seeds / seeded_pcg (R RNG contract)              jk_seed_source
 *                      src/pcg.h:37-85
 *   make_hap_set, add_substitution / add_insertion /    jk_hap_builder_new, jk_add_substitution / _insertion /
 *   add_deletion       src/ref_hap_access.cpp:127,816   _deletion, jk_hap_builder_view
 *   FileBGZF / bgzip_file  src/io.h:150, src/hts.h:140  jk_bgzf_deflate (and comp_method "bgzip" of the sessions)
 *   create_genome_cpp  src/create_sequences.cpp:151     jk_create_genome  (genome stays in device memory)
 *   read_fasta_noind / read_fasta_ind                   jk_read_fasta
 *                      src/io_fasta.cpp:153, :389
 *
 * Semantics kept from the reference: `n_threads` is the number of independent generator streams
 * ("lanes").  Lane t behaves exactly like OpenMP thread t of the reference: its own pcg64 seeded
 * from 8 sub-seed words, its own read quota from split_int(), its own per-chromosome quotas from
 * reads_per_group(), its own gamma-distribution state.  The reference caps n_threads at
 * omp_get_max_threads() (src/util.h:197-208); here it is the GPU's parallelism and is typically
 * 2^16..2^20.  Output order is lane-major (all reads of lane 0, then lane 1, ...), which is one of
 * the interleavings the reference's `#pragma omp critical` pool flush (src/hts.h:401-412) can
 * produce; with n_threads = 1 it is the reference's exact single-thread output.
 *
 * Errors: every function returns JK_OK (0) or a nonzero jk_status; jk_last_error() gives the
 * message (thread-local).  The reference throws Rcpp::exception (src/util.h:172-176); the shim
 * re-throws jk_last_error() through Rcpp::stop.
 */
#ifndef JACKALOPE_HIP_H
#define JACKALOPE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum jk_status {
    JK_OK = 0,
    JK_ERR_ARG = 1,          /* invalid argument (message mirrors the reference's stop() text where one exists) */
    JK_ERR_UNSUPPORTED = 2,  /* valid for the reference, not implemented on the GPU path (fails loudly) */
    JK_ERR_DEVICE = 3,       /* HIP runtime error / no GPU */
    JK_ERR_IO = 4,           /* output file could not be opened/written (src/io.h:288-290) */
    JK_ERR_SEEDS = 5,        /* seed source exhausted */
    JK_ERR_ABORTED = 6       /* abort flag raised (Progress::check_abort, src/hts.h:396-399) */
} jk_status;

/* View of a RefGenome (src/ref_classes.h:127-180).  Borrowed for the duration of the call. */
typedef struct jk_ref_genome {
    uint64_t n_chroms;
    const char* const* chrom_names;   /* RefChrom::name   */
    const char* const* chrom_seqs;    /* RefChrom::nucleos, chrom_lens[i] bytes, need not be NUL-terminated */
    const uint64_t* chrom_lens;
    const char* name;                 /* RefGenome::name, "REF" (src/ref_classes.h:138); NULL = "REF" */
    int32_t seqs_on_device;           /* 0: chrom_seqs are host pointers; 1: device pointers (jk_genome_view) */
} jk_ref_genome;

/* View of a HapSet (src/hap_classes.h:500-611): for haplotype h and chromosome c, cell = h*n_chroms+c.
 * Mutations of all cells are concatenated in cell order; mutation m of the set has old_pos[m],
 * new_pos[m] (AllMutations, src/hap_classes.h:100-104) and inserted/substituted bytes
 * nuc_blob[nuc_off[m] .. nuc_off[m+1]) (empty = deletion, the reference's nullptr). */
typedef struct jk_hap_set {
    uint64_t n_haps, n_chroms;
    const char* const* hap_names;     /* HapGenome::name, "hap0".. */
    jk_ref_genome ref;
    const uint64_t* chrom_size;       /* [n_haps*n_chroms] HapChrom::chrom_size */
    const uint64_t* n_mut;            /* [n_haps*n_chroms] */
    const uint64_t* old_pos;          /* [sum n_mut] */
    const uint64_t* new_pos;          /* [sum n_mut] */
    const uint64_t* nuc_off;          /* [sum n_mut + 1] */
    const char* nuc_blob;
} jk_hap_set;

/* qual_probs / quals of one read end as illumina_ref_cpp receives them
 * (vector<vector<vector<double>>>, [nt T,C,A,G][position][k]; src/hts_illumina.cpp:604-609), flattened. */
typedef struct jk_illumina_profile {
    uint32_t read_length;
    const uint32_t* n_quals;          /* [4*read_length], nt-major */
    const double* probs;              /* flat, same order */
    const uint8_t* quals;             /* flat, same order (uint8 = unsigned char, src/jackalope_types.h:26) */
} jk_illumina_profile;

/* Where the 32-bit sub-seed words come from.  The reference draws them from R's RNG with
 * Rcpp::runif(8, 0, 4294967296) truncated to integers (src/pcg.h:37-46,63-71), in this order:
 * 8 per lane for all lanes (mt_seeds, src/hts.h:339), then 8 per reads_per_group() call that has
 * n_reads > 0 (src/hts.h:62-64), lane by lane (src/hts.h:349-353).  Either give the words up front
 * (`words`/`n_words`) or a callback that fills the next 8. */
typedef int (*jk_seed_fn)(void* user, uint32_t* out8);
typedef struct jk_seed_source {
    const uint32_t* words;
    uint64_t n_words;
    jk_seed_fn fn;
    void* user;
} jk_seed_source;

/* Arguments of illumina_ref_cpp / illumina_hap_cpp, same names and meaning
 * (src/hts_illumina.cpp:589-612, :662-687). */
typedef struct jk_illumina_args {
    int32_t paired;
    int32_t matepair;
    const char* out_prefix;           /* files <prefix>_R1.fq[, <prefix>_R2.fq] (src/hts.h:342-346) */
    int32_t sep_files;                /* hap only: one pair of files per haplotype (src/hts.h:512-552) */
    int32_t compress;                 /* 0 = plain, 1..9 = level; files get ".gz" appended (src/io.h) */
    const char* comp_method;          /* "bgzip" (BGZF made on the device, default), "bgzip-host" (BGZF, zlib on the host) or "gzip" */
    uint64_t n_reads;
    double prob_dup;
    uint64_t n_threads;               /* number of lanes, see header comment */
    int32_t show_progress;            /* accepted, ignored (no progress bar) */
    uint64_t read_pool_size;
    const double* haplotype_probs;    /* hap only, [n_haps] */
    double frag_len_shape, frag_len_scale;
    uint64_t frag_len_min, frag_len_max;
    jk_illumina_profile profile1; double ins_prob1, del_prob1;
    jk_illumina_profile profile2; double ins_prob2, del_prob2;   /* ignored unless paired */
    const char* const* barcodes;      /* ref: barcodes[0]; hap: one per haplotype (padded with "") */
    uint64_t n_barcodes;
    jk_seed_source seeds;
    const volatile int32_t* abort_flag;   /* may be NULL; polled between batches */
    /* Sharding for multi-GPU runs: this process generates lanes [lane_begin, lane_end) of the
     * n_threads lanes (all seeds/quotas are still derived for every lane, so every rank agrees).
     * lane_end = 0 means n_threads. */
    uint64_t lane_begin, lane_end;
    int32_t device;                   /* HIP device ordinal */
    uint64_t max_batch_bytes;         /* cap on device memory for one batch's read pools; 0 = default */
    /* Set-up in O(own lanes) for multi-rank runs (seed words given as an array only).  By default a shard derives
     * the seed-word position of its first lane by planning the haplotype-level split of every lane before it.
     * A host that knows the position -- ranks exchange the word counts jk_session_shard_seed_words() reports, the
     * "seed-offset reduction" of the multi-GPU design -- passes it here and only this shard's lanes are planned. */
    int32_t seed_offset_given;
    uint64_t seed_offset_words;       /* position (in 32-bit words, from the start of `seeds.words`) of lane_begin's first add_n_reads word */
    /* One-shot entry points only (jk_illumina_ref / jk_illumina_hap): generate on several devices of this node, one
     * host thread per device, device k taking the k-th contiguous block of lanes; the files are the same as with one
     * device.  n_devices = 0: `device` alone.  (The reference's analogue: OpenMP threads over one shared sink,
     * src/hts.h:357-417.) */
    const int32_t* devices;
    uint32_t n_devices;
    /* Session API: 1 = this session streams (jk_session_run): every generator launch's FASTQ goes to the sink as it
     * completes -- device BGZF when asked for, pinned double-buffered copy to the host, file writes on writer threads,
     * all overlapped with the next launch -- and no image stays in device memory, so a run is not limited by it
     * (the reference flushes pool by pool, src/hts.h:401-412).  jk_session_generate / _fetch / _write are then not
     * available.  The one-shot entry points always stream. */
    int32_t stream_output;
} jk_illumina_args;

/* Arguments of pacbio_ref_cpp / pacbio_hap_cpp, same names and meaning (src/hts_pacbio.cpp:579-602, :646-671). */
typedef struct jk_pacbio_args {
    const char* out_prefix;           /* file <prefix>_R1.fq */
    int32_t sep_files;                /* hap only */
    int32_t compress;                 /* 0 = plain, 1..9 = level */
    const char* comp_method;
    uint64_t n_reads;
    uint64_t n_threads;               /* number of lanes */
    int32_t show_progress;
    uint64_t read_pool_size;
    double prob_dup;
    double scale, sigma, loc;         /* lognormal read lengths (used when n_read_lens == 0) */
    double min_read_len;
    const double* read_probs;         /* custom read lengths: sampling weights, [n_read_lens] */
    const uint64_t* read_lens;        /*                      lengths */
    uint64_t n_read_lens;
    uint64_t max_passes;
    const double* chi2_params_n;      /* [3] */
    const double* chi2_params_s;      /* [5] */
    const double* sqrt_params;        /* [2] */
    const double* norm_params;        /* [2] */
    double prob_thresh, prob_ins, prob_del, prob_subst;
    const double* haplotype_probs;    /* hap only, [n_haps]; NULL = all 1 */
    jk_seed_source seeds;
    const volatile int32_t* abort_flag;
    uint64_t lane_begin, lane_end;
    int32_t device;
    uint64_t max_batch_bytes;
    int32_t seed_offset_given;        /* as in jk_illumina_args */
    uint64_t seed_offset_words;
    const int32_t* devices;           /* as in jk_illumina_args (jk_pacbio_ref / jk_pacbio_hap) */
    uint32_t n_devices;
    int32_t stream_output;            /* as in jk_illumina_args */
} jk_pacbio_args;

const char* jk_last_error(void);
const char* jk_version(void);
int jk_device_count(void);                /* MI355X devices visible to this process (0 when there is none) */
/* The device arena: large device buffers (read pools, FASTQ images, genome) of a closed session stay allocated and are
   reused by the next session of the process -- the next haplotype of write_reads_cpp_sep_files_'s loop
   (/root/reference/src/hts.h:512-552 opens a writer per haplotype), the next call from R -- instead of going through the
   driver's allocator, which clears memory it hands out.  Parked memory is released when an allocation needs it, on
   jk_device_arena_trim (device < 0: every device) and at process exit; JK_ARENA=0 in the environment turns the arena off. */
void jk_device_arena_trim(int device);
void jk_device_arena_stats(int device, uint64_t* parked_bytes, uint64_t* hits, uint64_t* misses);

/* One-shot entry points: generate and write the FASTQ files, like the reference's functions. */
int jk_illumina_ref(const jk_ref_genome* genome, const jk_illumina_args* args);
int jk_illumina_hap(const jk_hap_set* haps, const jk_illumina_args* args);
int jk_pacbio_ref(const jk_ref_genome* genome, const jk_pacbio_args* args);
int jk_pacbio_hap(const jk_hap_set* haps, const jk_pacbio_args* args);

/* Jobs: a one-shot call in two steps per output file set, for hosts whose RNG and interrupt handling live on one
 * thread (R).  A file set is the pair <prefix>_R1/_R2 -- one per call, or one per haplotype with sep_files
 * (write_reads_cpp_sep_files_, src/hts.h:512-552).  jk_job_plan_next does, on the calling thread, what the reference
 * does before its parallel region for the next file set -- it is the only step that reads seed words (mt_seeds, then
 * the add_n_reads of every thread, src/hts.h:339,349-353) -- and jk_job_run then generates and writes that file set on
 * args.devices from any thread, while another thread may poll jk_job_progress (the reference's progress bar,
 * src/hts.h:414) and raise *args.abort_flag (Progress::check_abort, src/hts.h:396-399).  The genome / haplotype views
 * and everything the args point to are borrowed until jk_job_free.  jk_illumina_ref & co. are exactly this loop. */
typedef struct jk_job jk_job;
int jk_illumina_ref_job(const jk_ref_genome* genome, const jk_illumina_args* args, jk_job** out);
int jk_illumina_hap_job(const jk_hap_set* haps, const jk_illumina_args* args, jk_job** out);
int jk_pacbio_ref_job(const jk_ref_genome* genome, const jk_pacbio_args* args, jk_job** out);
int jk_pacbio_hap_job(const jk_hap_set* haps, const jk_pacbio_args* args, jk_job** out);
uint32_t jk_job_n_files(const jk_job* j);
int jk_job_plan_next(jk_job* j);
int jk_job_run(jk_job* j);
int jk_job_progress(const jk_job* j, uint64_t* reads_done, uint64_t* reads_total);
uint64_t jk_job_seed_words_used(const jk_job* j);
void jk_job_free(jk_job* j);

/* Session API (what the one-shot calls are made of; used by tests and bench.py so that the
 * generated FASTQ can stay resident in HBM). */
typedef struct jk_session jk_session;

int jk_illumina_ref_open(const jk_ref_genome* genome, const jk_illumina_args* args, jk_session** out);
int jk_illumina_hap_open(const jk_hap_set* haps, const jk_illumina_args* args, jk_session** out);
int jk_pacbio_ref_open(const jk_ref_genome* genome, const jk_pacbio_args* args, jk_session** out);
int jk_pacbio_hap_open(const jk_hap_set* haps, const jk_pacbio_args* args, jk_session** out);
/* Run every batch of this session's lanes: generator kernel, per-lane byte-count scan, pool
 * compaction into the lane-major FASTQ images.  May be called repeatedly (same output each time). */
int jk_session_generate(jk_session* s);
/* The same in two halves: _generate_async queues one pass over the session's lanes and returns, _wait completes the
   oldest queued pass (sizes, timing and the FASTQ image are then those of that pass).  Up to two passes may be queued:
   the generator launches of the second run beside the last compaction of the first, which is what a caller with job
   after job to run (the per-haplotype loop of /root/reference/src/hts.h:512-552, repeated illumina() calls) wants.
   Illumina sessions without stream_output. */
int jk_session_generate_async(jk_session* s);
int jk_session_wait(jk_session* s);
/* Streaming sessions (args.stream_output): run every batch and write <out_prefix>_R<e+1>.fq[.gz] while generating.
 * out_prefix "" (or NULL) = null sink: the FASTQ (or its BGZF form when compress > 0) is brought to host memory and
 * dropped, which is what bench.py times as the D2H-inclusive rate.  Blocks until the files are closed; another thread
 * may poll jk_session_progress meanwhile and raise *args.abort_flag to stop it (JK_ERR_ABORTED). */
int jk_session_run(jk_session* s);
/* Reads whose FASTQ has left the device so far / reads planned for this session's lanes.  Thread-safe. */
int jk_session_progress(const jk_session* s, uint64_t* reads_done, uint64_t* reads_total);
/* bytes[e] = FASTQ bytes of read end e (e < n_ends); reads = reads made (all ends). */
int jk_session_sizes(const jk_session* s, uint64_t bytes[2], uint64_t* reads, uint32_t* n_ends);
/* Device pointer of the FASTQ image of read end e (valid until the next generate/close). */
int jk_session_device_ptr(const jk_session* s, uint32_t end, const void** dptr);
/* Copy the FASTQ image of read end e to host memory (cap >= bytes[e]). */
int jk_session_fetch(const jk_session* s, uint32_t end, void* dst, uint64_t cap);
/* Copy bytes [byte_off, byte_off + n) of the FASTQ image of read end e to host memory: lane windows of an image
 * that is too large to fetch whole (the images of BASELINE configs[2..4] are 100-200 GB). */
int jk_session_fetch_range(const jk_session* s, uint32_t end, uint64_t byte_off, uint64_t n, void* dst);
/* Write <out_prefix>_R<e+1>.fq[.gz] for every end: FileUncomp / FileGZ / FileBGZF of src/io.h:58-295,
 * chosen by args.compress and args.comp_method.  Compression runs on the host after generation, as the
 * reference does for n_threads > 1 (src/hts.h:478-490). */
int jk_session_write(const jk_session* s);
/* A lane shard (lane_begin/lane_end a proper sub-range) holds only its part of the files: jk_session_write refuses
 * it; this call writes the shard's image of read end e at byte file_offset[e] of <out_prefix>_R<e+1>.fq, creating the
 * file if needed and never truncating it (ranks sharing a prefix write disjoint ranges; offsets = exclusive prefix of
 * the ranks' jk_session_sizes, the count exchange of the multi-GPU design).  Uncompressed output only. */
int jk_session_write_shard(const jk_session* s, const uint64_t file_offset[2]);
/* Timing of the last generate(): HIP-event milliseconds on the session's stream.
 * ms[0] generator kernel(s), ms[1] scan + compaction kernels, ms[2] whole generate() (device). */
int jk_session_timing(const jk_session* s, double ms[3]);
/* Number of generator launches (batches of lanes) one generate() makes. */
uint32_t jk_session_batches(const jk_session* s);
/* PacBio: how often the last generate() re-planned its buffers and ran again (a lane's pool or the FASTQ image, both
 * sized from the read-length model, turned out too small; the kernels never write past either). */
uint32_t jk_session_retries(const jk_session* s);
/* Diagnostic (Illumina sessions): the generator decides `u < Prob[i]` of the alias step (src/alias_sampler.h:53-60) from the
 * high 32 bits of the exact 64-bit cut point and reads the low 32 only when the draw's high word EQUALS the cut point's --
 * once per 2^32 draws.  The lanes (relative to the shard) where that happened since the session was opened are noted, up to
 * 61 of them; *n is the number of times it happened.  tests/test_gpu_full_size.py compares exactly those lanes with the
 * oracle. */
int jk_session_rare_branch_lanes(const jk_session* s, uint32_t* n, uint32_t* lanes, uint32_t cap);
/* Number of sub-seed words consumed while opening the session. */
uint64_t jk_session_seed_words_used(const jk_session* s);
/* Positions [begin, end) in the seed-word stream of the add_n_reads words of this shard's lanes (after the
 * n_threads * 8 words of mt_seeds): what ranks exchange to give each other args.seed_offset_words. */
int jk_session_shard_seed_words(const jk_session* s, uint64_t* begin, uint64_t* end);
/* Per-lane counts, for the multi-GPU count/offset exchange: n = lane_end - lane_begin entries. */
int jk_session_lane_bytes(const jk_session* s, uint32_t end, uint64_t* out, uint64_t n);
void jk_session_close(jk_session* s);

/* Host-side pieces of the path that the shim or tests may want on their own. */
void jk_split_int(uint64_t x, uint64_t n, uint64_t* out);                         /* src/util.h:245-258 */
int jk_reads_per_group(uint64_t n_reads, const double* probs, uint64_t n,         /* src/hts.h:58-103 */
                       jk_seed_source* seeds, uint64_t* out);
void jk_alias_build(const double* probs, uint64_t n, double* Prob, uint64_t* Alias);   /* src/alias_sampler.h:68-106 */
/* What the sessions do for `n_threads` lanes before any kernel runs (write_reads_one_filetype_, src/hts.h:334-353):
 * split_int of the reads, mt_seeds, then per lane the filler's add_n_reads -- reference genome: reads_per_group over
 * the chromosomes; haplotypes (hap != 0): reads_per_group over haplotype_probs, per haplotype over its chromosomes,
 * then each read maker's own add_n_reads (which sees half the count when maker_halves, the paired Illumina case).
 * chrom_probs: [n_haps or 1][n_chroms] chromosome sizes.  Outputs for lanes [lane_begin, lane_end) (0 = n_threads):
 * lane_seeds [n_shard * 8], quotas [cell][lane of the shard] in reads (all ends), words3 = {seed words consumed,
 * first and one-past-last position of the shard's add_n_reads words}.  Host only. */
int jk_plan_lane_quotas(int32_t hap, uint32_t n_ends, int32_t maker_halves, const double* hap_probs, uint64_t n_haps,
                        const double* chrom_probs, uint64_t n_chroms, uint64_t n_reads, uint64_t n_threads,
                        uint64_t lane_begin, uint64_t lane_end, jk_seed_source* seeds, int32_t offset_given, uint64_t offset_words,
                        uint32_t* lane_seeds, uint32_t* quotas, uint64_t* words3);
int jk_hap_chrom_full(const jk_hap_set* haps, uint64_t hap, uint64_t chrom, char* out, uint64_t cap); /* src/hap_classes.cpp:80-116 (host) */
/* pcg64 seeded from 8 words (src/pcg.h:48-85), jumped `steps` outputs ahead (engine::advance,
 * inst/include/pcg/pcg_random.hpp:419-434), then n outputs: the jump tables of jk_create_genome (host) */
void jk_pcg_advance_outputs(const uint32_t* words8, uint64_t steps, uint64_t n, uint64_t* out);

/* create_genome (third "next" row): create_genome_cpp / create_chromosomes_
 * (src/create_sequences.cpp:59-169) with the same arguments -- n_chroms chromosomes "chrom0".., lengths
 * from a gamma distribution with the given mean and sd (all = len_mean when len_sd is 0), bases drawn
 * with equilibrium frequencies pi_tcag[T,C,A,G]; n_threads engines, chromosomes dealt to them in
 * contiguous blocks as `omp for schedule(static)` does.  Seeds: 8 words per thread (mt_seeds,
 * src/pcg.h:63-71).  The genome is made in device memory and stays there: jk_genome_view gives a
 * jk_ref_genome whose chrom_seqs are device pointers (seqs_on_device = 1) for jk_illumina_ref /
 * jk_pacbio_ref; jk_genome_fetch copies one chromosome to the host.  The view borrows from the handle. */
typedef struct jk_genome jk_genome;
int jk_create_genome(uint64_t n_chroms, double len_mean, double len_sd, const double* pi_tcag, uint64_t n_threads,
                     jk_seed_source* seeds, int device, jk_genome** out);
int jk_genome_view(jk_genome* g, jk_ref_genome* view);
int jk_genome_fetch(const jk_genome* g, uint64_t chrom, char* dst, uint64_t cap);
/* read_fasta (fourth "next" row): read_fasta_noind / read_fasta_ind
 * (src/io_fasta.cpp:153-169, :389-408; R/read_write.R:24-52).  Files may be plain, gzip or bgzip (zlib's
 * gzread, as in the reference).  fai_files NULL = non-indexed: a line containing '>' starts a
 * chromosome (cut_names: name up to the first space), sequence lines lose '\n' and one '\r' before
 * it; with index files the spans they list are read and only '\n' is dropped.  Every sequence byte
 * goes through the reference's filter (TCAGN kept, tcagn upper-cased when remove_soft_mask, anything
 * else becomes a zero byte).  The host reads the file and locates header lines; newline removal,
 * filtering and packing run on the device, where the genome stays (same handle as jk_create_genome). */
int jk_read_fasta(const char* const* fasta_files, const char* const* fai_files, uint64_t n_files, int32_t cut_names,
                  int32_t remove_soft_mask, int device, jk_genome** out);
uint64_t jk_genome_seed_words_used(const jk_genome* g);
double jk_genome_ms(const jk_genome* g);          /* device milliseconds of the generating kernel */
void jk_genome_free(jk_genome* g);

/* BGZF compression of a byte image that is already in device memory (second
 * "next" row): the device-side replacement of FileBGZF / bgzip_file (src/io.h:150-236, src/hts.h:140-180).
 * d_src (16-byte aligned) and d_dst are DEVICE pointers; d_dst needs jk_bgzf_bound(n) bytes.  The result
 * is a complete BGZF file image (blocks of <= 0xff00 input bytes, then the end-of-file block): one
 * dynamic-Huffman DEFLATE block of literals per BGZF block, or a stored block where that is smaller.
 * jk_session_write uses it when comp_method is "bgzip"; "bgzip-host" deflates the same blocks with zlib
 * on the host at the requested level, "gzip" writes one gzip stream on the host (FileGZ, src/io.h:58-147).
 * ms (may be NULL) receives the device time of the compression kernels. */
uint64_t jk_bgzf_bound(uint64_t n);
int jk_bgzf_deflate(int device, const void* d_src, uint64_t n, void* d_dst, uint64_t cap, uint64_t* out_bytes, double* ms);

/* Mutation tables, write side (first "next" row): the haplotype objects the
 * sequencers read are built by these three edits.  A builder is the reference's XPtr<HapSet>; the
 * functions replace make_hap_set (src/ref_hap_access.cpp:127-132) and add_substitution /
 * add_insertion / add_deletion (src/ref_hap_access.cpp:816-865 -> HapChrom::add_*,
 * src/hap_classes.cpp:295-509), with the same 0-based indices and argument meaning.  Host code: the
 * tables live in host memory until a sequencer session uploads them.  The chromosome bytes of `ref`
 * are borrowed and must outlive the builder (as the RefGenome must outlive a HapSet); everything
 * else of `ref` is copied. */
typedef struct jk_hap_builder jk_hap_builder;
int jk_hap_builder_new(const jk_ref_genome* ref, uint64_t n_haps, jk_hap_builder** out);
/* Start from existing tables (e.g. read back from another tool) instead of an unmutated set. */
int jk_hap_builder_from(const jk_hap_set* haps, jk_hap_builder** out);
int jk_add_substitution(jk_hap_builder* b, uint64_t hap_ind, uint64_t chrom_ind, char nucleo, uint64_t new_pos);
int jk_add_insertion(jk_hap_builder* b, uint64_t hap_ind, uint64_t chrom_ind, const char* nucleos, uint64_t new_pos);
int jk_add_deletion(jk_hap_builder* b, uint64_t hap_ind, uint64_t chrom_ind, uint64_t size, uint64_t new_pos);
/* Flat view of the current tables for jk_*_hap / jk_hap_chrom_full; valid until the next edit or free. */
int jk_hap_builder_view(jk_hap_builder* b, jk_hap_set* out);
void jk_hap_builder_free(jk_hap_builder* b);

/* Elementary arithmetic of the path evaluated by the SAME inline functions the kernels use, on the
 * host (`jk_host_*`) and on the device (`jk_dev_*`, arrays of n inputs already in host memory; they
 * are copied to the GPU, evaluated there by one thread per element and copied back).  These exist
 * so that tests can compare each primitive with the oracle.  `what`: */
enum {
    JK_OP_PCG_STREAM = 0,   /* in: 8 seed words per stream (as u64[8] each), out: `aux` outputs per stream */
    JK_OP_RUNIF_INDEX = 1,  /* in: x, aux = n            -> (uint64)(runif_01 * n)          */
    JK_OP_RUNIF_DOUBLE = 2, /* in: x                     -> bits of (double)runif_01         */
    JK_OP_CANONICAL = 3,    /* in: x                     -> bits of generate_canonical       */
    JK_OP_N_QUAL = 4,       /* in: x                     -> runif_01*10 + '!' as uint8       */
    JK_OP_LT_HALF = 5,      /* in: x                     -> runif_01 < 0.5                   */
    JK_OP_FRAG_START = 6,   /* in: x, aux = span         -> (uint64)(u * span)               */
    JK_OP_LOG = 7,          /* in: bits of a double      -> bits of log                      */
    JK_OP_SQRT = 8,         /* in: bits of a double      -> bits of sqrt                     */
    JK_OP_GAMMA_STREAM = 9, /* in: 8 seed words per stream, out: `aux` gamma draws (bits); shape/scale via jk_eval_set_gamma */
    JK_OP_EXP = 10,         /* in: bits of x            -> bits of exp(x), ~0 if outside the supported range */
    JK_OP_POW = 11,         /* in: (bits x, bits y)     -> bits of pow(x, y), ~0 if outside the main path */
    JK_OP_LOG10 = 12,       /* in: bits of x            -> bits of log10(x) */
    JK_OP_QNORM = 13,       /* in: bits of p            -> bits of qnorm(p, 0, 1) (AS 241) */
    JK_OP_RUNIF_AB = 14,    /* in: (x, bits a, c.m, c.e) -> bits of (double)(a + runif_01 * c), c = b - a in x87 */
    JK_OP_RUNIF_INDEX32 = 15, /* as RUNIF_INDEX through the kernels' 32-bit routine (jk_dev_eval; n < 2^32) */
    JK_OP_ALIAS_INDEX32 = 16  /* as RUNIF_INDEX through the quality step's routine (jk_dev_eval; n <= 255) */
};
int jk_host_eval(int what, const uint64_t* in, uint64_t n, uint64_t aux, uint64_t* out);
int jk_dev_eval(int device, int what, const uint64_t* in, uint64_t n, uint64_t aux, uint64_t* out);
void jk_eval_set_gamma(double shape, double scale);
void jk_x87_one_minus(double p, uint64_t* m, int32_t* e);   /* 1 - p in x87 extended precision */

#ifdef __cplusplus
}
#endif
#endif /* JACKALOPE_HIP_H */
