// jk_api.hip -- the one translation unit of libjackalope_hip.so: the C ABI of include/jackalope_hip.h.
//
//   kernels      jk_illumina_kernel.h  jk_pacbio_kernel.h  jk_bgzf_kernel.h  jk_genome_kernel.h  jk_fasta_kernel.h
//   arithmetic   jk_math.h  jk_math2.h  jk_log_data.h (device + host)   jk_nmath.h  jk_host.h  jk_haps.h (host)
//   host driver  jk_common.h  jk_session.h
//                api_illumina.h  api_pacbio.h   set-up of a run (the GPU counterpart of what write_reads_cpp_ /
//                                               write_reads_one_filetype_ do before their parallel region,
//                                               reference src/hts.h:323-500)
//                api_sinks.h                    plain / gzip / BGZF sinks, BGZF on the device; pinned copy-out pipe
//                api_launch.h                   one pass over the batches: generator, scan, compaction on two streams;
//                                               resident (generate) or streamed to the sinks (run)
//                api_job.h                      a whole call: file sets (sep_files), devices, progress
//                api_eval.h                     primitive evaluation hooks for the tests
//   C ABI        this file (sessions, one-shot calls, BGZF, eval), api_builder.h (host helpers, mutation-table
//                builder), api_genome.h (create_genome, read_fasta)
#include <hip/hip_runtime.h>

#include <cerrno>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <zlib.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <memory>
#include <string>
#include <vector>

#include "../../include/jackalope_hip.h"
#include "jk_haps.h"
#include "jk_host.h"
#include "jk_plan.h"
#include "jk_illumina_kernel.h"
#include "jk_math2.h"
#include "jk_nmath.h"
#include "jk_pacbio_kernel.h"
#include "jk_bgzf_kernel.h"
#include "jk_genome_kernel.h"
#include "jk_fasta_kernel.h"
#include "jk_plan_kernel.h"
#include "jk_common.h"
#include "jk_session.h"
#include "api_illumina.h"
#include "api_pacbio.h"
#include "api_sinks.h"
#include "api_launch.h"
#include "api_eval.h"

using namespace jk;

namespace jk {
// PacBio pools and images are sized from the read-length model; a run that outgrows them is planned again, larger
template <typename Run>
static void with_replan(jk_session& s, Run run) {
    s.retries = 0;
    for (int attempt = 0;; attempt++) {
        try { run(); return; }
        catch (const Error& e) {
            const bool again = (e.code == JK_ERR_RETRY || e.code == JK_ERR_RETRY_IMAGE) && attempt < 6 && s.replan;
            if (!again) {
                if (e.code == JK_ERR_RETRY) throw Error(JK_ERR_DEVICE, "PacBio pools overflowed even after growing them");
                if (e.code == JK_ERR_RETRY_IMAGE) throw Error(JK_ERR_DEVICE, "the PacBio FASTQ image overflowed even after growing it");
                throw;
            }
            if (e.code == JK_ERR_RETRY) s.pool_scale *= 2; else s.image_scale *= 2;
            s.retries++;
            s.replan();
        }
    }
}
static void generate_with_retry(jk_session& s) { with_replan(s, [&] { launch_generate(s); }); }
static void stream_with_retry(jk_session& s, const std::string& suffix, bool with_eof) { with_replan(s, [&] { launch_stream(s, suffix, with_eof); }); }
}  // namespace jk
#include "api_job.h"

extern "C" {

const char* jk_last_error(void) { return g_last_error.c_str(); }
const char* jk_version(void) { return "jackalope_hip 0.2 (gfx950)"; }
int jk_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }
void jk_device_arena_trim(int device) { DevArena::get().trim(device); }
void jk_device_arena_stats(int device, uint64_t* bytes, uint64_t* hits, uint64_t* misses) {
    DevArena& a = DevArena::get();
    if (bytes) *bytes = a.bytes(device);
    std::lock_guard<std::mutex> l(a.m);
    if (hits) *hits = a.hits;
    if (misses) *misses = a.misses;
}
#ifdef JK_TIMELINE
int jk_debug_timeline(uint64_t* out, uint64_t n_words) {     // experiment builds only
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(jk::g_timeline), n_words * 8) == hipSuccess ? 0 : 1;
}
#endif

int jk_illumina_ref_open(const jk_ref_genome* genome, const jk_illumina_args* args, jk_session** out) {
    return guarded([&] {
        if (!genome || !args || !out) throw Error(JK_ERR_ARG, "NULL argument");
        std::unique_ptr<jk_session> s(new jk_session());
        SeedReader seeds{args->seeds};
        open_illumina_ref(*s, *genome, *args, seeds);
        *out = s.release();
    });
}

static std::vector<double> hap_probs_of(const jk_hap_set& hs, const jk_illumina_args& a) {
    // R passes rep(1, n_haps) when haplotype_probs is NULL (R/hts_illumina.R:665-667)
    if (!a.haplotype_probs) return std::vector<double>(hs.n_haps, 1.0);
    return std::vector<double>(a.haplotype_probs, a.haplotype_probs + hs.n_haps);
}

int jk_illumina_hap_open(const jk_hap_set* haps, const jk_illumina_args* args, jk_session** out) {
    return guarded([&] {
        if (!haps || !args || !out) throw Error(JK_ERR_ARG, "NULL argument");
        if (args->sep_files) throw Error(JK_ERR_UNSUPPORTED, "sep_files needs one session per haplotype: use jk_illumina_hap");
        std::unique_ptr<jk_session> s(new jk_session());
        SeedReader seeds{args->seeds};
        open_illumina_hap(*s, *haps, *args, hap_probs_of(*haps, *args), args->n_reads, seeds);
        *out = s.release();
    });
}

static std::vector<double> pb_hap_probs_of(const jk_hap_set& hs, const jk_pacbio_args& a) {
    if (!a.haplotype_probs) return std::vector<double>(hs.n_haps, 1.0);
    return std::vector<double>(a.haplotype_probs, a.haplotype_probs + hs.n_haps);
}

int jk_pacbio_ref_open(const jk_ref_genome* genome, const jk_pacbio_args* args, jk_session** out) {
    return guarded([&] {
        if (!genome || !args || !out) throw Error(JK_ERR_ARG, "NULL argument");
        if (genome->n_chroms == 0) throw Error(JK_ERR_ARG, "reference genome has no chromosomes");
        std::unique_ptr<jk_session> s(new jk_session());
        SeedReader seeds{args->seeds};
        open_pacbio_ref(*s, *genome, *args, seeds);
        *out = s.release();
    });
}

int jk_pacbio_hap_open(const jk_hap_set* haps, const jk_pacbio_args* args, jk_session** out) {
    return guarded([&] {
        if (!haps || !args || !out) throw Error(JK_ERR_ARG, "NULL argument");
        if (args->sep_files) throw Error(JK_ERR_UNSUPPORTED, "sep_files needs one session per haplotype: use jk_pacbio_hap");
        std::unique_ptr<jk_session> s(new jk_session());
        SeedReader seeds{args->seeds};
        open_pacbio_hap(*s, *haps, *args, pb_hap_probs_of(*haps, *args), args->n_reads, seeds);
        *out = s.release();
    });
}

int jk_session_generate(jk_session* s) {
    return guarded([&] { if (!s) throw Error(JK_ERR_ARG, "NULL session"); generate_with_retry(*s); });
}

int jk_session_generate_async(jk_session* s) {
    return guarded([&] { if (!s) throw Error(JK_ERR_ARG, "NULL session"); launch_generate_async(*s); });
}
int jk_session_wait(jk_session* s) {
    return guarded([&] { if (!s) throw Error(JK_ERR_ARG, "NULL session"); launch_wait(*s); });
}

int jk_session_sizes(const jk_session* s, uint64_t bytes[2], uint64_t* reads, uint32_t* n_ends) {
    return guarded([&] {
        if (!s || !(s->generated || s->streamed)) throw Error(JK_ERR_ARG, "session has not generated yet");
        if (bytes) { bytes[0] = s->bytes[0]; bytes[1] = s->bytes[1]; }
        if (reads) *reads = s->reads_made;
        if (n_ends) *n_ends = s->n_ends;
    });
}

int jk_session_device_ptr(const jk_session* s, uint32_t end, const void** dptr) {
    return guarded([&] {
        if (!s || !s->generated || end >= s->n_ends || !dptr) throw Error(JK_ERR_ARG, "bad session/end");
        *dptr = s->d_out[end].p;
    });
}

int jk_session_fetch(const jk_session* s, uint32_t end, void* dst, uint64_t cap) {
    return guarded([&] {
        if (!s || !s->generated || end >= s->n_ends) throw Error(JK_ERR_ARG, "bad session/end");
        if (cap < s->bytes[end]) throw Error(JK_ERR_ARG, "destination too small");
        JK_HIP(hipSetDevice(s->device));
        if (s->bytes[end]) JK_HIP(hipMemcpy(dst, s->d_out[end].p, s->bytes[end], hipMemcpyDeviceToHost));
    });
}

int jk_session_fetch_range(const jk_session* s, uint32_t end, uint64_t byte_off, uint64_t n, void* dst) {
    return guarded([&] {
        if (!s || !s->generated || end >= s->n_ends) throw Error(JK_ERR_ARG, "bad session/end");
        if (byte_off > s->bytes[end] || n > s->bytes[end] - byte_off) throw Error(JK_ERR_ARG, "range outside the FASTQ image");
        if (n && !dst) throw Error(JK_ERR_ARG, "NULL destination");
        JK_HIP(hipSetDevice(s->device));
        if (n) JK_HIP(hipMemcpy(dst, s->d_out[end].as<uint8_t>() + byte_off, n, hipMemcpyDeviceToHost));
    });
}

int jk_session_write_shard(const jk_session* s, const uint64_t file_offset[2]) {
    return guarded([&] {
        if (!s || !file_offset) throw Error(JK_ERR_ARG, "NULL argument");
        JK_HIP(hipSetDevice(s->device));
        write_shard(*s, file_offset);
    });
}

int jk_session_shard_seed_words(const jk_session* s, uint64_t* begin, uint64_t* end) {
    return guarded([&] {
        if (!s) throw Error(JK_ERR_ARG, "NULL session");
        if (begin) *begin = s->shard_seed_begin;
        if (end) *end = s->shard_seed_end;
    });
}

int jk_session_write(const jk_session* s) {
    return guarded([&] { if (!s) throw Error(JK_ERR_ARG, "NULL session"); JK_HIP(hipSetDevice(s->device)); write_files(*s); });
}

int jk_session_timing(const jk_session* s, double ms[3]) {
    return guarded([&] {
        if (!s || !(s->generated || s->streamed)) throw Error(JK_ERR_ARG, "session has not generated yet");
        ms[0] = s->ms[0]; ms[1] = s->ms[1]; ms[2] = s->ms[2];
    });
}

int jk_session_rare_branch_lanes(const jk_session* s, uint32_t* n, uint32_t* lanes, uint32_t cap) {
    return guarded([&] {
        if (!s || !n) throw Error(JK_ERR_ARG, "NULL argument");
        if (s->pacbio) throw Error(JK_ERR_ARG, "the rare-branch log belongs to the Illumina generator");
        JK_HIP(hipSetDevice(s->device));
        uint32_t log[1 + JK_RARE_LOG_CAP];
        JK_HIP(hipDeviceSynchronize());
        JK_HIP(hipMemcpy(log, s->d_err.as<uint32_t>() + 2, sizeof log, hipMemcpyDeviceToHost));
        *n = log[0];
        for (uint32_t i = 0; i < cap && i < log[0] && i < JK_RARE_LOG_CAP; i++) lanes[i] = log[1 + i];
    });
}

uint64_t jk_session_seed_words_used(const jk_session* s) { return s ? s->seed_words_used : 0; }
uint32_t jk_session_retries(const jk_session* s) { return s ? s->retries : 0; }
uint32_t jk_session_batches(const jk_session* s) { return s ? (uint32_t)s->batches.size() : 0; }

int jk_session_lane_bytes(const jk_session* s, uint32_t end, uint64_t* out, uint64_t n) {
    return guarded([&] {
        if (!s || !s->generated || end >= s->n_ends || n != s->n_shard) throw Error(JK_ERR_ARG, "bad session/end/count");
        JK_HIP(hipSetDevice(s->device));
        if (n) JK_HIP(hipMemcpy(out, s->d_lane_bytes[end].p, n * 8, hipMemcpyDeviceToHost));
    });
}

void jk_session_close(jk_session* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    delete s;
}

// ---- jobs and the one-shot entry points ------------------------------------------------------------------------
static int make_job(jk_job::Kind kind, const jk_ref_genome* genome, const jk_hap_set* haps, const jk_illumina_args* ia,
                    const jk_pacbio_args* pa, jk_job** out) {
    return guarded([&] {
        if ((!genome && !haps) || (!ia && !pa) || !out) throw Error(JK_ERR_ARG, "NULL argument");
        std::unique_ptr<jk_job> j(new jk_job());
        j->kind = kind; j->genome = genome; j->haps = haps;
        if (ia) { j->ia = *ia; j->seeds = SeedReader{ia->seeds}; job_common_init(*j, ia->n_threads, ia->devices, ia->n_devices, ia->device, ia->sep_files); }
        else { j->pa = *pa; j->seeds = SeedReader{pa->seeds}; job_common_init(*j, pa->n_threads, pa->devices, pa->n_devices, pa->device, pa->sep_files); }
        if ((ia && (ia->lane_begin || ia->lane_end)) || (pa && (pa->lane_begin || pa->lane_end)))
            throw Error(JK_ERR_ARG, "lane shards belong to the session API; a job (and a one-shot call) covers all lanes, over `devices`");
        job_start(*j);
        *out = j.release();
    });
}
int jk_illumina_ref_job(const jk_ref_genome* genome, const jk_illumina_args* args, jk_job** out) { return make_job(jk_job::ILL_REF, genome, nullptr, args, nullptr, out); }
int jk_illumina_hap_job(const jk_hap_set* haps, const jk_illumina_args* args, jk_job** out) { return make_job(jk_job::ILL_HAP, nullptr, haps, args, nullptr, out); }
int jk_pacbio_ref_job(const jk_ref_genome* genome, const jk_pacbio_args* args, jk_job** out) { return make_job(jk_job::PB_REF, genome, nullptr, nullptr, args, out); }
int jk_pacbio_hap_job(const jk_hap_set* haps, const jk_pacbio_args* args, jk_job** out) { return make_job(jk_job::PB_HAP, nullptr, haps, nullptr, args, out); }
uint32_t jk_job_n_files(const jk_job* j) { return j ? j->n_files : 0; }
int jk_job_plan_next(jk_job* j) { return guarded([&] { if (!j) throw Error(JK_ERR_ARG, "NULL job"); job_plan_next(*j); }); }
int jk_job_run(jk_job* j) { return guarded([&] { if (!j) throw Error(JK_ERR_ARG, "NULL job"); job_run(*j); }); }
int jk_job_progress(const jk_job* j, uint64_t* reads_done, uint64_t* reads_total) {
    return guarded([&] {
        if (!j) throw Error(JK_ERR_ARG, "NULL job");
        std::lock_guard<std::mutex> l(j->m);
        uint64_t d = j->done_before.load();
        for (const jk_session* s : j->live) d += s->progress_done.load();
        if (reads_done) *reads_done = d;
        if (reads_total) *reads_total = j->total_reads;
    });
}
uint64_t jk_job_seed_words_used(const jk_job* j) { return j ? j->seeds.pos : 0; }
void jk_job_free(jk_job* j) { delete j; }

static int one_shot(jk_job::Kind kind, const jk_ref_genome* genome, const jk_hap_set* haps, const jk_illumina_args* ia, const jk_pacbio_args* pa) {
    jk_job* j = nullptr;
    int rc = make_job(kind, genome, haps, ia, pa, &j);
    for (uint32_t f = 0; rc == JK_OK && f < j->n_files; f++) {
        const volatile int32_t* af = ia ? ia->abort_flag : pa->abort_flag;
        if (af && *af) { g_last_error = "aborted"; rc = JK_ERR_ABORTED; break; }
        rc = jk_job_plan_next(j);
        if (rc == JK_OK) rc = jk_job_run(j);
    }
    const std::string keep = g_last_error;
    jk_job_free(j);
    g_last_error = keep;
    return rc;
}
int jk_illumina_ref(const jk_ref_genome* genome, const jk_illumina_args* args) { return one_shot(jk_job::ILL_REF, genome, nullptr, args, nullptr); }
int jk_illumina_hap(const jk_hap_set* haps, const jk_illumina_args* args) { return one_shot(jk_job::ILL_HAP, nullptr, haps, args, nullptr); }
int jk_pacbio_ref(const jk_ref_genome* genome, const jk_pacbio_args* args) { return one_shot(jk_job::PB_REF, genome, nullptr, nullptr, args); }
int jk_pacbio_hap(const jk_hap_set* haps, const jk_pacbio_args* args) { return one_shot(jk_job::PB_HAP, nullptr, haps, nullptr, args); }

int jk_session_run(jk_session* s) {
    return guarded([&] {
        if (!s) throw Error(JK_ERR_ARG, "NULL session");
        if (s->n_shard != s->n_lanes_total)
            throw Error(JK_ERR_UNSUPPORTED, "a lane shard cannot stream into the run's files: its place in them depends on the other shards "
                        "(generate resident and use jk_session_write_shard, or run the whole call as a job over `devices`)");
        stream_with_retry(*s, "", true);
    });
}
int jk_session_progress(const jk_session* s, uint64_t* reads_done, uint64_t* reads_total) {
    return guarded([&] {
        if (!s) throw Error(JK_ERR_ARG, "NULL session");
        if (reads_done) *reads_done = s->progress_done.load();
        if (reads_total) *reads_total = s->progress_total;
    });
}

uint64_t jk_bgzf_bound(uint64_t n) { return bgzf_bound(n); }

int jk_bgzf_deflate(int device, const void* d_src, uint64_t n, void* d_dst, uint64_t cap, uint64_t* out_bytes, double* ms) {
    return guarded([&] {
        if ((n && !d_src) || !d_dst || !out_bytes) throw Error(JK_ERR_ARG, "NULL pointer");
        JK_HIP(hipSetDevice(device));
        *out_bytes = bgzf_deflate_device(device, nullptr, static_cast<const uint8_t*>(d_src), n, static_cast<uint8_t*>(d_dst), cap, ms);
    });
}

void jk_eval_set_gamma(double shape, double scale) { g_eval_shape = shape; g_eval_scale = scale; }

// 1 - p in x87 extended precision (what `b - a` of runif_ab(eng, p, 1) is), as significand and exponent
void jk_x87_one_minus(double p, uint64_t* m, int32_t* e) {
    const long double c = 1.0L - static_cast<long double>(p);
    int ex = 0;
    const long double fr = frexpl(c, &ex);              // c = fr * 2^ex, fr in [0.5, 1)
    *m = c > 0 ? static_cast<uint64_t>(ldexpl(fr, 64)) : 0;
    *e = ex - 64;
}

int jk_host_eval(int what, const uint64_t* in, uint64_t n, uint64_t aux, uint64_t* out) {
    return guarded([&] {
        const jk_gamma_param gp = eval_gamma_param();
        for (uint64_t i = 0; i < n; i++) eval_one(what, in, i, aux, out, gp);
    });
}

int jk_dev_eval(int device, int what, const uint64_t* in, uint64_t n, uint64_t aux, uint64_t* out) {
    return guarded([&] {
        JK_HIP(hipSetDevice(device));
        uint64_t n_in, n_out;
        eval_sizes(what, n, aux, &n_in, &n_out);
        DevBuf din, dout;
        din.alloc(n_in * 8); dout.alloc(n_out * 8);
        JK_HIP(hipMemcpy(din.p, in, n_in * 8, hipMemcpyHostToDevice));
        const uint32_t block = 256;
        const uint32_t grid = (uint32_t)((n + block - 1) / block);
        hipLaunchKernelGGL(eval_kernel, dim3(grid ? grid : 1), dim3(block), 0, 0, what, din.as<uint64_t>(), n, aux, dout.as<uint64_t>(), eval_gamma_param());
        JK_HIP(hipGetLastError());
        JK_HIP(hipDeviceSynchronize());
        JK_HIP(hipMemcpy(out, dout.p, n_out * 8, hipMemcpyDeviceToHost));
    });
}

}  // extern "C"

#include "api_builder.h"
#include "api_genome.h"
