// api_job.h -- a whole illumina() / pacbio() call: its output file sets, planned on the calling thread (that is where
// R's RNG lives) and run on any thread over one or several devices
// (part of the one translation unit jk_api.hip; see the include list there)
//
// The reference's entry points (src/hts_illumina.cpp:589-739, src/hts_pacbio.cpp:579-715) end in write_reads_cpp_ or,
// with sep_files, in write_reads_cpp_sep_files_ (src/hts.h:441-552): one run per haplotype with one-hot haplotype
// probabilities and the prefix <out_prefix>_<haplotype>, the reads per file drawn first by one reads_per_group call.
// A job is that loop: jk_job_plan_next() does for the next file set what the reference does before its parallel
// region (seed words, quotas), jk_job_run() generates and writes it.  The one-shot calls run the loop themselves.
#pragma once

#include <sys/sendfile.h>

struct jk_job {
    enum Kind { ILL_REF, ILL_HAP, PB_REF, PB_HAP } kind = ILL_REF;
    const jk_ref_genome* genome = nullptr;
    const jk_hap_set* haps = nullptr;
    jk_illumina_args ia{};
    jk_pacbio_args pa{};
    jk::SeedReader seeds{};
    std::vector<int> devices;
    // file sets
    uint32_t n_files = 1, next_file = 0;
    bool sep_files = false;
    std::vector<uint64_t> per_file;            // reads per file set (sep_files)
    std::vector<double> hap_probs;             // of the whole call
    // the file set planned last
    bool planned = false;
    std::vector<double> cur_probs;
    uint64_t cur_reads = 0;
    std::string cur_prefix;
    uint64_t T = 1;
    jk::LanePlan plan;
    // progress of the running file set
    mutable std::mutex m;
    std::vector<jk_session*> live;
    uint64_t total_reads = 0;
    std::atomic<uint64_t> done_before{0};      // reads of the file sets already written

    bool is_hap() const { return kind == ILL_HAP || kind == PB_HAP; }
    bool is_pacbio() const { return kind == PB_REF || kind == PB_HAP; }
};

namespace jk {

static void job_common_init(jk_job& j, uint64_t n_threads, const int32_t* devices, uint32_t n_devices, int32_t device, int32_t sep_files) {
    j.T = n_threads ? n_threads : 1;
    if (n_devices > 0 && !devices) throw Error(JK_ERR_ARG, "n_devices > 0 but devices is NULL");
    if (n_devices > 64) throw Error(JK_ERR_ARG, "too many devices");
    for (uint32_t k = 0; k < n_devices; k++) j.devices.push_back(devices[k]);
    if (j.devices.empty()) j.devices.push_back(device);
    if ((uint64_t)j.devices.size() > j.T) j.devices.resize((size_t)j.T);       // at least one lane per device
    j.sep_files = j.is_hap() && sep_files != 0;
    j.n_files = j.sep_files ? (uint32_t)j.haps->n_haps : 1u;
}

// the part of a call that precedes its first file set: with sep_files the reads per file (src/hts.h:526-529)
static void job_start(jk_job& j) {
    const uint64_t n_reads = j.is_pacbio() ? j.pa.n_reads : j.ia.n_reads;
    j.total_reads = n_reads;
    if (j.is_hap()) {
        const double* hp = j.is_pacbio() ? j.pa.haplotype_probs : j.ia.haplotype_probs;
        // R passes rep(1, n_haps) when haplotype_probs is NULL (R/hts_illumina.R:665-667)
        j.hap_probs = hp ? std::vector<double>(hp, hp + j.haps->n_haps) : std::vector<double>(j.haps->n_haps, 1.0);
    }
    if (j.sep_files) {
        const uint64_t n_ends = (!j.is_pacbio() && j.ia.paired) ? 2 : 1;
        j.per_file = reads_per_group(n_reads / n_ends, j.hap_probs, j.seeds);
        for (uint64_t& v : j.per_file) v *= n_ends;
        j.total_reads = 0;
        for (uint64_t v : j.per_file) j.total_reads += v;
    }
}

static void job_plan_next(jk_job& j) {
    if (j.next_file >= j.n_files) throw Error(JK_ERR_ARG, "every file set of this job has been planned already");
    const uint32_t f = j.next_file;
    const std::string prefix = j.is_pacbio() ? (j.pa.out_prefix ? j.pa.out_prefix : "") : (j.ia.out_prefix ? j.ia.out_prefix : "");
    if (j.sep_files) {
        j.cur_probs.assign(j.haps->n_haps, 0.0);
        j.cur_probs[f] = 1;
        j.cur_reads = j.per_file[f];
        j.cur_prefix = prefix.empty() ? prefix : prefix + "_" + (j.haps->hap_names ? j.haps->hap_names[f] : "");
    } else {
        j.cur_probs = j.hap_probs;
        j.cur_reads = j.is_pacbio() ? j.pa.n_reads : j.ia.n_reads;
        j.cur_prefix = prefix;
    }
    const uint32_t n_ends = (!j.is_pacbio() && j.ia.paired) ? 2u : 1u;
    std::vector<uint64_t> per_lane = split_int(j.cur_reads / n_ends, j.T);
    for (uint64_t& v : per_lane) v *= n_ends;
    QuotaModel Q;
    switch (j.kind) {
        case jk_job::ILL_REF: Q = quota_model_ref(*j.genome, n_ends); break;
        case jk_job::PB_REF: Q = quota_model_ref(*j.genome, 1); break;
        case jk_job::ILL_HAP: Q = quota_model_hap(*j.haps, j.cur_probs, n_ends, j.ia.paired != 0); break;
        case jk_job::PB_HAP: Q = quota_model_hap(*j.haps, j.cur_probs, 1, false); break;
    }
    if (Q.n_chroms == 0) throw Error(JK_ERR_ARG, "reference genome has no chromosomes");
    j.plan = plan_lane_quotas(Q, per_lane, 0, j.T, j.seeds, false, 0, defer_splits());
    j.planned = true;
    j.next_file++;
}

// append `src` to `dst` (parts of one output file written by different devices), then remove it
static void append_file(const std::string& dst, const std::string& src) {
    struct Fd { int fd = -1; ~Fd() { if (fd >= 0) ::close(fd); } } in, out;
    in.fd = ::open(src.c_str(), O_RDONLY);
    out.fd = ::open(dst.c_str(), O_WRONLY);       // (sendfile refuses O_APPEND descriptors: seek to the end instead)
    if (in.fd < 0 || out.fd < 0) throw Error(JK_ERR_IO, "Unable to open file " + (in.fd < 0 ? src : dst) + ".\n");
    struct stat st;
    if (::fstat(in.fd, &st) != 0 || ::lseek(out.fd, 0, SEEK_END) < 0) throw Error(JK_ERR_IO, "stat of " + src + " / seek in " + dst + " failed");
    off_t left = st.st_size;
    bool use_sendfile = true;
    std::vector<char> buf;
    while (left > 0) {
        ssize_t n;
        if (use_sendfile) {
            n = ::sendfile(out.fd, in.fd, nullptr, (size_t)std::min<off_t>(left, (off_t)1 << 30));
            if (n < 0 && (errno == EINVAL || errno == ENOSYS)) { use_sendfile = false; continue; }    // file system without it: copy through a buffer
        } else {
            if (buf.empty()) buf.resize(8u << 20);
            n = ::read(in.fd, buf.data(), (size_t)std::min<off_t>(left, (off_t)buf.size()));
            for (ssize_t done = 0; n > 0 && done < n;) {
                const ssize_t w = ::write(out.fd, buf.data() + done, (size_t)(n - done));
                if (w < 0) { if (errno == EINTR) continue; n = -1; break; }
                done += w;
            }
        }
        if (n < 0) { if (errno == EINTR) continue; throw Error(JK_ERR_IO, "appending " + src + " to " + dst + " failed: " + std::strerror(errno)); }
        if (n == 0) break;
        left -= n;
    }
    if (left != 0) throw Error(JK_ERR_IO, "appending " + src + " to " + dst + " came up short");
    const int fd = out.fd; out.fd = -1;
    if (::close(fd) != 0) throw Error(JK_ERR_IO, "error closing " + dst);
    ::unlink(src.c_str());
}

// one device's share of the file set: open a streaming session on lanes [lo, hi) and run it
static void job_run_device(jk_job& j, size_t k, uint64_t lo, uint64_t hi, const std::string& suffix, bool with_eof) {
    std::unique_ptr<jk_session> s(new jk_session());
    SeedReader none{};                   // the plan exists already: no seed is read here
    if (j.is_pacbio()) {
        jk_pacbio_args a = j.pa;
        a.out_prefix = j.cur_prefix.c_str(); a.n_reads = j.cur_reads; a.lane_begin = lo; a.lane_end = hi;
        a.device = j.devices[k]; a.stream_output = 1; a.sep_files = 0;
        if (j.kind == jk_job::PB_REF) open_pacbio_ref(*s, *j.genome, a, none, &j.plan);
        else open_pacbio_hap(*s, *j.haps, a, j.cur_probs, j.cur_reads, none, &j.plan);
    } else {
        jk_illumina_args a = j.ia;
        a.out_prefix = j.cur_prefix.c_str(); a.n_reads = j.cur_reads; a.lane_begin = lo; a.lane_end = hi;
        a.device = j.devices[k]; a.stream_output = 1; a.sep_files = 0;
        if (j.kind == jk_job::ILL_REF) open_illumina_ref(*s, *j.genome, a, none, &j.plan);
        else open_illumina_hap(*s, *j.haps, a, j.cur_probs, j.cur_reads, none, &j.plan);
    }
    struct Live {
        jk_job& j; jk_session* s;
        Live(jk_job& j_, jk_session* s_) : j(j_), s(s_) { std::lock_guard<std::mutex> l(j.m); j.live.push_back(s); }
        ~Live() {
            std::lock_guard<std::mutex> l(j.m);
            j.done_before.fetch_add(s->progress_done.load());
            j.live.erase(std::find(j.live.begin(), j.live.end(), s));
        }
    } live(j, s.get());
    stream_with_retry(*s, suffix, with_eof);
}

static void job_run(jk_job& j) {
    if (!j.planned) throw Error(JK_ERR_ARG, "jk_job_run before jk_job_plan_next");
    j.planned = false;
    const size_t D = j.devices.size();
    if (D == 1) { job_run_device(j, 0, 0, j.T, "", true); return; }
    // contiguous lane blocks, as even as possible (the rule split_int applies to reads); device k > 0 writes
    // <file>.part<k>, appended to device 0's file in order when all are done
    std::vector<uint64_t> cut(D + 1, 0);
    for (size_t k = 0; k < D; k++) cut[k + 1] = cut[k] + j.T / D + (k < j.T % D ? 1 : 0);
    std::vector<std::string> errs(D);
    std::vector<int> codes(D, 0);
    std::vector<std::thread> pool;
    for (size_t k = 0; k < D; k++)
        pool.emplace_back([&, k] {
            try { job_run_device(j, k, cut[k], cut[k + 1], k ? ".part" + std::to_string(k) : "", k + 1 == D); }
            catch (const Error& e) { errs[k] = e.what(); codes[k] = e.code; }
            catch (const std::exception& e) { errs[k] = e.what(); codes[k] = JK_ERR_DEVICE; }
        });
    for (std::thread& t : pool) t.join();
    const uint32_t n_ends = (!j.is_pacbio() && j.ia.paired) ? 2u : 1u;
    const int compress = j.is_pacbio() ? j.pa.compress : j.ia.compress;
    auto file_of = [&](uint32_t e) { return j.cur_prefix + "_R" + std::to_string(e + 1) + ".fq" + (compress > 0 ? ".gz" : ""); };
    for (size_t k = 0; k < D; k++)
        if (codes[k]) {
            if (!j.cur_prefix.empty())
                for (uint32_t e = 0; e < n_ends; e++) for (size_t q = 1; q < D; q++) ::unlink((file_of(e) + ".part" + std::to_string(q)).c_str());
            throw Error(codes[k], "device " + std::to_string(j.devices[k]) + ": " + errs[k]);
        }
    if (j.cur_prefix.empty()) return;          // null sink
    for (uint32_t e = 0; e < n_ends; e++)
        for (size_t k = 1; k < D; k++) append_file(file_of(e), file_of(e) + ".part" + std::to_string(k));
}

static void job_run_all(jk_job& j) {
    job_start(j);
    for (uint32_t f = 0; f < j.n_files; f++) {
        const volatile int32_t* af = j.is_pacbio() ? j.pa.abort_flag : j.ia.abort_flag;
        if (af && *af) throw Error(JK_ERR_ABORTED, "aborted");
        job_plan_next(j);
        job_run(j);
    }
}

}  // namespace jk
