// jk_api.hip -- C ABI (include/jackalope_hip.h) of the MI355X read-generation path: host driver
// (the GPU counterpart of write_reads_cpp_ / write_reads_one_filetype_, reference src/hts.h:323-500)
// around the kernels in jk_illumina_kernel.h.
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
#include "jk_illumina_kernel.h"
#include "jk_math2.h"
#include "jk_nmath.h"
#include "jk_pacbio_kernel.h"
#include "jk_bgzf_kernel.h"
#include "jk_genome_kernel.h"
#include "jk_fasta_kernel.h"

namespace jk {

static thread_local std::string g_last_error;

#define JK_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            throw Error(JK_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_));        \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t n = 0;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
    void alloc(size_t bytes) {
        release();
        if (bytes == 0) bytes = 16;
        JK_HIP(hipMalloc(&p, bytes));
        n = bytes;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
    template <typename T> void upload(const std::vector<T>& v) {
        alloc(v.size() * sizeof(T));
        if (!v.empty()) JK_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    }
};

constexpr int JK_ERR_RETRY = 1000;   // internal: PacBio pools were too small, regenerate with larger ones

static inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }
static inline uint32_t n_digits(uint64_t v) { uint32_t d = 1; while (v >= 10) { v /= 10; d++; } return d; }

struct Batch {
    uint64_t lane0;        // first lane (relative to the shard)
    uint32_t n_lanes;
    uint64_t pool_bytes;   // per read end
};

}  // namespace jk

#ifndef JK_ILL_BLOCK
#define JK_ILL_BLOCK 1024     // threads per generator workgroup (one workgroup per CU when the tables sit in LDS)
#endif

using namespace jk;

struct jk_session {
    int device = 0;
    hipStream_t stream = nullptr;      // generator kernels
    hipStream_t cp_stream = nullptr;   // scan + compaction of the previous batch, overlapping the next one
    hipStream_t stream2 = nullptr;     // Illumina: generator launches of odd batches (see launch_generate)
    bool two_gen_streams = false;
    uint32_t n_ends = 1;
    bool paired = false;
    std::string out_prefix;
    // genome
    DevBuf d_seq, d_chrom_off, d_chrom_len, d_hdr_blob, d_hdr_off;
    uint32_t n_chroms = 0;
    // tables
    IlluminaTables tables;
    DevBuf d_info, d_thresh, d_quals, d_mm;
    bool lds_tables = false;
    size_t lds_bytes = 0, lds_launch = 0, evw_set = 0;
    uint32_t lds_seg_off = 0;
    bool hap = false;
    int compress = 0;          // 0 = plain FASTQ, 1..9 = compression level
    bool bgzip = true;         // comp_method: "bgzip" (BGZF blocks) or "gzip"
    bool host_deflate = false; // comp_method "bgzip-host": BGZF blocks deflated by zlib on the host at level `compress`
    bool pacbio = false;
    PacbioKernelParams kpb{};
    DevBuf d_len_thresh, d_len_alias, d_lens, d_thr_tab, d_pass_tab, d_ev2;
    uint32_t ev_words = 0;
    uint64_t nuc_base = 0;     // offset of the haplotypes' nucleotide blob inside d_seq
    double pool_scale = 1.25;  // PacBio: pool capacity relative to the expected bytes (grown on overflow)
    std::function<void()> replan;   // PacBio: re-plan pools after pool_scale changed
    DevBuf d_cell_off, d_new_pos, d_ref_shift, d_nuc_len, d_nuc_off, d_cell_size, d_bc_blob, d_bc_len;
    // lanes of this shard
    uint64_t n_lanes_total = 0, lane_begin = 0, lane_end = 0, n_shard = 0;
    std::vector<uint64_t> pool_off_host;          // per batch-relative offsets, concatenated per batch (n+1 each)
    DevBuf d_seeds, d_lane_reads, d_chrom_reads, d_pool_off;
    std::vector<Batch> batches;
    std::vector<uint64_t> batch_pool_off_index;   // index into d_pool_off of each batch's first entry
    DevBuf d_pool[2][2] /* [ping-pong][end] */, d_out[2], d_lane_bytes[2], d_lane_off[2], d_block_sums, d_base[2], d_lane_made, d_evw, d_err;
    uint64_t out_cap = 0;
    IlluminaKernelParams kp{};                    // template, per-batch fields filled at launch
    // results of the last generate()
    uint64_t bytes[2] = {0, 0};
    uint64_t reads_made = 0;
    double ms[3] = {0, 0, 0};
    bool generated = false;
    uint64_t seed_words_used = 0;
    const volatile int32_t* abort_flag = nullptr;
    std::vector<hipEvent_t> events;       // [0] start, [1+2b] / [2+2b] around generator b, last = end
    std::vector<hipEvent_t> gen_done, cp_done;   // per batch, for the two-stream hand-off

    ~jk_session() {
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        for (hipEvent_t e : gen_done) (void)hipEventDestroy(e);
        for (hipEvent_t e : cp_done) (void)hipEventDestroy(e);
        if (stream) (void)hipStreamDestroy(stream);
        if (cp_stream) (void)hipStreamDestroy(cp_stream);
        if (stream2) (void)hipStreamDestroy(stream2);
    }
};

namespace jk {

// Chromosomes (+ optionally the haplotypes' nucleotide blob) into one encoded device buffer.
static void upload_genome(jk_session& s, const jk_ref_genome& g, const char* blob_bytes, uint64_t blob_len) {
    if (g.n_chroms == 0) throw Error(JK_ERR_ARG, "reference genome has no chromosomes");
    if (g.n_chroms > 0xffffffffULL) throw Error(JK_ERR_UNSUPPORTED, "too many chromosomes");
    std::vector<uint64_t> off(g.n_chroms), len(g.n_chroms);
    uint64_t total = 64;
    for (uint64_t i = 0; i < g.n_chroms; i++) { off[i] = total; len[i] = g.chrom_lens[i]; total = align_up(total + len[i], 64) + 64; }
    s.nuc_base = total;
    total = align_up(total + blob_len, 64) + 64;
    s.d_seq.alloc(total);
    JK_HIP(hipMemset(s.d_seq.p, 'N', total));
    if (blob_len) JK_HIP(hipMemcpy(s.d_seq.as<uint8_t>() + s.nuc_base, blob_bytes, blob_len, hipMemcpyHostToDevice));
    for (uint64_t i = 0; i < g.n_chroms; i++)
        if (len[i]) JK_HIP(hipMemcpy(s.d_seq.as<uint8_t>() + off[i], g.chrom_seqs[i], len[i],
                                     g.seqs_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    // T,C,A,G -> 0..3, everything else -> 4 (what nt_map / cmp_map of the reference distinguish)
    DevBuf bad; bad.alloc(4);
    JK_HIP(hipMemset(bad.p, 0, 4));
    hipLaunchKernelGGL(encode_bases_kernel, dim3(2048), dim3(256), 0, 0, s.d_seq.as<uint8_t>(), total, bad.as<uint32_t>());
    JK_HIP(hipGetLastError());
    JK_HIP(hipDeviceSynchronize());
    uint32_t bad_h = 0;
    JK_HIP(hipMemcpy(&bad_h, bad.p, 4, hipMemcpyDeviceToHost));
    if (bad_h) throw Error(JK_ERR_UNSUPPORTED, "the genome contains bytes 0xfc-0xff, which the GPU path cannot represent");
    s.d_chrom_off.upload(off);
    s.d_chrom_len.upload(len);
    s.n_chroms = (uint32_t)g.n_chroms;
}

// compress / comp_method of the reference's entry points (write_reads_cpp_, src/hts.h:453-496)
static void set_compression(jk_session& s, int compress, const char* comp_method) {
    if (compress < 0 || compress > 9) throw Error(JK_ERR_ARG, "\nInvalid bgzip compress level of " + std::to_string(compress) + ". It must be in range [0,9].");
    s.compress = compress;
    const std::string m = comp_method ? comp_method : "bgzip";
    if (compress > 0 && m != "gzip" && m != "bgzip" && m != "bgzip-host") throw Error(JK_ERR_ARG, "\nUnrecognized compression method.");
    s.bgzip = (m != "gzip");
    s.host_deflate = (m == "bgzip-host");
}

static inline uint8_t encode_base(char c) { return c == 'T' ? 0 : c == 'C' ? 1 : c == 'A' ? 2 : c == 'G' ? 3 : 4; }

// ---- pieces shared by the reference-genome and haplotype entry points --------------------------

// Argument checks + error-model tables + every per-run constant of the kernel.
static void setup_model(jk_session& s, const jk_illumina_args& a) {
    set_compression(s, a.compress, a.comp_method);
    if (a.frag_len_shape < 1.0) throw Error(JK_ERR_UNSUPPORTED, "frag_len_shape < 1 (fragment sd > mean) is not implemented on the GPU path");
    if (!(a.frag_len_scale > 0)) throw Error(JK_ERR_ARG, "frag_len_scale must be > 0");
    s.paired = a.paired != 0;
    s.n_ends = s.paired ? 2 : 1;
    s.out_prefix = a.out_prefix ? a.out_prefix : "";
    s.abort_flag = a.abort_flag;
    s.device = a.device;
    JK_HIP(hipSetDevice(s.device));
    JK_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    JK_HIP(hipStreamCreateWithFlags(&s.cp_stream, hipStreamNonBlocking));
    JK_HIP(hipStreamCreateWithFlags(&s.stream2, hipStreamNonBlocking));
    if (const char* e = std::getenv("JK_TWO_GEN_STREAMS")) s.two_gen_streams = std::atoi(e) != 0;

    s.tables = build_illumina_tables(a);
    const uint32_t L = s.tables.read_length;
    s.ev_words = (2 * L + 63) / 64 + 1;
    if (s.ev_words > (uint32_t)JK_MAX_EVW) throw Error(JK_ERR_UNSUPPORTED, "read lengths above 480 are not implemented on the GPU path");

    IlluminaKernelParams& P = s.kp;
    P.read_len = L; P.n_ends = s.n_ends; P.paired = s.paired; P.matepair = (s.paired && a.matepair) ? 1 : 0;
    P.ev_words = s.ev_words;
    P.frag_min = a.frag_len_min; P.frag_max = a.frag_len_max;
    {   // gamma_distribution<double>::param_type::_M_initialize (random.tcc:2330-2346), alpha >= 1
        const double a1 = a.frag_len_shape - 1.0 / 3.0;
        P.gp.a1 = a1;
        P.gp.a2 = 1.0 / std::sqrt(9.0 * a1);
        P.gp.beta = a.frag_len_scale;
    }
    const double insp[2] = {a.ins_prob1, a.ins_prob2}, delp[2] = {a.del_prob1, a.del_prob2};
    for (uint32_t r = 0; r < 2; r++) {
        // u > (ins + del) -> match ; else u > ins -> deletion ; else insertion (hts_illumina.cpp:133-144)
        Threshold tm = threshold_le(insp[r] + delp[r]);
        Threshold td = threshold_le(insp[r]);
        P.th_match[r] = tm.th; P.never_match[r] = tm.all;
        P.th_del[r] = td.th; P.never_del[r] = td.all;
    }
    {   // dup < prob_dup (src/hts.h:265-266)
        Threshold t = threshold_lt(a.prob_dup);
        P.th_dup = t.th; P.dup_all = t.all;
    }
    P.pool_size = a.read_pool_size;
}

static void check_barcode(const std::string& bc, uint32_t L) {
    if (bc.size() > (size_t)JK_MAX_BARCODE) throw Error(JK_ERR_UNSUPPORTED, "barcodes longer than 32 bases are not implemented on the GPU path");
    if (bc.size() >= L) throw Error(JK_ERR_ARG, "barcode must be shorter than the read length");
}

// Lanes of the run and of this process's shard; per-lane read quotas (src/hts.h:334-336).
static std::vector<uint64_t> plan_lanes(jk_session& s, uint64_t n_threads, uint64_t lane_begin, uint64_t lane_end, uint64_t n_reads) {
    uint64_t T = n_threads ? n_threads : 1;
    s.n_lanes_total = T;
    s.lane_begin = lane_begin;
    s.lane_end = lane_end ? lane_end : T;
    if (s.lane_begin > s.lane_end || s.lane_end > T) throw Error(JK_ERR_ARG, "lane shard out of range");
    s.n_shard = s.lane_end - s.lane_begin;
    std::vector<uint64_t> per_lane = split_int(n_reads / s.n_ends, T);
    for (uint64_t& v : per_lane) v *= s.n_ends;
    if (per_lane[0] > 0xffffffffULL) throw Error(JK_ERR_UNSUPPORTED, "more than 2^32 reads per lane: raise n_threads");
    return per_lane;
}

// mt_seeds (src/pcg.h:37-46): 8 words per lane for ALL lanes, in lane order; keep this shard's.
static std::vector<uint32_t> take_lane_seeds(jk_session& s, SeedReader& seeds) {
    std::vector<uint32_t> lane_seeds(s.n_shard * 8);
    uint32_t w[8];
    for (uint64_t t = 0; t < s.n_lanes_total; t++) {
        seeds.take8(w);
        if (t >= s.lane_begin && t < s.lane_end) std::memcpy(&lane_seeds[(t - s.lane_begin) * 8], w, sizeof(w));
    }
    return lane_seeds;
}

// Pools: tiles of 64 lanes (one wave), every lane of a tile gets the capacity of the tile's largest
// quota of maximal records; a batch is a run of whole tiles.  Then all device buffers.
// lane_cap[l] = pool bytes lane l may need.  Plans batches/tiles and allocates everything that does not
// depend on the sequencer model.  Returns the largest number of lanes in a batch.
static uint32_t plan_pools_common(jk_session& s, uint64_t max_batch_bytes, uint64_t lanes_per_batch,
                                  const std::vector<uint64_t>& lane_cap, const std::vector<uint64_t>& lane_reads,
                                  const std::vector<uint32_t>& lane_seeds, const std::vector<uint32_t>& quotas) {
    s.batches.clear(); s.batch_pool_off_index.clear();
    const uint64_t max_batch = max_batch_bytes ? max_batch_bytes : (8ULL << 30);
    uint64_t max_batch_lanes = lanes_per_batch;
    if (const char* e = std::getenv("JK_BATCH_LANES")) { const long long v = std::atoll(e); if (v >= 64) max_batch_lanes = (uint64_t)v / 64 * 64; }
    std::vector<uint64_t> pool_off;
    uint64_t out_cap = 0, max_pool = 0;
    uint32_t max_lanes = 0;
    uint64_t l = 0;
    while (l < s.n_shard) {
        Batch b{l, 0, 0};
        s.batch_pool_off_index.push_back(pool_off.size());
        pool_off.push_back(0);
        uint64_t used = 0;
        while (l < s.n_shard && b.n_lanes < max_batch_lanes) {
            const uint64_t tl = std::min<uint64_t>(64, s.n_shard - l);
            uint64_t mx = 0;
            for (uint64_t k = 0; k < tl; k++) mx = std::max(mx, lane_cap[l + k]);
            const uint64_t cap = align_up(mx, 4) * 64;
            if (b.n_lanes > 0 && used + cap > max_batch) break;
            used += cap; pool_off.push_back(used); b.n_lanes += (uint32_t)tl; l += tl;
        }
        b.pool_bytes = used;
        out_cap += used;
        max_pool = std::max(max_pool, used);
        max_lanes = std::max(max_lanes, b.n_lanes);
        s.batches.push_back(b);
    }
    s.out_cap = out_cap;
    s.d_seeds.upload(lane_seeds);
    s.d_lane_reads.upload(lane_reads);
    s.d_chrom_reads.upload(quotas);
    s.d_pool_off.upload(pool_off);
    for (uint32_t e = 0; e < s.n_ends; e++) {
        s.d_pool[0][e].alloc(max_pool + 64);
        if (s.batches.size() > 1) s.d_pool[1][e].alloc(max_pool + 64);
        s.d_out[e].alloc(out_cap + 64);
        s.d_lane_bytes[e].alloc(s.n_shard * 8);
        s.d_lane_off[e].alloc(s.n_shard * 8);
        s.d_base[e].alloc((s.batches.size() + 1) * 8);
    }
    s.d_lane_made.alloc(s.n_shard * 8);
    s.d_block_sums.alloc((max_lanes / SCAN_BLOCK + 2) * 8);
    s.d_err.alloc(4);
    for (hipEvent_t e : s.events) (void)hipEventDestroy(e);
    for (hipEvent_t e : s.gen_done) (void)hipEventDestroy(e);
    for (hipEvent_t e : s.cp_done) (void)hipEventDestroy(e);
    s.events.assign(2 + 2 * s.batches.size() + 2, nullptr);
    for (hipEvent_t& e : s.events) JK_HIP(hipEventCreate(&e));
    s.gen_done.assign(s.batches.size(), nullptr);
    s.cp_done.assign(s.batches.size(), nullptr);
    for (hipEvent_t& e : s.gen_done) JK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (hipEvent_t& e : s.cp_done) JK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return max_lanes;
}

static void plan_pools_and_alloc(jk_session& s, const jk_illumina_args& a, const std::vector<uint64_t>& lane_reads,
                                 uint64_t rec_max, const std::vector<uint32_t>& lane_seeds,
                                 const std::vector<uint32_t>& quotas) {
    // A batch is one generator launch.  Default: 2^18 lanes = one 1024-thread workgroup on each of the
    // 256 CUs, so every launch is a single full wave of workgroups and the pool compaction of batch b
    // (HBM-bound, second stream) runs under the generator of batch b+1 (ALU-bound).
    std::vector<uint64_t> lane_cap(s.n_shard);
    for (uint64_t l = 0; l < s.n_shard; l++) lane_cap[l] = (lane_reads[l] / s.n_ends) * rec_max;
    const uint32_t max_lanes = plan_pools_common(s, a.max_batch_bytes, 1ULL << 18, lane_cap, lane_reads, lane_seeds, quotas);
    s.d_info.upload(s.tables.info);
    s.d_thresh.upload(s.tables.thresh);
    s.d_quals.upload(s.tables.quals);
    s.d_mm.upload(s.tables.mm_thresh);
    s.evw_set = (size_t)s.n_ends * 4 * s.ev_words * std::max<uint32_t>(max_lanes, 1);      // u64 words per generator in flight
    s.d_evw.alloc(2 * s.evw_set * 8);

    s.lds_bytes = (s.tables.thresh.size() + (s.tables.thresh.size() & 1)) * 8 + 256 * 8 + s.tables.info.size() * 4 + align_up(s.tables.quals.size() * 2, 16);
    // haplotype runs add the per-lane segment table (4 segments x 12 bytes x 1024 lanes) after the tables
    const size_t seg_bytes = s.hap ? (size_t)4 * 12 * JK_ILL_BLOCK : 0;
    s.lds_tables = s.lds_bytes + seg_bytes <= 158 * 1024;
    s.lds_seg_off = s.lds_tables ? (uint32_t)align_up(s.lds_bytes, 16) : 0;
    s.lds_launch = (s.lds_tables ? align_up(s.lds_bytes, 16) : 0) + seg_bytes;

    IlluminaKernelParams& P = s.kp;
    P.g.seq = s.d_seq.as<uint8_t>();
    P.g.chrom_off = s.d_chrom_off.as<uint64_t>();
    P.g.chrom_len = s.d_chrom_len.as<uint64_t>();
    P.g.hdr_blob = s.d_hdr_blob.as<uint8_t>();
    P.g.hdr_off = s.d_hdr_off.as<uint32_t>();
    P.g.n_chroms = s.n_chroms;
    P.evw = s.d_evw.as<uint64_t>();
    P.err = s.d_err.as<uint32_t>();
    P.info = s.d_info.as<uint32_t>(); P.thresh = s.d_thresh.as<uint64_t>();
    P.quals = s.d_quals.as<uint16_t>(); P.mm_thresh = s.d_mm.as<uint64_t>();
    P.n_info = (uint32_t)s.tables.info.size(); P.n_entries = (uint32_t)s.tables.thresh.size();

    P.lds_seg_off = s.lds_seg_off;
    {
        const int lb = (int)s.lds_launch;
        if (s.lds_tables) {
            JK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&illumina_kernel<true, 1, JK_ILL_BLOCK, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lb));
            JK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&illumina_kernel<true, 2, JK_ILL_BLOCK, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lb));
            JK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&illumina_kernel<true, 1, JK_ILL_BLOCK, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lb));
            JK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&illumina_kernel<true, 2, JK_ILL_BLOCK, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lb));
        } else if (lb) {
            JK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&illumina_kernel<false, 1, JK_ILL_BLOCK, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lb));
            JK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&illumina_kernel<false, 2, JK_ILL_BLOCK, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lb));
        }
    }
}

static uint64_t record_max(size_t max_hdr, uint64_t max_chrom, bool paired, uint32_t L) {
    return max_hdr + n_digits(max_chrom) + 2 + (paired ? 2 : 0) + 1 + (uint64_t)L + 3 + L + 1;
}

// ---- illumina_ref_cpp (src/hts_illumina.cpp:589-649): everything the reference does on the calling
// thread before the parallel region, plus device set-up.
static void open_illumina_ref(jk_session& s, const jk_ref_genome& g, const jk_illumina_args& a, SeedReader& seeds) {
    setup_model(s, a);
    const uint32_t L = s.tables.read_length;
    const std::string barcode = (a.barcodes && a.n_barcodes > 0 && a.barcodes[0]) ? a.barcodes[0] : "";
    check_barcode(barcode, L);
    upload_genome(s, g, nullptr, 0);
    uint64_t min_chrom = ~0ULL, max_chrom = 0;
    size_t max_hdr = 0;
    const std::string gname = g.name ? g.name : "REF";
    for (uint64_t i = 0; i < g.n_chroms; i++) {
        min_chrom = std::min<uint64_t>(min_chrom, g.chrom_lens[i]);
        max_chrom = std::max<uint64_t>(max_chrom, g.chrom_lens[i]);
        max_hdr = std::max(max_hdr, 3 + gname.size() + std::strlen(g.chrom_names ? g.chrom_names[i] : ""));
    }
    {
        std::vector<uint8_t> blob;
        std::vector<uint32_t> hoff(g.n_chroms + 1);
        for (uint64_t i = 0; i < g.n_chroms; i++) {
            hoff[i] = (uint32_t)blob.size();
            std::string h = "@" + gname + "-" + (g.chrom_names ? g.chrom_names[i] : "") + "-";
            blob.insert(blob.end(), h.begin(), h.end());
        }
        hoff[g.n_chroms] = (uint32_t)blob.size();
        s.d_hdr_blob.upload(blob);
        s.d_hdr_off.upload(hoff);
    }
    const uint64_t frag_lb = std::min<uint64_t>(a.frag_len_min <= a.frag_len_max ? a.frag_len_min : a.frag_len_max, min_chrom);
    if (frag_lb < std::max<uint64_t>(barcode.size(), 1))
        throw Error(JK_ERR_UNSUPPORTED, "fragments shorter than the barcode (or empty) are not implemented on the GPU path");

    // ---- lanes, quotas, seeds: same order of seed consumption as src/hts.h:334-353
    std::vector<uint64_t> per_lane = plan_lanes(s, a.n_threads, a.lane_begin, a.lane_end, a.n_reads);
    const uint64_t T = s.n_lanes_total;
    std::vector<uint32_t> lane_seeds = take_lane_seeds(s, seeds);
    std::vector<uint64_t> lane_reads(s.n_shard);
    std::vector<uint32_t> chrom_reads((size_t)s.n_chroms * s.n_shard, 0);
    const std::vector<std::vector<double>> chrom_probs(1, std::vector<double>(g.chrom_lens, g.chrom_lens + g.n_chroms));
    DeferredSplits splits(&chrom_probs, chrom_reads.data(), s.n_shard, s.paired ? 2u : 1u);
    for (uint64_t t = 0; t < T; t++) {
        // IlluminaOneGenome::add_n_reads (src/hts_illumina.h:410-418)
        uint64_t n = per_lane[t];
        if (s.paired) n /= 2;
        const bool mine = t >= s.lane_begin && t < s.lane_end;
        if (!mine) {                       // only keep the seed stream in step
            if (n > 0) { uint32_t w[8]; seeds.take8(w); }
            continue;
        }
        const uint64_t l = t - s.lane_begin;
        lane_reads[l] = per_lane[t];
        splits.add(n, seeds, 0, 0, l);
    }
    splits.flush();
    s.seed_words_used = seeds.pos;

    IlluminaKernelParams& P = s.kp;
    P.bc_len = (uint32_t)barcode.size();
    std::memset(P.barcode, 0, sizeof(P.barcode));
    for (size_t k = 0; k < barcode.size(); k++) P.barcode[k] = encode_base(barcode[k]);
    plan_pools_and_alloc(s, a, lane_reads, record_max(max_hdr, max_chrom, s.paired, L), lane_seeds, chrom_reads);
}

// Mutation tables of a haplotype set -> device form (see HapDev); also uploads the genome + nucleotide blob.
static void upload_hap_tables(jk_session& s, const jk_hap_set& hs, uint64_t& min_chrom, uint64_t& max_chrom,
                              std::vector<uint64_t>& cell_size) {
    const uint64_t nh = hs.n_haps, nc = hs.ref.n_chroms;
    // ---- mutation tables -> device form (see HapDev)
    const uint64_t n_cells = nh * nc;
    std::vector<uint64_t> cell_off(n_cells + 1, 0);
    for (uint64_t k = 0; k < n_cells; k++) {
        if (hs.n_mut[k] > 0x7fffffffULL) throw Error(JK_ERR_UNSUPPORTED, "more than 2^31 mutations on one haplotype chromosome");
        cell_off[k + 1] = cell_off[k] + hs.n_mut[k];
    }
    const uint64_t n_mut = cell_off[n_cells];
    const uint64_t blob_len = n_mut ? hs.nuc_off[n_mut] : 0;
    upload_genome(s, hs.ref, hs.nuc_blob, blob_len);          // sets s.nuc_base = offset of the blob in seq
    std::vector<int64_t> ref_shift(n_mut);
    std::vector<uint32_t> nuc_len(n_mut);
    std::vector<uint64_t> nuc_dev_off(n_mut), new_pos(hs.new_pos, hs.new_pos + n_mut);
    cell_size.assign(hs.chrom_size, hs.chrom_size + n_cells);
    for (uint64_t k = 0; k < n_cells; k++) {
        const uint64_t ref_len = hs.ref.chrom_lens[k % nc];
        min_chrom = std::min(min_chrom, cell_size[k]);
        max_chrom = std::max(max_chrom, cell_size[k]);
        for (uint64_t m = cell_off[k]; m < cell_off[k + 1]; m++) {
            // size_modifier (src/hap_classes.h:314-333)
            int64_t smod = (m + 1 < cell_off[k + 1]) ? (int64_t)(hs.new_pos[m + 1] - hs.old_pos[m + 1])
                                                     : (int64_t)(cell_size[k] - ref_len);
            smod += (int64_t)(hs.old_pos[m] - hs.new_pos[m]);
            const uint64_t have = hs.nuc_off[m + 1] - hs.nuc_off[m];
            // equal new_pos happens: a deletion covers no haplotype position, so an edit right after it shares its new_pos
            if (m > cell_off[k] && hs.new_pos[m] < hs.new_pos[m - 1]) throw Error(JK_ERR_ARG, "mutation new_pos must not decrease within a chromosome");
            if (smod >= 0 && have < (uint64_t)smod + 1) throw Error(JK_ERR_ARG, "mutation has fewer nucleotides than its size modifier needs");
            if (smod + 1 > 0x7fffffffLL) throw Error(JK_ERR_UNSUPPORTED, "insertion longer than 2^31 bases");
            nuc_len[m] = smod >= 0 ? (uint32_t)(smod + 1) : 0u;
            nuc_dev_off[m] = s.nuc_base + hs.nuc_off[m];
            ref_shift[m] = (int64_t)hs.old_pos[m] - smod - (int64_t)hs.new_pos[m];
            // the reference run after this mutation must stay inside the chromosome
            const uint64_t run_end = (m + 1 < cell_off[k + 1]) ? hs.new_pos[m + 1] : cell_size[k];
            const int64_t last_ref = (int64_t)run_end - 1 + ref_shift[m];
            if (run_end > hs.new_pos[m] + nuc_len[m] && (last_ref < 0 || (uint64_t)last_ref >= ref_len))
                throw Error(JK_ERR_ARG, "mutation table points outside the reference chromosome");
        }
    }
    s.d_cell_off.upload(cell_off);
    s.d_new_pos.upload(new_pos);
    s.d_ref_shift.upload(ref_shift);
    s.d_nuc_len.upload(nuc_len);
    s.d_nuc_off.upload(nuc_dev_off);
    s.d_cell_size.upload(cell_size);
}

static void set_hap_params(const jk_session& s, HapDev& h, uint32_t n_haps) {
    h.cell_mut_off = s.d_cell_off.as<uint64_t>();
    h.new_pos = s.d_new_pos.as<uint64_t>();
    h.ref_shift = s.d_ref_shift.as<int64_t>();
    h.nuc_len = s.d_nuc_len.as<uint32_t>();
    h.nuc_off = s.d_nuc_off.as<uint64_t>();
    h.cell_size = s.d_cell_size.as<uint64_t>();
    h.bc_blob = s.d_bc_blob.as<uint8_t>();
    h.bc_len = s.d_bc_len.as<uint32_t>();
    h.n_haps = n_haps;
}

// ---- illumina_hap_cpp (src/hts_illumina.cpp:662-739), one set of output files (sep_files handled by
// the caller: it opens one session per haplotype with one-hot probabilities, src/hts.h:512-552).
static void open_illumina_hap(jk_session& s, const jk_hap_set& hs, const jk_illumina_args& a,
                              const std::vector<double>& hap_probs, uint64_t n_reads, SeedReader& seeds) {
    setup_model(s, a);
    s.hap = true;
    const uint32_t L = s.tables.read_length;
    const uint64_t nh = hs.n_haps, nc = hs.ref.n_chroms;
    if (nh == 0 || nc == 0) throw Error(JK_ERR_ARG, "haplotype set is empty");
    if (nh * nc > 0x7fffffffULL) throw Error(JK_ERR_UNSUPPORTED, "too many (haplotype, chromosome) cells");
    if (hap_probs.size() != nh) throw Error(JK_ERR_ARG, "haplotype_probs must have one entry per haplotype");
    // barcodes: padded with "" to one per haplotype (src/hts_illumina.h:550)
    std::vector<std::string> bcs(nh);
    for (uint64_t h = 0; h < nh && h < a.n_barcodes; h++) bcs[h] = (a.barcodes && a.barcodes[h]) ? a.barcodes[h] : "";
    size_t max_bc = 0;
    for (const std::string& b : bcs) { check_barcode(b, L); max_bc = std::max(max_bc, b.size()); }

    uint64_t min_chrom = ~0ULL, max_chrom = 0;
    std::vector<uint64_t> cell_size;
    upload_hap_tables(s, hs, min_chrom, max_chrom, cell_size);
    const uint64_t n_cells = nh * nc;
    {
        std::vector<uint8_t> blob(nh * JK_MAX_BARCODE, 0);
        std::vector<uint32_t> blen(nh);
        for (uint64_t h = 0; h < nh; h++) {
            blen[h] = (uint32_t)bcs[h].size();
            for (size_t k = 0; k < bcs[h].size(); k++) blob[h * JK_MAX_BARCODE + k] = encode_base(bcs[h][k]);
        }
        s.d_bc_blob.upload(blob);
        s.d_bc_len.upload(blen);
    }
    size_t max_hdr = 0;
    {   // "@<haplotype>-<chromosome>-" per cell
        std::vector<uint8_t> blob;
        std::vector<uint32_t> hoff(n_cells + 1);
        for (uint64_t k = 0; k < n_cells; k++) {
            hoff[k] = (uint32_t)blob.size();
            std::string h = std::string("@") + (hs.hap_names ? hs.hap_names[k / nc] : "") + "-" +
                            (hs.ref.chrom_names ? hs.ref.chrom_names[k % nc] : "") + "-";
            max_hdr = std::max(max_hdr, h.size());
            blob.insert(blob.end(), h.begin(), h.end());
        }
        hoff[n_cells] = (uint32_t)blob.size();
        s.d_hdr_blob.upload(blob);
        s.d_hdr_off.upload(hoff);
    }
    const uint64_t frag_lb = std::min<uint64_t>(a.frag_len_min <= a.frag_len_max ? a.frag_len_min : a.frag_len_max, min_chrom);
    if (frag_lb < std::max<uint64_t>(max_bc, 1))
        throw Error(JK_ERR_UNSUPPORTED, "fragments shorter than the barcode (or empty) are not implemented on the GPU path");

    // ---- lanes, quotas, seeds.  IlluminaHaplotypes::add_n_reads (src/hts_illumina.h:620-644) per lane:
    // reads_per_group over haplotypes, then per haplotype reads_per_group over its chromosomes, then
    // each read maker's own add_n_reads (halves the pair count again when paired; its result is never
    // read by the haplotype path, but it consumes 8 seed words when it has reads).
    std::vector<uint64_t> per_lane = plan_lanes(s, a.n_threads, a.lane_begin, a.lane_end, n_reads);
    const uint64_t T = s.n_lanes_total;
    std::vector<uint32_t> lane_seeds = take_lane_seeds(s, seeds);
    std::vector<uint64_t> lane_reads(s.n_shard);
    std::vector<uint32_t> vc((size_t)n_cells * s.n_shard, 0);
    std::vector<std::vector<double>> chrom_probs(nh, std::vector<double>(nc));
    for (uint64_t h = 0; h < nh; h++) for (uint64_t c = 0; c < nc; c++) chrom_probs[h][c] = (double)cell_size[h * nc + c];
    DeferredSplits splits(&chrom_probs, vc.data(), s.n_shard, s.paired ? 2u : 1u);
    for (uint64_t t = 0; t < T; t++) {
        uint64_t n = per_lane[t];
        if (s.paired) n /= 2;
        const bool mine = t >= s.lane_begin && t < s.lane_end;
        std::vector<uint64_t> hap_reads = reads_per_group(n, hap_probs, seeds);
        for (uint64_t h = 0; h < nh; h++) {
            if (mine) splits.add(hap_reads[h], seeds, (uint32_t)h, h * nc, t - s.lane_begin);
            else if (hap_reads[h] > 0) { uint32_t w[8]; seeds.take8(w); }
        }
        for (uint64_t h = 0; h < nh; h++) {
            uint64_t m = hap_reads[h];
            if (s.paired) m /= 2;
            if (m > 0) { uint32_t w[8]; seeds.take8(w); }
        }
        if (mine) lane_reads[t - s.lane_begin] = per_lane[t];
    }
    splits.flush();
    s.seed_words_used = seeds.pos;

    IlluminaKernelParams& P = s.kp;
    P.bc_len = 0;
    std::memset(P.barcode, 0, sizeof(P.barcode));
    set_hap_params(s, P.h, (uint32_t)nh);
    plan_pools_and_alloc(s, a, lane_reads, record_max(max_hdr, max_chrom, s.paired, L), lane_seeds, vc);
}

// ---- pacbio_ref_cpp / pacbio_hap_cpp (src/hts_pacbio.cpp:579-715): host set-up -----------------------
// Everything that depends only on the run's parameters or on an integer is tabulated here with the host's
// libm (exactly what the reference calls) and the nmath restatements of jk_nmath.h.
struct PacbioHostModel {
    std::vector<uint64_t> len_thresh; std::vector<uint32_t> len_alias; std::vector<uint64_t> lens;
    std::vector<double> thr_tab; std::vector<PassEntry> pass_tab;
    double min_exp = 0;
    uint64_t len_hi = 0;      // pool sizing: a read length few reads exceed
    uint64_t len_cap = 0;     // hard cap (event scratch)
};

static PacbioHostModel setup_pacbio_model(jk_session& s, const jk_pacbio_args& a, uint64_t max_chrom) {
    set_compression(s, a.compress, a.comp_method);
    if (!a.chi2_params_n || !a.chi2_params_s || !a.sqrt_params || !a.norm_params) throw Error(JK_ERR_ARG, "PacBio parameter vectors must not be NULL");
    s.pacbio = true; s.paired = false; s.n_ends = 1;
    s.out_prefix = a.out_prefix ? a.out_prefix : "";
    s.abort_flag = a.abort_flag;
    s.device = a.device;
    JK_HIP(hipSetDevice(s.device));
    JK_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    JK_HIP(hipStreamCreateWithFlags(&s.cp_stream, hipStreamNonBlocking));

    PacbioHostModel M;
    PacbioKernelParams& P = s.kpb;
    // read lengths (PacBioReadLenSampler, src/hts_pacbio.h:45-109)
    if (a.n_read_lens == 0) {
        P.use_lognormal = 1;
        P.ln_mu = std::log(a.scale); P.ln_sigma = a.sigma; P.ln_loc = a.loc;
        P.min_read_len = std::ceil(a.min_read_len);
        if (P.min_read_len < 1) P.min_read_len = 1;
        const double hi = std::exp(P.ln_mu + 4.0 * a.sigma) + a.loc, cap = std::exp(P.ln_mu + 9.0 * a.sigma) + a.loc;
        M.len_hi = (uint64_t)std::max(hi, P.min_read_len + 1.0);
        M.len_cap = (uint64_t)std::max(cap, P.min_read_len + 1.0);
    } else {
        if (!a.read_probs || !a.read_lens) throw Error(JK_ERR_ARG, "Probability and read lengths vector should be the same length.");
        P.use_lognormal = 0;
        AliasTable at = alias_build(std::vector<double>(a.read_probs, a.read_probs + a.n_read_lens));
        for (uint64_t i = 0; i < a.n_read_lens; i++) {
            Threshold th = threshold_lt(at.prob[i]);
            M.len_thresh.push_back(th.all ? ~uint64_t(0) : th.th);
            M.len_alias.push_back(th.all ? (uint32_t)i : (uint32_t)at.alias[i]);
            M.lens.push_back(a.read_lens[i]);
            M.len_hi = std::max(M.len_hi, a.read_lens[i]);
        }
        M.len_cap = M.len_hi;
        if (a.n_read_lens >= (1ULL << 31)) throw Error(JK_ERR_UNSUPPORTED, "too many custom read lengths");
        P.n_lens = (uint32_t)a.n_read_lens;
    }
    M.len_hi = std::min(M.len_hi, max_chrom);
    M.len_cap = std::min(M.len_cap, max_chrom);
    // passes (PacBioPassSampler): qchisq(0.9925, n(L)) for every read length that changes n
    for (int i = 0; i < 3; i++) P.cn[i] = a.chi2_params_n[i];
    for (int i = 0; i < 5; i++) P.cs[i] = a.chi2_params_s[i];
    P.max_passes_d = static_cast<double>(a.max_passes);
    if (a.max_passes < 1 || a.max_passes > 100000) throw Error(JK_ERR_ARG, "max_passes out of range");
    {
        const double n2 = P.cn[2];
        uint64_t cap = n2 >= 1 ? (uint64_t)std::min(std::floor(n2), (double)M.len_cap) : 0;
        if (cap > (64ULL << 20)) throw Error(JK_ERR_UNSUPPORTED, "chi2_params_n[3] too large for the GPU path's threshold table");
        M.thr_tab.resize(cap + 2);
        for (uint64_t L = 0; L <= cap + 1; L++) {
            const double Ld = (L <= cap) ? (double)L : std::max((double)(cap + 1), n2);   // last entry: the capped value
            double n = P.cn[0] * std::min(Ld, n2) + P.cn[1];
            if (n < 0.001) n = 0.001;
            M.thr_tab[L] = qchisq_upper_tail_point(0.9925, n);
        }
        P.thr_cap = (uint32_t)(cap + 1);
    }
    // qualities/errors (PacBioQualityError)
    P.np0 = a.norm_params[0]; P.np1 = a.norm_params[1]; P.sp1 = a.sqrt_params[1];
    P.prob_ins = a.prob_ins; P.prob_del = a.prob_del; P.prob_subst = a.prob_subst;
    {   // calc_min_exp (src/hts_pacbio.cpp:50-91)
        auto total_at = [&](double e) { return std::pow(a.prob_ins, e) + std::pow(a.prob_del, e) + std::pow(a.prob_subst, e); };
        double min_exp_ = 1, total = total_at(min_exp_), left, right;
        if (total < a.prob_thresh) {
            while (total < a.prob_thresh) { min_exp_ /= 2; total = total_at(min_exp_); }
            left = min_exp_; right = min_exp_ * 2;
        } else {
            while (total > a.prob_thresh) { min_exp_ *= 2; total = total_at(min_exp_); }
            left = min_exp_ / 2; right = min_exp_;
        }
        for (int i = 0; i < 15; i++) {
            const double m = (left + right) / 2;
            total = total_at(m);
            if (total == a.prob_thresh) { min_exp_ = m; break; }
            else if (total > a.prob_thresh) { left = m; min_exp_ = (m + right) / 2; }
            else { right = m; min_exp_ = (left + m) / 2; }
        }
        M.min_exp = min_exp_;
    }
    M.pass_tab.resize(a.max_passes + 2);
    for (uint64_t k = 0; k < M.pass_tab.size(); k++) {
        const double passes = (double)k;
        PassEntry& e = M.pass_tab[k];
        e.sig = 1 / (1 + std::pow(2, (-2.5 / 3 * passes + 6.5 / 3)));                 // sigmoid (hts_pacbio.h:333-335)
        e.sqrtv = std::sqrt(passes + a.sqrt_params[0]);
        const double lower_thresh = (M.min_exp - (e.sqrtv - a.sqrt_params[1])) / e.sig;   // update_probs (hts_pacbio.cpp:101-104)
        e.a_bar = (lower_thresh - a.norm_params[0]) / a.norm_params[1];
        if (lower_thresh < (a.norm_params[0] + 5 * a.norm_params[1])) {
            e.method = 0;
            e.p = pnorm_std(e.a_bar);
            jk_x87_one_minus(e.p, &e.c_m, &e.c_e);
        } else { e.method = 1; e.p = 0; e.c_m = 0; e.c_e = 0; }
    }
    {   // dup < prob_dup
        Threshold t = threshold_lt(a.prob_dup);
        P.th_dup = t.th; P.dup_all = t.all;
    }
    P.pool_size = a.read_pool_size;
    return M;
}

static void finish_pacbio(jk_session& s, uint64_t max_batch_bytes, const PacbioHostModel& M, size_t max_hdr, uint64_t max_chrom,
                          const std::vector<uint64_t>& lane_reads, const std::vector<uint32_t>& lane_seeds,
                          const std::vector<uint32_t>& quotas) {
    // pools: sized for reads of length len_hi; the kernel checks before every record and the session retries
    // with a larger scale if a lane ran out (s.pool_scale)
    const uint64_t rec = max_hdr + n_digits(max_chrom) + 3 + 2 * M.len_hi + 8;
    std::vector<uint64_t> lane_cap(s.n_shard);
    for (uint64_t l = 0; l < s.n_shard; l++)       // per-lane regions are contiguous and hold whole 128-byte lines
        lane_cap[l] = align_up((uint64_t)((double)(lane_reads[l] * rec) * s.pool_scale) + 2 * M.len_cap + 64, 128) + 128;
    const uint32_t max_lanes = plan_pools_common(s, max_batch_bytes ? max_batch_bytes : (48ULL << 30), 1ULL << 18,
                                                 lane_cap, lane_reads, lane_seeds, quotas);
    s.ev_words = (uint32_t)((2 * M.len_cap + 64 + 31) / 32);
    s.d_ev2.alloc((size_t)s.ev_words * std::max<uint32_t>(max_lanes, 1) * 8);
    s.d_len_thresh.upload(M.len_thresh); s.d_len_alias.upload(M.len_alias); s.d_lens.upload(M.lens);
    s.d_thr_tab.upload(M.thr_tab); s.d_pass_tab.upload(M.pass_tab);
    PacbioKernelParams& P = s.kpb;
    P.g.seq = s.d_seq.as<uint8_t>();
    P.g.chrom_off = s.d_chrom_off.as<uint64_t>();
    P.g.chrom_len = s.d_chrom_len.as<uint64_t>();
    P.g.hdr_blob = s.d_hdr_blob.as<uint8_t>();
    P.g.hdr_off = s.d_hdr_off.as<uint32_t>();
    P.g.n_chroms = s.n_chroms;
    P.ev = s.d_ev2.as<uint64_t>(); P.ev_words = s.ev_words;
    P.err = s.d_err.as<uint32_t>();
    P.len_thresh = s.d_len_thresh.as<uint64_t>(); P.len_alias = s.d_len_alias.as<uint32_t>(); P.lens = s.d_lens.as<uint64_t>();
    P.thr_tab = s.d_thr_tab.as<double>(); P.pass_tab = s.d_pass_tab.as<PassEntry>();
}

static void upload_headers(jk_session& s, const std::vector<std::string>& hdrs, size_t& max_hdr) {
    std::vector<uint8_t> blob;
    std::vector<uint32_t> hoff(hdrs.size() + 1);
    for (size_t k = 0; k < hdrs.size(); k++) {
        hoff[k] = (uint32_t)blob.size();
        max_hdr = std::max(max_hdr, hdrs[k].size());
        blob.insert(blob.end(), hdrs[k].begin(), hdrs[k].end());
    }
    hoff[hdrs.size()] = (uint32_t)blob.size();
    s.d_hdr_blob.upload(blob);
    s.d_hdr_off.upload(hoff);
}

static void open_pacbio_ref(jk_session& s, const jk_ref_genome& g, const jk_pacbio_args& a, SeedReader& seeds) {
    uint64_t max_chrom = 0;
    for (uint64_t i = 0; i < g.n_chroms; i++) max_chrom = std::max<uint64_t>(max_chrom, g.chrom_lens[i]);
    PacbioHostModel M = setup_pacbio_model(s, a, max_chrom);
    upload_genome(s, g, nullptr, 0);
    const std::string gname = g.name ? g.name : "REF";
    std::vector<std::string> hdrs;
    for (uint64_t i = 0; i < g.n_chroms; i++) hdrs.push_back("@" + gname + "-" + (g.chrom_names ? g.chrom_names[i] : "") + "-");
    size_t max_hdr = 0;
    upload_headers(s, hdrs, max_hdr);
    // lanes, quotas, seeds (src/hts.h:334-353 with n_read_ends = 1; PacBioOneGenome::add_n_reads, hts_pacbio.h:499-503)
    std::vector<uint64_t> per_lane = plan_lanes(s, a.n_threads, a.lane_begin, a.lane_end, a.n_reads);
    const uint64_t T = s.n_lanes_total;
    std::vector<uint32_t> lane_seeds = take_lane_seeds(s, seeds);
    std::vector<uint64_t> lane_reads(s.n_shard);
    std::vector<uint32_t> chrom_reads((size_t)s.n_chroms * s.n_shard, 0);
    const std::vector<std::vector<double>> chrom_probs(1, std::vector<double>(g.chrom_lens, g.chrom_lens + g.n_chroms));
    DeferredSplits splits(&chrom_probs, chrom_reads.data(), s.n_shard, 1u);
    for (uint64_t t = 0; t < T; t++) {
        const uint64_t n = per_lane[t];
        const bool mine = t >= s.lane_begin && t < s.lane_end;
        if (!mine) { if (n > 0) { uint32_t w[8]; seeds.take8(w); } continue; }
        const uint64_t l = t - s.lane_begin;
        lane_reads[l] = n;
        splits.add(n, seeds, 0, 0, l);
    }
    splits.flush();
    s.seed_words_used = seeds.pos;
    const uint64_t mbb = a.max_batch_bytes;
    jk_session* sp = &s;
    s.replan = [=]() { finish_pacbio(*sp, mbb, M, max_hdr, max_chrom, lane_reads, lane_seeds, chrom_reads); };
    s.replan();
}

static void open_pacbio_hap(jk_session& s, const jk_hap_set& hs, const jk_pacbio_args& a,
                            const std::vector<double>& hap_probs, uint64_t n_reads, SeedReader& seeds) {
    const uint64_t nh = hs.n_haps, nc = hs.ref.n_chroms;
    if (nh == 0 || nc == 0) throw Error(JK_ERR_ARG, "haplotype set is empty");
    if (hap_probs.size() != nh) throw Error(JK_ERR_ARG, "haplotype_probs must have one entry per haplotype");
    uint64_t max_c = 0;
    for (uint64_t k = 0; k < nh * nc; k++) max_c = std::max<uint64_t>(max_c, hs.chrom_size[k]);
    PacbioHostModel M = setup_pacbio_model(s, a, max_c);
    s.hap = true;
    uint64_t min_chrom = ~0ULL, max_chrom = 0;
    std::vector<uint64_t> cell_size;
    upload_hap_tables(s, hs, min_chrom, max_chrom, cell_size);
    {   // no barcodes on this path
        std::vector<uint8_t> blob(nh * JK_MAX_BARCODE, 0); std::vector<uint32_t> blen(nh, 0);
        s.d_bc_blob.upload(blob); s.d_bc_len.upload(blen);
    }
    const uint64_t n_cells = nh * nc;
    std::vector<std::string> hdrs;
    for (uint64_t k = 0; k < n_cells; k++)
        hdrs.push_back(std::string("@") + (hs.hap_names ? hs.hap_names[k / nc] : "") + "-" + (hs.ref.chrom_names ? hs.ref.chrom_names[k % nc] : "") + "-");
    size_t max_hdr = 0;
    upload_headers(s, hdrs, max_hdr);
    // PacBioHaplotypes::add_n_reads (src/hts_pacbio.h:683-700)
    std::vector<uint64_t> per_lane = plan_lanes(s, a.n_threads, a.lane_begin, a.lane_end, n_reads);
    const uint64_t T = s.n_lanes_total;
    std::vector<uint32_t> lane_seeds = take_lane_seeds(s, seeds);
    std::vector<uint64_t> lane_reads(s.n_shard);
    std::vector<uint32_t> vc((size_t)n_cells * s.n_shard, 0);
    std::vector<std::vector<double>> chrom_probs(nh, std::vector<double>(nc));
    for (uint64_t h = 0; h < nh; h++) for (uint64_t c = 0; c < nc; c++) chrom_probs[h][c] = (double)cell_size[h * nc + c];
    DeferredSplits splits(&chrom_probs, vc.data(), s.n_shard, 1u);
    for (uint64_t t = 0; t < T; t++) {
        const uint64_t n = per_lane[t];
        const bool mine = t >= s.lane_begin && t < s.lane_end;
        std::vector<uint64_t> hap_reads = reads_per_group(n, hap_probs, seeds);
        for (uint64_t h = 0; h < nh; h++) {
            if (mine) splits.add(hap_reads[h], seeds, (uint32_t)h, h * nc, t - s.lane_begin);
            else if (hap_reads[h] > 0) { uint32_t w[8]; seeds.take8(w); }
        }
        for (uint64_t h = 0; h < nh; h++) if (hap_reads[h] > 0) { uint32_t w[8]; seeds.take8(w); }   // read_makers[h].add_n_reads
        if (mine) lane_reads[t - s.lane_begin] = n;
    }
    splits.flush();
    s.seed_words_used = seeds.pos;
    set_hap_params(s, s.kpb.h, (uint32_t)nh);
    const uint64_t mbb = a.max_batch_bytes;
    jk_session* sp = &s;
    s.replan = [=]() { finish_pacbio(*sp, mbb, M, max_hdr, max_chrom, lane_reads, lane_seeds, vc); };
    s.replan();
}

static void launch_generate(jk_session& s) {
    JK_HIP(hipSetDevice(s.device));
    JK_HIP(hipMemsetAsync(s.d_err.p, 0, 4, s.stream));
    for (uint32_t e = 0; e < s.n_ends; e++) JK_HIP(hipMemsetAsync(s.d_base[e].p, 0, 8, s.stream));
    size_t ev = 0;
    JK_HIP(hipEventRecord(s.events[ev++], s.stream));
    JK_HIP(hipStreamWaitEvent(s.cp_stream, s.events[0], 0));
    for (size_t b = 0; b < s.batches.size(); b++) {
        if (s.abort_flag && *s.abort_flag) throw Error(JK_ERR_ABORTED, "aborted");
        const Batch& B = s.batches[b];
        const int pp = (int)(b & 1);       // ping-pong pool set
        if (s.pacbio) {
            PacbioKernelParams Q = s.kpb;
            Q.n_lanes = B.n_lanes;
            Q.seeds = s.d_seeds.as<uint32_t>() + B.lane0 * 8;
            Q.lane_reads = s.d_lane_reads.as<uint64_t>() + B.lane0;
            Q.chrom_reads = s.d_chrom_reads.as<uint32_t>() + B.lane0;
            Q.chrom_stride = (uint32_t)s.n_shard;
            Q.pool_off = s.d_pool_off.as<uint64_t>() + s.batch_pool_off_index[b];
            Q.pool = s.d_pool[pp][0].as<uint8_t>();
            Q.lane_bytes = s.d_lane_bytes[0].as<uint64_t>() + B.lane0;
            Q.lane_made = s.d_lane_made.as<uint64_t>() + B.lane0;
            if (b >= 2) JK_HIP(hipStreamWaitEvent(s.stream, s.cp_done[b - 2], 0));
            JK_HIP(hipEventRecord(s.events[ev++], s.stream));
            const uint32_t pgrid = (B.n_lanes + 255) / 256;
            if (s.hap) hipLaunchKernelGGL((pacbio_kernel<true>), dim3(pgrid), dim3(256), 0, s.stream, Q);
            else hipLaunchKernelGGL((pacbio_kernel<false>), dim3(pgrid), dim3(256), 0, s.stream, Q);
            JK_HIP(hipGetLastError());
            JK_HIP(hipEventRecord(s.events[ev++], s.stream));
            JK_HIP(hipEventRecord(s.gen_done[b], s.stream));
            JK_HIP(hipStreamWaitEvent(s.cp_stream, s.gen_done[b], 0));
            const uint32_t nbp = (B.n_lanes + SCAN_BLOCK - 1) / SCAN_BLOCK;
            uint64_t* lb = s.d_lane_bytes[0].as<uint64_t>() + B.lane0;
            uint64_t* lo = s.d_lane_off[0].as<uint64_t>() + B.lane0;
            uint64_t* bs = s.d_block_sums.as<uint64_t>();
            uint64_t* base = s.d_base[0].as<uint64_t>() + b;
            hipLaunchKernelGGL(scan_block_kernel, dim3(nbp), dim3(SCAN_BLOCK), 0, s.cp_stream, lb, lo, bs, B.n_lanes);
            hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_BLOCK), 0, s.cp_stream, bs, nbp, base);
            hipLaunchKernelGGL(scan_add_kernel, dim3(nbp), dim3(SCAN_BLOCK), 0, s.cp_stream, lo, bs, B.n_lanes);
            hipLaunchKernelGGL(compact_linear_kernel, dim3(B.n_lanes), dim3(256), 0, s.cp_stream,
                               s.d_pool[pp][0].as<uint8_t>(), Q.pool_off, lb, lo, s.d_out[0].as<uint8_t>(), base, B.n_lanes);
            JK_HIP(hipGetLastError());
            JK_HIP(hipEventRecord(s.cp_done[b], s.cp_stream));
            continue;
        }
        IlluminaKernelParams P = s.kp;
        P.n_lanes = B.n_lanes;
        P.seeds = s.d_seeds.as<uint32_t>() + B.lane0 * 8;
        P.lane_reads = s.d_lane_reads.as<uint64_t>() + B.lane0;
        // quotas are laid out [chromosome or cell][lane of the shard]: row stride n_shard
        P.chrom_reads = s.d_chrom_reads.as<uint32_t>() + B.lane0;
        P.pool_off = s.d_pool_off.as<uint64_t>() + s.batch_pool_off_index[b];
        for (uint32_t e = 0; e < 2; e++) {
            P.pool[e] = e < s.n_ends ? s.d_pool[pp][e].as<uint8_t>() : nullptr;
            P.lane_bytes[e] = e < s.n_ends ? s.d_lane_bytes[e].as<uint64_t>() + B.lane0 : nullptr;
        }
        P.lane_made = s.d_lane_made.as<uint64_t>() + B.lane0;
        P.evw = s.d_evw.as<uint64_t>() + (size_t)pp * s.evw_set;
        P.chrom_stride = (uint32_t)s.n_shard;
        // Two generators may be in flight (each has its own pool set and indel scratch): the next batch's
        // workgroups then take over CUs as the current batch's finish instead of waiting for its slowest one.
        hipStream_t gs = (s.two_gen_streams && (b & 1)) ? s.stream2 : s.stream;
        if (s.two_gen_streams && b == 1) JK_HIP(hipStreamWaitEvent(s.stream2, s.events[0], 0));
#ifndef JK_ILL_BLOCK
#define JK_ILL_BLOCK 1024
#endif
        const uint32_t block = JK_ILL_BLOCK;
        const uint32_t grid = (B.n_lanes + block - 1) / block;
        // the pool set is free again once the compaction of batch b-2 has read it
        if (b >= 2) JK_HIP(hipStreamWaitEvent(gs, s.cp_done[b - 2], 0));
        JK_HIP(hipEventRecord(s.events[ev++], gs));
#define JK_LAUNCH(LDS, NE, HAP, SH) hipLaunchKernelGGL((illumina_kernel<LDS, NE, JK_ILL_BLOCK, HAP>), dim3(grid), dim3(block), SH, gs, P)
        if (s.lds_tables) {
            if (s.hap) { if (s.n_ends == 2) JK_LAUNCH(true, 2, true, s.lds_launch); else JK_LAUNCH(true, 1, true, s.lds_launch); }
            else       { if (s.n_ends == 2) JK_LAUNCH(true, 2, false, s.lds_launch); else JK_LAUNCH(true, 1, false, s.lds_launch); }
        } else {
            if (s.hap) { if (s.n_ends == 2) JK_LAUNCH(false, 2, true, s.lds_launch); else JK_LAUNCH(false, 1, true, s.lds_launch); }
            else       { if (s.n_ends == 2) JK_LAUNCH(false, 2, false, 0); else JK_LAUNCH(false, 1, false, 0); }
        }
#undef JK_LAUNCH
        JK_HIP(hipGetLastError());
        JK_HIP(hipEventRecord(s.events[ev++], gs));
        JK_HIP(hipEventRecord(s.gen_done[b], gs));
        JK_HIP(hipStreamWaitEvent(s.cp_stream, s.gen_done[b], 0));
        const uint32_t nb = (B.n_lanes + SCAN_BLOCK - 1) / SCAN_BLOCK;
        for (uint32_t e = 0; e < s.n_ends; e++) {
            uint64_t* lb = s.d_lane_bytes[e].as<uint64_t>() + B.lane0;
            uint64_t* lo = s.d_lane_off[e].as<uint64_t>() + B.lane0;
            uint64_t* bs = s.d_block_sums.as<uint64_t>();
            uint64_t* base = s.d_base[e].as<uint64_t>() + b;
            hipLaunchKernelGGL(scan_block_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, s.cp_stream, lb, lo, bs, B.n_lanes);
            hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_BLOCK), 0, s.cp_stream, bs, nb, base);
            hipLaunchKernelGGL(scan_add_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, s.cp_stream, lo, bs, B.n_lanes);
            hipLaunchKernelGGL(compact_pools_kernel, dim3((B.n_lanes + 63) / 64), dim3(256), 0, s.cp_stream,
                               s.d_pool[pp][e].as<uint8_t>(), P.pool_off, lb, lo, s.d_out[e].as<uint8_t>(), base, B.n_lanes);
            JK_HIP(hipGetLastError());
        }
        JK_HIP(hipEventRecord(s.cp_done[b], s.cp_stream));
    }
    if (!s.batches.empty()) JK_HIP(hipStreamWaitEvent(s.stream, s.cp_done[s.batches.size() - 1], 0));
    JK_HIP(hipEventRecord(s.events[ev++], s.stream));
    JK_HIP(hipStreamSynchronize(s.stream));
    JK_HIP(hipStreamSynchronize(s.cp_stream));

    uint32_t err = 0;
    JK_HIP(hipMemcpy(&err, s.d_err.p, 4, hipMemcpyDeviceToHost));
    if (err & JK_KERR_PB_ALPHA) throw Error(JK_ERR_UNSUPPORTED, "chi-square shape n/2 < 1 (chi2_params_n) is not implemented on the GPU path");
    if (err & JK_KERR_PB_MATH) throw Error(JK_ERR_UNSUPPORTED, "a PacBio parameter led to an exp/pow argument outside the range implemented on the GPU");
    if (err & JK_KERR_PB_TOO_LONG) throw Error(JK_ERR_UNSUPPORTED, "a read was longer than the GPU path's cap (9 sigma of the log-normal, or it needed > 2x its length in reference positions)");
    if (err & JK_KERR_PB_SPACE) throw Error(JK_ERR_UNSUPPORTED, "a read ran past its chromosome window (reads as long as their chromosome are not implemented on the GPU path)");
    if ((err & JK_KERR_POOL_OVERFLOW) && s.pacbio) throw Error(JK_ERR_RETRY, "pool overflow");
    if (err & JK_KERR_POOL_OVERFLOW) throw Error(JK_ERR_DEVICE, "internal error: a lane overflowed its pool region");
    if (err & JK_KERR_TOO_MANY_DELETIONS) throw Error(JK_ERR_UNSUPPORTED, "a read needed more than 2x read_length reference positions (deletion probability too high for the GPU path)");
    for (uint32_t e = 0; e < s.n_ends; e++)
        JK_HIP(hipMemcpy(&s.bytes[e], s.d_base[e].as<uint64_t>() + s.batches.size(), 8, hipMemcpyDeviceToHost));
    {
        std::vector<uint64_t> made(s.n_shard);
        if (s.n_shard) JK_HIP(hipMemcpy(made.data(), s.d_lane_made.p, s.n_shard * 8, hipMemcpyDeviceToHost));
        s.reads_made = 0;
        for (uint64_t v : made) s.reads_made += v;
    }
    float t = 0;
    double gen = 0, rest = 0;
    for (size_t b = 0; b < s.batches.size(); b++) {
        JK_HIP(hipEventElapsedTime(&t, s.events[1 + 2 * b], s.events[2 + 2 * b]));
        gen += t;
    }
    JK_HIP(hipEventElapsedTime(&t, s.events[0], s.events[ev - 1]));
    rest = t - gen;
    s.ms[0] = gen; s.ms[1] = rest; s.ms[2] = t;
    s.generated = true;
}

// ---- output sinks (src/io.h:58-295): plain file, gzip (zlib gzFile) or BGZF ---------------------------
// BGZF = concatenated gzip members of <= 0xff00 input bytes with a 'BC' extra field and a fixed empty
// end-of-file member (the format htslib's bgzf_write produces; readable by gzip, zcat, bgzip, samtools).
static const size_t BGZF_IN = 0xff00;

static void bgzf_compress_block(const uint8_t* src, size_t n, int level, std::vector<uint8_t>& out) {
    const size_t start = out.size();
    out.resize(start + 18 + compressBound(n) + 8);
    uint8_t* h = out.data() + start;
    static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    std::memcpy(h, head, 16);
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw Error(JK_ERR_IO, "deflateInit2 failed");
    zs.next_in = const_cast<uint8_t*>(src); zs.avail_in = (uInt)n;
    zs.next_out = h + 18; zs.avail_out = (uInt)(out.size() - start - 18 - 8);
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { deflateEnd(&zs); throw Error(JK_ERR_IO, "deflate failed"); }
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    const size_t total = 18 + clen + 8;
    if (total > 65536) throw Error(JK_ERR_IO, "BGZF block did not compress below 64 KiB");
    h[16] = (uint8_t)((total - 1) & 0xff); h[17] = (uint8_t)((total - 1) >> 8);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), src, (uInt)n);
    uint8_t* t = h + 18 + clen;
    for (int i = 0; i < 4; i++) { t[i] = (uint8_t)(crc >> (8 * i)); t[4 + i] = (uint8_t)((uint32_t)n >> (8 * i)); }
    out.resize(start + total);
}

// ---- BGZF on the device (jk_bgzf_kernel.h) -----------------------------------------------------------
static const uint8_t kBgzfEof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};

static uint64_t bgzf_bound(uint64_t n) {
    const uint64_t nb = (n + BGZF_BLOCK_IN - 1) / BGZF_BLOCK_IN;
    return n + nb * 31 + sizeof(kBgzfEof);          // every block stored: 18 + 5 + 8 bytes around its input
}

struct BgzfDeviceTables { DevBuf crc, x512, x8; };
static BgzfTables bgzf_tables(int device) {
    static std::vector<std::unique_ptr<BgzfDeviceTables>> per_device(64);
    if (device < 0 || device >= 64) throw Error(JK_ERR_ARG, "bad device ordinal");
    if (!per_device[device]) {
        std::vector<uint32_t> crc(4 * 256), x512(1024), x8(64);
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1u) ? CRC_POLY : 0u);
            crc[i] = c;
        }
        for (int k = 1; k < 4; k++)                 // slicing tables: one more zero byte appended per level
            for (uint32_t i = 0; i < 256; i++) crc[k * 256 + i] = (crc[(k - 1) * 256 + i] >> 8) ^ crc[crc[(k - 1) * 256 + i] & 0xffu];
        uint32_t xb = 0x80000000u;                   // x^0
        for (int k = 0; k < 8; k++) xb = crc_mulmod(xb, 0x40000000u);     // x^8
        x8[0] = 0x80000000u;
        for (int r = 1; r < 64; r++) x8[r] = crc_mulmod(x8[r - 1], xb);
        const uint32_t step = crc_mulmod(x8[63], xb);                    // x^512
        x512[0] = 0x80000000u;
        for (int j = 1; j < 1024; j++) x512[j] = crc_mulmod(x512[j - 1], step);
        std::unique_ptr<BgzfDeviceTables> t(new BgzfDeviceTables);
        t->crc.upload(crc); t->x512.upload(x512); t->x8.upload(x8);
        per_device[device] = std::move(t);
    }
    BgzfTables T;
    T.crc_tab = per_device[device]->crc.as<uint32_t>();
    T.x512 = per_device[device]->x512.as<uint32_t>();
    T.x8 = per_device[device]->x8.as<uint32_t>();
    return T;
}

// d_src[0..n) -> complete BGZF file image (blocks + end-of-file block) at d_dst; returns its size.
// Works through the input in groups of blocks so that the slot scratch stays at 512 MiB.
static uint64_t bgzf_deflate_device(int device, hipStream_t stream, const uint8_t* d_src, uint64_t n, uint8_t* d_dst,
                                    uint64_t cap, double* ms) {
    if (reinterpret_cast<uintptr_t>(d_src) & 15u) throw Error(JK_ERR_ARG, "BGZF input must be 16-byte aligned");
    if (cap < bgzf_bound(n)) throw Error(JK_ERR_ARG, "BGZF destination smaller than jk_bgzf_bound()");
    const BgzfTables T = bgzf_tables(device);
    const uint64_t n_blocks = (n + BGZF_BLOCK_IN - 1) / BGZF_BLOCK_IN;
    const uint64_t GROUP = 8192;
    const uint64_t n_groups = (n_blocks + GROUP - 1) / GROUP;
    DevBuf slots, sizes, offs, sums, base;
    const uint64_t g_blocks = std::min<uint64_t>(GROUP, std::max<uint64_t>(n_blocks, 1));
    slots.alloc(g_blocks * BGZF_SLOT);
    sizes.alloc(g_blocks * 8); offs.alloc(g_blocks * 8);
    sums.alloc(((g_blocks + SCAN_BLOCK - 1) / SCAN_BLOCK) * 8);
    base.alloc((n_groups + 1) * 8);
    JK_HIP(hipMemsetAsync(base.p, 0, (n_groups + 1) * 8, stream));
    hipEvent_t e0, e1;
    JK_HIP(hipEventCreate(&e0)); JK_HIP(hipEventCreate(&e1));
    JK_HIP(hipEventRecord(e0, stream));
    for (uint64_t g = 0; g < n_groups; g++) {
        const uint64_t b0 = g * GROUP;
        const uint32_t nb = (uint32_t)std::min<uint64_t>(GROUP, n_blocks - b0);
        const uint64_t off = b0 * BGZF_BLOCK_IN;
        hipLaunchKernelGGL(bgzf_deflate_kernel, dim3(nb), dim3(BGZF_THREADS), 0, stream, d_src + off, n - off,
                           slots.as<uint8_t>(), sizes.as<uint64_t>(), T);
        const uint32_t nsb = (nb + SCAN_BLOCK - 1) / SCAN_BLOCK;
        hipLaunchKernelGGL(scan_block_kernel, dim3(nsb), dim3(SCAN_BLOCK), 0, stream, sizes.as<uint64_t>(), offs.as<uint64_t>(),
                           sums.as<uint64_t>(), nb);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_BLOCK), 0, stream, sums.as<uint64_t>(), nsb, base.as<uint64_t>() + g);
        hipLaunchKernelGGL(scan_add_kernel, dim3(nsb), dim3(SCAN_BLOCK), 0, stream, offs.as<uint64_t>(), sums.as<uint64_t>(), nb);
        hipLaunchKernelGGL(bgzf_gather_kernel, dim3(nb), dim3(256), 0, stream, slots.as<uint8_t>(), sizes.as<uint64_t>(),
                           offs.as<uint64_t>(), d_dst, base.as<uint64_t>() + g);
    }
    JK_HIP(hipGetLastError());
    JK_HIP(hipEventRecord(e1, stream));
    uint64_t total = 0;
    JK_HIP(hipMemcpyAsync(&total, base.as<uint64_t>() + n_groups, 8, hipMemcpyDeviceToHost, stream));
    JK_HIP(hipStreamSynchronize(stream));
    JK_HIP(hipMemcpy(d_dst + total, kBgzfEof, sizeof(kBgzfEof), hipMemcpyHostToDevice));
    if (ms) { float t = 0; JK_HIP(hipEventElapsedTime(&t, e0, e1)); *ms = t; }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return total + sizeof(kBgzfEof);
}

// Device image -> host consumer through two pinned buffers: the copy of piece k+1 runs while `sink`
// works on piece k (file write, zlib).  `piece` is a whole number of BGZF input blocks.
template <typename Sink>
static void stream_to_host(const uint8_t* d_src, uint64_t n, size_t piece, Sink&& sink) {
    struct Pinned {
        void* p[2] = {nullptr, nullptr};
        hipStream_t st = nullptr;
        hipEvent_t ev[2] = {nullptr, nullptr};
        ~Pinned() {
            for (int k = 0; k < 2; k++) { if (p[k]) (void)hipHostFree(p[k]); if (ev[k]) (void)hipEventDestroy(ev[k]); }
            if (st) (void)hipStreamDestroy(st);
        }
    } P;
    if (n == 0) return;
    piece = (size_t)std::min<uint64_t>(piece, n);
    JK_HIP(hipStreamCreateWithFlags(&P.st, hipStreamNonBlocking));
    for (int k = 0; k < 2; k++) { JK_HIP(hipHostMalloc(&P.p[k], piece, hipHostMallocDefault)); JK_HIP(hipEventCreate(&P.ev[k])); }
    const uint64_t n_pieces = (n + piece - 1) / piece;
    auto issue = [&](uint64_t k) {
        const uint64_t off = k * piece;
        JK_HIP(hipMemcpyAsync(P.p[k & 1], d_src + off, (size_t)std::min<uint64_t>(piece, n - off), hipMemcpyDeviceToHost, P.st));
        JK_HIP(hipEventRecord(P.ev[k & 1], P.st));
    };
    issue(0);
    for (uint64_t k = 0; k < n_pieces; k++) {
        JK_HIP(hipEventSynchronize(P.ev[k & 1]));
        if (k + 1 < n_pieces) issue(k + 1);
        sink(static_cast<const uint8_t*>(P.p[k & 1]), (size_t)std::min<uint64_t>(piece, n - k * piece));
    }
}

static void write_files(const jk_session& s) {
    if (!s.generated) throw Error(JK_ERR_ARG, "jk_session_write before jk_session_generate");
    const size_t CH = BGZF_IN * 1024;                     // 66.8 MB, a whole number of BGZF blocks
    const unsigned n_thr = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    for (uint32_t e = 0; e < s.n_ends; e++) {
        std::string fn = s.out_prefix + "_R" + std::to_string(e + 1) + ".fq";
        if (s.compress > 0) fn += ".gz";
        struct Files {
            FILE* f = nullptr; gzFile gz = nullptr;
            ~Files() { if (f) std::fclose(f); if (gz) gzclose(gz); }
        } F;
        if (s.compress > 0 && !s.bgzip) {
            const std::string mode = "wb" + std::to_string(s.compress);
            F.gz = gzopen(fn.c_str(), mode.c_str());
            if (!F.gz) throw Error(JK_ERR_IO, "gzopen of " + fn + " failed.\n");
        } else {
            F.f = std::fopen(fn.c_str(), "wb");
            if (!F.f) throw Error(JK_ERR_IO, "Unable to open file " + fn + ".\n");
        }
        auto put = [&](const uint8_t* p, size_t n) {
            if (std::fwrite(p, 1, n, F.f) != n) throw Error(JK_ERR_IO, "short write to " + fn);
        };
        const uint8_t* d_img = s.d_out[e].as<uint8_t>();
        if (s.compress == 0) {
            stream_to_host(d_img, s.bytes[e], CH, put);
        } else if (!s.bgzip) {
            stream_to_host(d_img, s.bytes[e], CH, [&](const uint8_t* p, size_t n) {
                if (gzwrite(F.gz, p, (unsigned)n) != (int)n) throw Error(JK_ERR_IO, "gzwrite to " + fn + " failed");
            });
        } else if (!s.host_deflate) {
            // BGZF blocks made on the device; only the compressed image crosses the host link
            DevBuf comp;
            comp.alloc(bgzf_bound(s.bytes[e]));
            const uint64_t n_comp = bgzf_deflate_device(s.device, s.stream, d_img, s.bytes[e], comp.as<uint8_t>(), comp.n, nullptr);
            stream_to_host(comp.as<uint8_t>(), n_comp, CH, put);
        } else {
            stream_to_host(d_img, s.bytes[e], CH, [&](const uint8_t* buf, size_t n) {
                const size_t n_blocks = (n + BGZF_IN - 1) / BGZF_IN;
                std::vector<std::vector<uint8_t>> parts(n_thr);
                std::vector<std::string> errs(n_thr);
                std::vector<std::thread> pool;
                for (unsigned t = 0; t < n_thr; t++) pool.emplace_back([&, t] {
                    try {
                        const size_t b0 = n_blocks * t / n_thr, b1 = n_blocks * (t + 1) / n_thr;
                        for (size_t b = b0; b < b1; b++)
                            bgzf_compress_block(buf + b * BGZF_IN, std::min(BGZF_IN, n - b * BGZF_IN), s.compress, parts[t]);
                    } catch (const std::exception& ex) { errs[t] = ex.what(); }
                });
                for (std::thread& th : pool) th.join();
                for (unsigned t = 0; t < n_thr; t++) {
                    if (!errs[t].empty()) throw Error(JK_ERR_IO, errs[t]);
                    if (!parts[t].empty()) put(parts[t].data(), parts[t].size());
                }
            });
            put(kBgzfEof, sizeof(kBgzfEof));
        }
        if (F.f) { FILE* f = F.f; F.f = nullptr; if (std::fclose(f) != 0) throw Error(JK_ERR_IO, "error closing " + fn); }
        if (F.gz) { gzFile g = F.gz; F.gz = nullptr; if (gzclose(g) != Z_OK) throw Error(JK_ERR_IO, "error closing " + fn); }
    }
}

// ---- primitive evaluation hooks ---------------------------------------------------------------
static double g_eval_shape = 16.0, g_eval_scale = 25.0;

struct EvalRng { jk_pcg64 e; JK_HD uint64_t operator()() { return jk_pcg_next(e); } };

JK_HD void eval_one(int what, const uint64_t* in, uint64_t i, uint64_t aux, uint64_t* out, jk_gamma_param gp) {
    switch (what) {
        case JK_OP_PCG_STREAM: {
            uint32_t w[8];
            for (int k = 0; k < 8; k++) w[k] = (uint32_t)in[i * 8 + k];
            jk_pcg64 e = jk_pcg_seed(w);
            for (uint64_t k = 0; k < aux; k++) out[i * aux + k] = jk_pcg_next(e);
            break;
        }
        case JK_OP_RUNIF_INDEX: out[i] = jk_runif_index(in[i], aux); break;
        case JK_OP_RUNIF_DOUBLE: out[i] = jk_d2u(jk_runif_double(in[i])); break;
        case JK_OP_CANONICAL: out[i] = jk_d2u(jk_canonical(in[i])); break;
        case JK_OP_N_QUAL: out[i] = jk_n_qual(in[i]); break;
        case JK_OP_LT_HALF: out[i] = jk_runif_lt_half(in[i]) ? 1 : 0; break;
        case JK_OP_FRAG_START: out[i] = jk_frag_start(in[i], aux); break;
        case JK_OP_LOG: out[i] = jk_d2u(jk_log(jk_u2d(in[i]))); break;
        case JK_OP_SQRT: out[i] = jk_d2u(jk_sqrt(jk_u2d(in[i]))); break;
        case JK_OP_GAMMA_STREAM: {
            uint32_t w[8];
            for (int k = 0; k < 8; k++) w[k] = (uint32_t)in[i * 8 + k];
            EvalRng r; r.e = jk_pcg_seed(w);
            jk_gamma_state st; st.saved = 0; st.saved_available = 0;
            for (uint64_t k = 0; k < aux; k++) out[i * aux + k] = jk_d2u(jk_gamma(gp, st, r));
            break;
        }
        case JK_OP_EXP: { double r = 0; bool ok = jk_exp(jk_u2d(in[i]), &r); out[i] = ok ? jk_d2u(r) : ~0ULL; break; }
        case JK_OP_POW: { bool ok = true; double r = jk_pow(jk_u2d(in[2 * i]), jk_u2d(in[2 * i + 1]), &ok); out[i] = ok ? jk_d2u(r) : ~0ULL; break; }
        case JK_OP_LOG10: out[i] = jk_d2u(jk_log10(jk_u2d(in[i]))); break;
        case JK_OP_QNORM: out[i] = jk_d2u(jk_qnorm(jk_u2d(in[i]))); break;
        case JK_OP_RUNIF_AB: {
            jk_x87 c; c.m = in[4 * i + 2]; c.e = (int32_t)(int64_t)in[4 * i + 3];
            out[i] = jk_d2u(jk_runif_ab(in[4 * i], jk_x87_from_double(jk_u2d(in[4 * i + 1])), c));
            break;
        }
        case JK_OP_RUNIF_INDEX32:          // the kernels' 32-bit form of RUNIF_INDEX (n < 2^32), device only
#if defined(__HIP_DEVICE_COMPILE__)
            out[i] = runif_index32(in[i], (uint32_t)aux);
#else
            out[i] = jk_runif_index(in[i], aux);
#endif
            break;
        default: break;
    }
}

__global__ void eval_kernel(int what, const uint64_t* in, uint64_t n, uint64_t aux, uint64_t* out, jk_gamma_param gp) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) eval_one(what, in, i, aux, out, gp);
}

static jk_gamma_param eval_gamma_param() {
    jk_gamma_param gp;
    gp.a1 = g_eval_shape - 1.0 / 3.0;
    gp.a2 = 1.0 / std::sqrt(9.0 * gp.a1);
    gp.beta = g_eval_scale;
    return gp;
}

static void eval_sizes(int what, uint64_t n, uint64_t aux, uint64_t* n_in, uint64_t* n_out) {
    const bool stream = (what == JK_OP_PCG_STREAM || what == JK_OP_GAMMA_STREAM);
    *n_in = stream ? n * 8 : (what == JK_OP_POW ? n * 2 : (what == JK_OP_RUNIF_AB ? n * 4 : n));
    *n_out = stream ? n * aux : n;
}

template <typename F>
static int guarded(F f) {
    try { f(); g_last_error.clear(); return JK_OK; }
    catch (const Error& e) { g_last_error = e.what(); return e.code; }
    catch (const std::bad_alloc&) { g_last_error = "out of host memory"; return JK_ERR_DEVICE; }
    catch (const std::exception& e) { g_last_error = e.what(); return JK_ERR_ARG; }
}

}  // namespace jk

extern "C" {

const char* jk_last_error(void) { return g_last_error.c_str(); }
const char* jk_version(void) { return "jackalope_hip 0.1 (gfx950)"; }

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

static void generate_with_retry(jk_session& s) {
    for (int attempt = 0;; attempt++) {
        try { launch_generate(s); return; }
        catch (const Error& e) {
            if (e.code != JK_ERR_RETRY || attempt >= 6 || !s.replan) {
                if (e.code == JK_ERR_RETRY) throw Error(JK_ERR_DEVICE, "PacBio pools overflowed even after growing them");
                throw;
            }
            s.pool_scale *= 2;
            s.replan();
        }
    }
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

int jk_pacbio_ref(const jk_ref_genome* genome, const jk_pacbio_args* args) {
    jk_session* s = nullptr;
    int rc = jk_pacbio_ref_open(genome, args, &s);
    if (rc == JK_OK) rc = jk_session_generate(s);
    if (rc == JK_OK) rc = jk_session_write(s);
    std::string keep = g_last_error;
    jk_session_close(s);
    g_last_error = keep;
    return rc;
}

int jk_pacbio_hap(const jk_hap_set* haps, const jk_pacbio_args* args) {
    return guarded([&] {
        if (!haps || !args) throw Error(JK_ERR_ARG, "NULL argument");
        SeedReader seeds{args->seeds};
        const std::vector<double> probs = pb_hap_probs_of(*haps, *args);
        if (!args->sep_files) {
            std::unique_ptr<jk_session> s(new jk_session());
            open_pacbio_hap(*s, *haps, *args, probs, args->n_reads, seeds);
            generate_with_retry(*s);
            write_files(*s);
            return;
        }
        std::vector<uint64_t> per_file = reads_per_group(args->n_reads, probs, seeds);   // src/hts.h:526-529, one read end
        for (uint64_t h = 0; h < haps->n_haps; h++) {
            if (args->abort_flag && *args->abort_flag) throw Error(JK_ERR_ABORTED, "aborted");
            std::vector<double> one_hot(haps->n_haps, 0.0);
            one_hot[h] = 1;
            std::unique_ptr<jk_session> s(new jk_session());
            open_pacbio_hap(*s, *haps, *args, one_hot, per_file[h], seeds);
            s->out_prefix += std::string("_") + (haps->hap_names ? haps->hap_names[h] : "");
            generate_with_retry(*s);
            write_files(*s);
        }
    });
}

int jk_session_generate(jk_session* s) {
    return guarded([&] { if (!s) throw Error(JK_ERR_ARG, "NULL session"); generate_with_retry(*s); });
}

int jk_session_sizes(const jk_session* s, uint64_t bytes[2], uint64_t* reads, uint32_t* n_ends) {
    return guarded([&] {
        if (!s || !s->generated) throw Error(JK_ERR_ARG, "session has not generated yet");
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

int jk_session_write(const jk_session* s) {
    return guarded([&] { if (!s) throw Error(JK_ERR_ARG, "NULL session"); JK_HIP(hipSetDevice(s->device)); write_files(*s); });
}

int jk_session_timing(const jk_session* s, double ms[3]) {
    return guarded([&] {
        if (!s || !s->generated) throw Error(JK_ERR_ARG, "session has not generated yet");
        ms[0] = s->ms[0]; ms[1] = s->ms[1]; ms[2] = s->ms[2];
    });
}

uint64_t jk_session_seed_words_used(const jk_session* s) { return s ? s->seed_words_used : 0; }
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

int jk_illumina_ref(const jk_ref_genome* genome, const jk_illumina_args* args) {
    jk_session* s = nullptr;
    int rc = jk_illumina_ref_open(genome, args, &s);
    if (rc == JK_OK) rc = jk_session_generate(s);
    if (rc == JK_OK) rc = jk_session_write(s);
    std::string keep = g_last_error;
    jk_session_close(s);
    g_last_error = keep;
    return rc;
}

int jk_illumina_hap(const jk_hap_set* haps, const jk_illumina_args* args) {
    return guarded([&] {
        if (!haps || !args) throw Error(JK_ERR_ARG, "NULL argument");
        SeedReader seeds{args->seeds};
        const std::vector<double> probs = hap_probs_of(*haps, *args);
        if (!args->sep_files) {
            std::unique_ptr<jk_session> s(new jk_session());
            open_illumina_hap(*s, *haps, *args, probs, args->n_reads, seeds);
            launch_generate(*s);
            write_files(*s);
            return;
        }
        // write_reads_cpp_sep_files_ (src/hts.h:512-552): reads per file, then one run per haplotype with
        // one-hot haplotype probabilities and prefix <out_prefix>_<haplotype>
        const uint64_t n_ends = args->paired ? 2 : 1;
        std::vector<uint64_t> per_file = reads_per_group(args->n_reads / n_ends, probs, seeds);
        for (uint64_t& v : per_file) v *= n_ends;
        for (uint64_t h = 0; h < haps->n_haps; h++) {
            if (args->abort_flag && *args->abort_flag) throw Error(JK_ERR_ABORTED, "aborted");
            std::vector<double> one_hot(haps->n_haps, 0.0);
            one_hot[h] = 1;
            std::unique_ptr<jk_session> s(new jk_session());
            open_illumina_hap(*s, *haps, *args, one_hot, per_file[h], seeds);
            s->out_prefix += std::string("_") + (haps->hap_names ? haps->hap_names[h] : "");
            launch_generate(*s);
            write_files(*s);
        }
    });
}

void jk_split_int(uint64_t x, uint64_t n, uint64_t* out) {
    std::vector<uint64_t> v = split_int(x, n);
    for (uint64_t i = 0; i < n; i++) out[i] = v[i];
}

int jk_reads_per_group(uint64_t n_reads, const double* probs, uint64_t n, jk_seed_source* seeds, uint64_t* out) {
    return guarded([&] {
        if (!seeds) throw Error(JK_ERR_ARG, "NULL seeds");
        SeedReader r{*seeds};
        std::vector<uint64_t> v = reads_per_group(n_reads, std::vector<double>(probs, probs + n), r);
        for (uint64_t i = 0; i < n; i++) out[i] = v[i];
        if (seeds->words) { seeds->words += r.pos; seeds->n_words -= r.pos; }
    });
}

void jk_alias_build(const double* probs, uint64_t n, double* Prob, uint64_t* Alias) {
    AliasTable t = alias_build(std::vector<double>(probs, probs + n));
    for (uint64_t i = 0; i < n; i++) { Prob[i] = t.prob[i]; Alias[i] = t.alias[i]; }
}

// HapChrom::get_chrom_full (src/hap_classes.cpp:80-116) on the host, from the flat view: walks the
// mutations in order and copies reference runs and mutation bytes.
int jk_hap_chrom_full(const jk_hap_set* hs, uint64_t hap, uint64_t chrom, char* out, uint64_t cap) {
    return guarded([&] {
        if (!hs || hap >= hs->n_haps || chrom >= hs->ref.n_chroms) throw Error(JK_ERR_ARG, "bad haplotype/chromosome index");
        const uint64_t nc = hs->ref.n_chroms, cell = hap * nc + chrom;
        uint64_t m0 = 0;
        for (uint64_t k = 0; k < cell; k++) m0 += hs->n_mut[k];
        const uint64_t m1 = m0 + hs->n_mut[cell];
        const uint64_t size = hs->chrom_size[cell], ref_len = hs->ref.chrom_lens[chrom];
        if (cap < size) throw Error(JK_ERR_ARG, "destination too small");
        if (hs->ref.seqs_on_device) throw Error(JK_ERR_UNSUPPORTED, "jk_hap_chrom_full reads the reference on the host; this one is in device memory");
        const char* ref = hs->ref.chrom_seqs[chrom];
        uint64_t pos = 0;
        const uint64_t first = m0 < m1 ? hs->new_pos[m0] : size;
        for (; pos < first; pos++) out[pos] = ref[pos];
        for (uint64_t m = m0; m < m1; m++) {
            int64_t smod = (m + 1 < m1) ? (int64_t)(hs->new_pos[m + 1] - hs->old_pos[m + 1]) : (int64_t)(size - ref_len);
            smod += (int64_t)(hs->old_pos[m] - hs->new_pos[m]);
            const uint64_t end = (m + 1 < m1) ? hs->new_pos[m + 1] : size;
            for (; pos < end; pos++) {
                const uint64_t ind = pos - hs->new_pos[m];
                if ((int64_t)ind > smod) out[pos] = ref[ind + hs->old_pos[m] - smod];
                else out[pos] = hs->nuc_blob[hs->nuc_off[m] + ind];
            }
        }
    });
}

// ---- mutation-table builder (host) ----
struct jk_hap_builder {
    uint64_t n_haps = 0, n_chroms = 0;
    std::vector<std::string> chrom_names, hap_names;
    std::string ref_name;
    std::vector<const char*> seqs;
    std::vector<uint64_t> lens;
    std::vector<jk::HapCell> cells;          // [hap * n_chroms + chrom]
    // flat view, rebuilt by jk_hap_builder_view
    std::vector<const char*> v_chrom_names, v_hap_names;
    std::vector<uint64_t> v_size, v_nmut, v_op, v_np, v_off;
    std::string v_blob;
};

static jk_hap_builder* builder_shell(const jk_ref_genome* ref, uint64_t n_haps, const char* const* hap_names) {
    if (!ref) throw Error(JK_ERR_ARG, "NULL reference genome");
    if (ref->seqs_on_device) throw Error(JK_ERR_UNSUPPORTED, "the mutation-table builder reads reference bases on the host; this genome is in device memory (fetch it first)");
    std::unique_ptr<jk_hap_builder> b(new jk_hap_builder);
    b->n_haps = n_haps;
    b->n_chroms = ref->n_chroms;
    b->ref_name = ref->name ? ref->name : "REF";
    for (uint64_t c = 0; c < ref->n_chroms; c++) {
        b->chrom_names.push_back(ref->chrom_names && ref->chrom_names[c] ? ref->chrom_names[c] : "chrom" + std::to_string(c));
        b->seqs.push_back(ref->chrom_seqs[c]);
        b->lens.push_back(ref->chrom_lens[c]);
    }
    // HapSet(ref, n) names haplotypes hap0.. (src/hap_classes.h:546-550)
    for (uint64_t h = 0; h < n_haps; h++)
        b->hap_names.push_back(hap_names && hap_names[h] ? hap_names[h] : "hap" + std::to_string(h));
    b->cells.resize(n_haps * ref->n_chroms);
    for (uint64_t h = 0; h < n_haps; h++)
        for (uint64_t c = 0; c < ref->n_chroms; c++) {
            jk::HapCell& cell = b->cells[h * ref->n_chroms + c];
            cell.ref = ref->chrom_seqs[c];
            cell.ref_len = cell.size = ref->chrom_lens[c];
        }
    return b.release();
}

int jk_hap_builder_new(const jk_ref_genome* ref, uint64_t n_haps, jk_hap_builder** out) {
    return guarded([&] {
        if (!out) throw Error(JK_ERR_ARG, "NULL output pointer");
        *out = builder_shell(ref, n_haps, nullptr);
    });
}

int jk_hap_builder_from(const jk_hap_set* hs, jk_hap_builder** out) {
    return guarded([&] {
        if (!hs || !out) throw Error(JK_ERR_ARG, "NULL haplotype set / output pointer");
        if (hs->n_chroms != hs->ref.n_chroms) throw Error(JK_ERR_ARG, "haplotype set and reference differ in chromosome count");
        std::unique_ptr<jk_hap_builder> b(builder_shell(&hs->ref, hs->n_haps, hs->hap_names));
        uint64_t m = 0;
        for (uint64_t k = 0; k < hs->n_haps * hs->n_chroms; k++) {
            jk::HapCell& cell = b->cells[k];
            cell.size = hs->chrom_size[k];
            for (uint64_t i = 0; i < hs->n_mut[k]; i++, m++) {
                cell.op.push_back(hs->old_pos[m]);
                cell.np.push_back(hs->new_pos[m]);
                cell.nt.emplace_back(hs->nuc_blob + hs->nuc_off[m], hs->nuc_blob + hs->nuc_off[m + 1]);
            }
        }
        *out = b.release();
    });
}

static jk::HapCell& builder_cell(jk_hap_builder* b, uint64_t hap, uint64_t chrom) {
    if (!b) throw Error(JK_ERR_ARG, "NULL builder");
    if (hap >= b->n_haps) throw Error(JK_ERR_ARG, "hap_ind out of range");
    if (chrom >= b->n_chroms) throw Error(JK_ERR_ARG, "chrom_ind out of range");
    return b->cells[hap * b->n_chroms + chrom];
}

// message of HapChrom::get_mut_ (src/hap_classes.cpp:731-735)
static const char* const kNewPosMsg = "new_pos should never be >= the chromosome size. "
    "Either re-calculate the chromosome size or closely examine new_pos.";

int jk_add_substitution(jk_hap_builder* b, uint64_t hap, uint64_t chrom, char nucleo, uint64_t new_pos) {
    return guarded([&] {
        jk::HapCell& cell = builder_cell(b, hap, chrom);
        if (new_pos >= cell.size || !cell.substitute(nucleo, new_pos)) throw Error(JK_ERR_ARG, kNewPosMsg);
    });
}

int jk_add_insertion(jk_hap_builder* b, uint64_t hap, uint64_t chrom, const char* nucleos, uint64_t new_pos) {
    return guarded([&] {
        jk::HapCell& cell = builder_cell(b, hap, chrom);
        if (!nucleos) throw Error(JK_ERR_ARG, "NULL nucleotides");
        if (new_pos >= cell.size || !cell.insert(nucleos, new_pos)) throw Error(JK_ERR_ARG, kNewPosMsg);
    });
}

int jk_add_deletion(jk_hap_builder* b, uint64_t hap, uint64_t chrom, uint64_t size, uint64_t new_pos) {
    // size 0 or a position past the end is a silent no-op in the reference (src/hap_classes.cpp:297)
    return guarded([&] { builder_cell(b, hap, chrom).remove(size, new_pos); });
}

int jk_hap_builder_view(jk_hap_builder* b, jk_hap_set* out) {
    return guarded([&] {
        if (!b || !out) throw Error(JK_ERR_ARG, "NULL builder / output pointer");
        b->v_size.clear(); b->v_nmut.clear(); b->v_op.clear(); b->v_np.clear(); b->v_blob.clear();
        b->v_off.assign(1, 0);
        for (const jk::HapCell& cell : b->cells) {
            b->v_size.push_back(cell.size);
            b->v_nmut.push_back(cell.count());
            b->v_op.insert(b->v_op.end(), cell.op.begin(), cell.op.end());
            b->v_np.insert(b->v_np.end(), cell.np.begin(), cell.np.end());
            for (const std::string& s : cell.nt) { b->v_blob += s; b->v_off.push_back(b->v_blob.size()); }
        }
        b->v_chrom_names.clear(); b->v_hap_names.clear();
        for (const std::string& s : b->chrom_names) b->v_chrom_names.push_back(s.c_str());
        for (const std::string& s : b->hap_names) b->v_hap_names.push_back(s.c_str());
        out->n_haps = b->n_haps;
        out->n_chroms = b->n_chroms;
        out->hap_names = b->v_hap_names.data();
        out->ref.n_chroms = b->n_chroms;
        out->ref.chrom_names = b->v_chrom_names.data();
        out->ref.chrom_seqs = b->seqs.data();
        out->ref.chrom_lens = b->lens.data();
        out->ref.name = b->ref_name.c_str();
        out->ref.seqs_on_device = 0;
        out->chrom_size = b->v_size.data();
        out->n_mut = b->v_nmut.data();
        out->old_pos = b->v_op.data();
        out->new_pos = b->v_np.data();
        out->nuc_off = b->v_off.data();
        out->nuc_blob = b->v_blob.c_str();
    });
}

void jk_hap_builder_free(jk_hap_builder* b) { delete b; }

// ---- create_genome (src/create_sequences.cpp:59-169) on the device ------------------------------
struct jk_genome {
    int device = 0;
    DevBuf seq;                                   // create_genome: all chromosomes; read_fasta: see `bufs`
    std::vector<std::unique_ptr<DevBuf>> bufs;    // read_fasta: one packed buffer per file
    std::vector<const uint8_t*> ptr;              // device address of every chromosome
    std::vector<uint64_t> off, len;
    std::vector<std::string> names;
    std::vector<const char*> v_names, v_seqs;
    uint64_t seed_words_used = 0;
    double ms = 0;
};

int jk_create_genome(uint64_t n_chroms, double len_mean, double len_sd, const double* pi_tcag, uint64_t n_threads,
                     jk_seed_source* seeds, int device, jk_genome** out) {
    return guarded([&] {
        if (!out || !pi_tcag || !seeds) throw Error(JK_ERR_ARG, "NULL pointer");
        if (n_chroms == 0 || n_chroms > 0xffffffffULL) throw Error(JK_ERR_ARG, "n_chroms must be in [1, 2^32)");
        if (!(len_mean >= 1)) throw Error(JK_ERR_ARG, "len_mean must be >= 1");
        if (!(len_sd >= 0)) throw Error(JK_ERR_ARG, "len_sd must be >= 0");
        if (n_threads == 0) throw Error(JK_ERR_ARG, "n_threads must be >= 1");
        double psum = 0;
        for (int i = 0; i < 4; i++) { if (!(pi_tcag[i] >= 0)) throw Error(JK_ERR_ARG, "pi_tcag must be >= 0"); psum += pi_tcag[i]; }
        if (!(psum > 0)) throw Error(JK_ERR_ARG, "at least one of pi_tcag must be > 0");
        const double shape = (len_mean * len_mean) / (len_sd * len_sd), scale = (len_sd * len_sd) / len_mean;
        if (len_sd > 0 && shape < 1.0) throw Error(JK_ERR_UNSUPPORTED, "len_sd > len_mean (gamma shape < 1) is not implemented on the GPU path");
        JK_HIP(hipSetDevice(device));
        std::unique_ptr<jk_genome> G(new jk_genome);
        G->device = device;

        // ---- host: seeds, lengths, first state of every chromosome
        SeedReader sr{*seeds};
        const uint64_t T = n_threads;
        std::vector<uint32_t> lane_seed(T * 8);
        for (uint64_t t = 0; t < T; t++) sr.take8(&lane_seed[t * 8]);           // mt_seeds (src/pcg.h:63-71)
        G->seed_words_used = sr.pos;
        jk_gamma_param gp;
        gp.a1 = shape - 1.0 / 3.0; gp.a2 = 1.0 / std::sqrt(9.0 * gp.a1); gp.beta = scale;
        std::vector<uint64_t> len(n_chroms), start(2 * n_chroms), inc(2 * T), adv(T * 64 * 4);
        std::vector<uint32_t> lane_of(n_chroms);
        // omp for schedule(static): contiguous blocks, the first n_chroms % T threads get one more
        const std::vector<uint64_t> per_lane = split_int(n_chroms, T);
        uint64_t c = 0;
        for (uint64_t t = 0; t < T; t++) {
            HostPcg eng{jk_pcg_seed(&lane_seed[t * 8])};
            PcgMap map[64];
            pcg_advance_table(eng.e, map);
            inc[2 * t] = eng.e.inc_hi; inc[2 * t + 1] = eng.e.inc_lo;
            for (int k = 0; k < 64; k++) {
                uint64_t* a = &adv[(t * 64 + k) * 4];
                a[0] = (uint64_t)(map[k].mult >> 64); a[1] = (uint64_t)map[k].mult;
                a[2] = (uint64_t)(map[k].plus >> 64); a[3] = (uint64_t)map[k].plus;
            }
            jk_gamma_state gs{0.0, 0};
            for (uint64_t i = 0; i < per_lane[t]; i++, c++) {
                uint64_t L;
                if (len_sd > 0) {
                    const double g = jk_gamma(gp, gs, eng);
                    L = g >= 18446744073709551616.0 ? ~0ULL : (uint64_t)g;
                    if (L < 1) L = 1;
                } else L = (uint64_t)len_mean;
                if (L >= (1ULL << 62)) throw Error(JK_ERR_UNSUPPORTED, "chromosome length >= 2^62");
                len[c] = L;
                lane_of[c] = (uint32_t)t;
                start[2 * c] = eng.e.s_hi; start[2 * c + 1] = eng.e.s_lo;
                pcg_advance(eng.e, map, 2 * L);                   // AliasSampler::sample takes two outputs per base
            }
        }
        // ---- layout + device tables
        G->off.resize(n_chroms); G->len = len;
        std::vector<uint64_t> run_first(n_chroms + 1, 0);
        uint64_t total = 0;
        for (uint64_t i = 0; i < n_chroms; i++) {
            G->off[i] = total;
            total = align_up(total + len[i], 64);
            run_first[i + 1] = run_first[i] + (len[i] + GENOME_RUN - 1) / GENOME_RUN;
            G->names.push_back("chrom" + std::to_string(i));      // create_genome_cpp, src/create_sequences.cpp:163-166
        }
        G->seq.alloc(total);
        for (uint64_t i = 0; i < n_chroms; i++) G->ptr.push_back(G->seq.as<uint8_t>() + G->off[i]);
        const AliasTable at = alias_build(std::vector<double>(pi_tcag, pi_tcag + 4));
        GenomeKernelParams P{};
        for (int i = 0; i < 4; i++) {
            const Threshold th = threshold_lt(at.prob[i]);
            P.thresh[i] = th.all ? ~0ULL : th.th;
            P.alias[i] = th.all ? (uint32_t)i : (uint32_t)at.alias[i];
        }
        DevBuf d_off, d_len, d_first, d_start, d_lane, d_inc, d_adv;
        d_off.upload(G->off); d_len.upload(len); d_first.upload(run_first); d_start.upload(start);
        d_lane.upload(lane_of); d_inc.upload(inc); d_adv.upload(adv);
        P.out = G->seq.as<uint8_t>();
        P.chrom_off = d_off.as<uint64_t>(); P.chrom_len = d_len.as<uint64_t>(); P.run_first = d_first.as<uint64_t>();
        P.start_state = d_start.as<uint64_t>(); P.chrom_lane = d_lane.as<uint32_t>();
        P.lane_inc = d_inc.as<uint64_t>(); P.lane_adv = d_adv.as<uint64_t>();
        P.n_runs = run_first[n_chroms]; P.n_chroms = (uint32_t)n_chroms;
        const uint64_t grid = (P.n_runs + GENOME_BLOCK - 1) / GENOME_BLOCK;
        if (grid > 0x7fffffffULL) throw Error(JK_ERR_UNSUPPORTED, "genome too large for one launch");
        hipEvent_t e0, e1;
        JK_HIP(hipEventCreate(&e0)); JK_HIP(hipEventCreate(&e1));
        JK_HIP(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(create_genome_kernel, dim3((uint32_t)grid), dim3(GENOME_BLOCK), 0, nullptr, P);
        JK_HIP(hipGetLastError());
        JK_HIP(hipEventRecord(e1, nullptr));
        JK_HIP(hipDeviceSynchronize());
        float t = 0;
        JK_HIP(hipEventElapsedTime(&t, e0, e1));
        G->ms = t;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        *out = G.release();
    });
}

int jk_genome_view(jk_genome* g, jk_ref_genome* view) {
    return guarded([&] {
        if (!g || !view) throw Error(JK_ERR_ARG, "NULL pointer");
        g->v_names.clear(); g->v_seqs.clear();
        for (size_t i = 0; i < g->names.size(); i++) {
            g->v_names.push_back(g->names[i].c_str());
            g->v_seqs.push_back(reinterpret_cast<const char*>(g->ptr[i]));
        }
        view->n_chroms = g->names.size();
        view->chrom_names = g->v_names.data();
        view->chrom_seqs = g->v_seqs.data();
        view->chrom_lens = g->len.data();
        view->name = "REF";
        view->seqs_on_device = 1;
    });
}

int jk_genome_fetch(const jk_genome* g, uint64_t chrom, char* dst, uint64_t cap) {
    return guarded([&] {
        if (!g || !dst) throw Error(JK_ERR_ARG, "NULL pointer");
        if (chrom >= g->len.size()) throw Error(JK_ERR_ARG, "chromosome index out of range");
        if (cap < g->len[chrom]) throw Error(JK_ERR_ARG, "destination too small");
        JK_HIP(hipSetDevice(g->device));
        if (g->len[chrom]) JK_HIP(hipMemcpy(dst, g->ptr[chrom], g->len[chrom], hipMemcpyDeviceToHost));
    });
}

// ---- read_fasta (src/io_fasta.cpp:41-169, :183-408): host reads + finds header lines, device packs ----
namespace jk {

// Whole (uncompressed) content of a file the way gzread presents it (src/io_fasta.cpp:83-96): gzip and bgzip
// members are inflated, anything else is passed through -- those files are mapped instead of copied.
struct HostText {
    const uint8_t* p = nullptr;
    size_t n = 0;
    void* map = nullptr; size_t map_len = 0;
    uint8_t* heap = nullptr;
    HostText() {}
    HostText(const HostText&) = delete;
    HostText& operator=(const HostText&) = delete;
    ~HostText() { if (map) munmap(map, map_len); std::free(heap); }
    const uint8_t* data() const { return p; }
    size_t size() const { return n; }
};

static void slurp_gz(const std::string& fn, HostText& T) {
    {
        const int fd = ::open(fn.c_str(), O_RDONLY);
        if (fd < 0) throw Error(JK_ERR_IO, "gzopen of " + fn + " failed: " + strerror(errno) + ".\n");
        uint8_t magic[2] = {0, 0};
        const ssize_t got = ::pread(fd, magic, 2, 0);
        struct stat st;
        const bool plain = !(got == 2 && magic[0] == 0x1f && magic[1] == 0x8b);
        if (plain && ::fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
            if (st.st_size > 0) {
                void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
                if (m != MAP_FAILED) {
                    ::close(fd);
                    T.map = m; T.map_len = (size_t)st.st_size; T.p = static_cast<const uint8_t*>(m); T.n = T.map_len;
                    return;
                }
            } else { ::close(fd); return; }
        }
        ::close(fd);
    }
    gzFile f = gzopen(fn.c_str(), "rb");
    if (!f) throw Error(JK_ERR_IO, "gzopen of " + fn + " failed: " + strerror(errno) + ".\n");
    gzbuffer(f, 1 << 20);
    size_t cap = 1 << 24, n = 0;
    uint8_t* buf = static_cast<uint8_t*>(std::malloc(cap));
    for (;;) {
        if (!buf) { gzclose(f); throw Error(JK_ERR_IO, "out of host memory reading " + fn); }
        const size_t want = std::min<size_t>(cap - n, 1u << 30);
        const int got = gzread(f, buf + n, (unsigned)want);
        if (got < 0) { int e; std::string m = gzerror(f, &e); gzclose(f); std::free(buf); throw Error(JK_ERR_IO, "Error: " + m + ".\n"); }
        n += (size_t)got;
        if ((size_t)got < want) break;
        if (n == cap) { cap += cap / 2; buf = static_cast<uint8_t*>(std::realloc(buf, cap)); }
    }
    gzclose(f);
    T.heap = buf; T.p = buf; T.n = n;
}

struct FastaPlan {
    std::vector<std::string> names;
    std::vector<uint64_t> iv_begin, iv_end;      // per chromosome, in chromosome order
};

// header lines of a non-indexed file (parse_fasta_line, src/io_fasta.cpp:43-65)
static FastaPlan plan_noind(const HostText& text, bool cut_names) {
    FastaPlan P;
    const uint8_t* t = text.data();
    const uint64_t n = text.size();
    uint64_t at = 0;
    bool first = true;
    while (at < n) {
        const uint8_t* q = static_cast<const uint8_t*>(std::memchr(t + at, '>', n - at));
        if (!q) break;
        const uint64_t g = (uint64_t)(q - t);
        uint64_t ls = g;                           // the whole line that holds this '>'
        while (ls > at && t[ls - 1] != '\n') ls--;
        const uint8_t* e = static_cast<const uint8_t*>(std::memchr(q, '\n', n - g));
        const uint64_t le = e ? (uint64_t)(e - t) : n;
        if (first) {
            for (uint64_t i = 0; i < ls; i++)
                if (t[i] != '\n' && t[i] != '\r') throw Error(JK_ERR_ARG, "FASTA file has sequence data before the first '>' line");
            first = false;
        } else {
            P.iv_end.push_back(ls);
        }
        std::string line(reinterpret_cast<const char*>(t + ls), le - ls);
        if (e && !line.empty() && line.back() == '\r') line.pop_back();
        std::string name;
        if (cut_names) {
            std::string::size_type spc = line.find(' ', 2);
            if (spc == std::string::npos) spc = line.size();
            name = line.substr(1, spc);
            name.erase(std::remove_if(name.begin(), name.end(), ::isspace), name.end());
        } else name = line.substr(1, line.size());
        P.names.push_back(name);
        P.iv_begin.push_back(e ? le + 1 : n);
        at = e ? le + 1 : n;
    }
    if (first) {
        for (uint64_t i = 0; i < n; i++)
            if (t[i] != '\n' && t[i] != '\r') throw Error(JK_ERR_ARG, "FASTA file has sequence data before the first '>' line");
    } else {
        P.iv_end.push_back(n);
    }
    return P;
}

// spans an index file describes (parse_line_fai / append_ref_ind, src/io_fasta.cpp:183-200, :270-370)
static FastaPlan plan_ind(const std::string& fai, uint64_t n) {
    FastaPlan P;
    HostText idx;
    slurp_gz(fai, idx);
    size_t at = 0;
    while (at <= idx.size()) {
        const uint8_t* e = at < idx.size() ? static_cast<const uint8_t*>(std::memchr(idx.data() + at, '\n', idx.size() - at)) : nullptr;
        const size_t le = e ? (size_t)(e - idx.data()) : idx.size();
        std::string line(reinterpret_cast<const char*>(idx.data() + at), le - at);
        if (e && !line.empty() && line.back() == '\r') line.pop_back();
        at = le + 1;
        if (line.empty()) continue;
        std::vector<std::string> cols(1, "");
        for (char ch : line) { if (ch == '\t') cols.push_back(""); else cols.back() += ch; }
        if (cols.size() < 4) throw Error(JK_ERR_ARG, "fasta index line has fewer than 4 tab-separated fields");
        uint64_t length, offset, line_len;
        try { length = std::stoull(cols[1]); offset = std::stoull(cols[2]); line_len = std::stoul(cols[3]); }
        catch (const std::exception&) { throw Error(JK_ERR_ARG, "fasta index line is not numeric"); }
        if (line_len == 0) throw Error(JK_ERR_ARG, "fasta index line length is 0");
        P.names.push_back(cols[0]);
        const uint64_t span = length + length / line_len;      // what the reference reads: len - 1 bytes from `offset`
        P.iv_begin.push_back(std::min(offset, n));
        P.iv_end.push_back(std::min(offset + span, n));
    }
    return P;
}

// pack one file's text; appends its chromosomes to G
static void fasta_pack_file(jk_genome& G, const HostText& text, const FastaPlan& plan, bool strip_cr, bool upper, double* ms) {
    const uint64_t n = text.size(), nc = plan.names.size();
    if (nc == 0) return;
    if (n && std::memchr(text.data(), 0, n)) throw Error(JK_ERR_UNSUPPORTED, "FASTA file contains NUL bytes (the reference truncates its read buffer there)");
    // intervals sorted by position (index files need not list chromosomes in file order)
    std::vector<uint32_t> order(nc);
    for (uint32_t i = 0; i < nc; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return plan.iv_begin[a] < plan.iv_begin[b]; });
    std::vector<uint64_t> ib(nc), ie(nc);
    for (uint32_t k = 0; k < nc; k++) {
        ib[k] = plan.iv_begin[order[k]]; ie[k] = plan.iv_end[order[k]];
        if (ie[k] < ib[k]) ie[k] = ib[k];
        if (k && ib[k] < ie[k - 1]) throw Error(JK_ERR_UNSUPPORTED, "fasta index entries overlap");
    }
    DevBuf d_text, d_ib, d_ie, d_cnt, d_off, d_sums, d_base, d_ivout;
    d_text.alloc(align_up(n, 16) + 64);
    JK_HIP(hipMemset(d_text.as<uint8_t>() + (n & ~15ULL), 0, d_text.n - (n & ~15ULL)));
    if (n) JK_HIP(hipMemcpy(d_text.p, text.data(), n, hipMemcpyHostToDevice));
    d_ib.upload(ib); d_ie.upload(ie);
    const uint64_t n_blocks = (n + FASTA_BLOCK_BYTES - 1) / FASTA_BLOCK_BYTES + 1;     // + 1: a block that owns position n
    if (n_blocks > 0x7fffffffULL) throw Error(JK_ERR_UNSUPPORTED, "FASTA file too large for one launch");
    d_cnt.alloc(n_blocks * 8); d_off.alloc(n_blocks * 8);
    const uint32_t nsb = (uint32_t)((n_blocks + SCAN_BLOCK - 1) / SCAN_BLOCK);
    d_sums.alloc((uint64_t)nsb * 8); d_base.alloc(16); d_ivout.alloc(nc * 8);
    JK_HIP(hipMemset(d_base.p, 0, 16));
    JK_HIP(hipMemset(d_ivout.p, 0xff, nc * 8));
    FastaParams P{};
    P.text = d_text.as<uint8_t>(); P.n = n;
    P.iv_begin = d_ib.as<uint64_t>(); P.iv_end = d_ie.as<uint64_t>(); P.n_iv = (uint32_t)nc;
    P.strip_cr = strip_cr ? 1 : 0; P.upper = upper ? 1 : 0;
    P.block_cnt = d_cnt.as<uint64_t>(); P.block_off = d_off.as<uint64_t>(); P.iv_out = d_ivout.as<uint64_t>();
    hipEvent_t e0, e1;
    JK_HIP(hipEventCreate(&e0)); JK_HIP(hipEventCreate(&e1));
    JK_HIP(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL(fasta_count_kernel, dim3((uint32_t)n_blocks), dim3(FASTA_THREADS), 0, nullptr, P);
    hipLaunchKernelGGL(scan_block_kernel, dim3(nsb), dim3(SCAN_BLOCK), 0, nullptr, d_cnt.as<uint64_t>(), d_off.as<uint64_t>(), d_sums.as<uint64_t>(), (uint32_t)n_blocks);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_BLOCK), 0, nullptr, d_sums.as<uint64_t>(), nsb, d_base.as<uint64_t>());
    hipLaunchKernelGGL(scan_add_kernel, dim3(nsb), dim3(SCAN_BLOCK), 0, nullptr, d_off.as<uint64_t>(), d_sums.as<uint64_t>(), (uint32_t)n_blocks);
    JK_HIP(hipGetLastError());
    uint64_t base_h[2] = {0, 0};
    JK_HIP(hipMemcpy(base_h, d_base.p, 16, hipMemcpyDeviceToHost));
    const uint64_t total = base_h[1];
    std::unique_ptr<DevBuf> out(new DevBuf);
    out->alloc(total + 64);
    P.out = out->as<uint8_t>();
    hipLaunchKernelGGL(fasta_pack_kernel, dim3((uint32_t)n_blocks), dim3(FASTA_THREADS), 0, nullptr, P);
    JK_HIP(hipGetLastError());
    JK_HIP(hipEventRecord(e1, nullptr));
    JK_HIP(hipDeviceSynchronize());
    float t = 0;
    JK_HIP(hipEventElapsedTime(&t, e0, e1));
    if (ms) *ms += t;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    std::vector<uint64_t> ivout(nc);
    JK_HIP(hipMemcpy(ivout.data(), d_ivout.p, nc * 8, hipMemcpyDeviceToHost));
    std::vector<uint64_t> off(nc), len(nc);
    for (uint32_t k = 0; k < nc; k++) if (ivout[k] == ~0ULL) ivout[k] = total;       // interval begins past every block
    for (uint32_t k = 0; k < nc; k++) {
        const uint64_t next = k + 1 < nc ? ivout[k + 1] : total;
        off[order[k]] = ivout[k]; len[order[k]] = next - ivout[k];
    }
    for (uint32_t i = 0; i < nc; i++) {
        G.names.push_back(plan.names[i]);
        G.off.push_back(off[i]); G.len.push_back(len[i]);
        G.ptr.push_back(out->as<uint8_t>() + off[i]);
    }
    G.bufs.push_back(std::move(out));
}

}  // namespace jk

int jk_read_fasta(const char* const* fasta_files, const char* const* fai_files, uint64_t n_files, int32_t cut_names,
                  int32_t remove_soft_mask, int device, jk_genome** out) {
    return guarded([&] {
        if (!out || !fasta_files) throw Error(JK_ERR_ARG, "NULL pointer");
        JK_HIP(hipSetDevice(device));
        std::unique_ptr<jk_genome> G(new jk_genome);
        G->device = device;
        for (uint64_t f = 0; f < n_files; f++) {
            if (!fasta_files[f] || (fai_files && !fai_files[f])) throw Error(JK_ERR_ARG, "NULL file name");
            HostText text;
            slurp_gz(fasta_files[f], text);
            const FastaPlan plan = fai_files ? plan_ind(fai_files[f], text.size()) : plan_noind(text, cut_names != 0);
            fasta_pack_file(*G, text, plan, /*strip_cr=*/fai_files == nullptr, remove_soft_mask != 0, &G->ms);
        }
        *out = G.release();
    });
}

uint64_t jk_genome_seed_words_used(const jk_genome* g) { return g ? g->seed_words_used : 0; }
double jk_genome_ms(const jk_genome* g) { return g ? g->ms : 0.0; }
void jk_genome_free(jk_genome* g) { delete g; }

// the jump-ahead create_genome relies on, on its own (host): seed, jump `steps` outputs ahead, n outputs
void jk_pcg_advance_outputs(const uint32_t* words8, uint64_t steps, uint64_t n, uint64_t* out) {
    jk_pcg64 e = jk_pcg_seed(words8);
    PcgMap map[64];
    pcg_advance_table(e, map);
    pcg_advance(e, map, steps);
    for (uint64_t i = 0; i < n; i++) out[i] = jk_pcg_next(e);
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
