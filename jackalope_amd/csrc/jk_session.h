// jk_session.h -- state of one sequencing run (what jk_*_open returns)
// (part of the one translation unit jk_api.hip; see the include list there)
#pragma once
#include <memory>
#include <atomic>


#ifndef JK_ILL_BLOCK
#define JK_ILL_BLOCK 1024     // threads per generator workgroup (one workgroup per CU when the tables sit in LDS)
#endif

using namespace jk;

namespace jk { class HostPipe; }

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
    std::shared_ptr<jk::HostPipe> host_pipe;     // pinned ring + writer threads of the sinks, made on first use and kept (api_sinks.h)
    DevBuf d_packed, d_nflags;        // Illumina: 2-bit copy of d_seq + per-64-base "not only TCAG" flags (GenomeDev::packed)
    uint32_t n_chroms = 0;
    // tables
    IlluminaTables tables;
    DevBuf d_tab, d_tab_lo, d_mm2;
    bool lds_tables = false;
    bool ent6 = false;                     // ... as 6-byte entries (two arrays): kernel flag E6
    size_t lds_bytes = 0, lds_launch = 0, evw_set = 0;
    uint32_t lds_seg_off = 0, lds_lut_off = 0, lds_cell_off = 0xffffffffu;
    bool hap = false;
    bool hap_materialised = false;   // haplotype chromosomes written out in d_seq (no table lookups in the kernel)
    int compress = 0;          // 0 = plain FASTQ, 1..9 = compression level
    bool bgzip = true;         // comp_method: "bgzip" (BGZF blocks) or "gzip"
    bool host_deflate = false; // comp_method "bgzip-host": BGZF blocks deflated by zlib on the host at level `compress`
    bool pacbio = false;
    bool streaming = false;    // batches' images go to the sink as they complete (one-shot entry points); no resident image
    PacbioKernelParams kpb{};
    DevBuf d_len_thresh, d_len_alias, d_lens, d_thr_tab, d_pass_tab, d_pb_hist, d_pb_jump, d_pb_rec_off;
    // PacBio, per launch in flight (pb_plan_kernel of launch b + 1 runs beside pb_emit_kernel of launch b): read records,
    // event-mask arena, stale characters, {mask counter, stale counter}
    DevBuf d_pb_recs[2], d_pb_masks[2], d_pb_stale[2], d_pb_ctr[2];
    uint64_t pb_mask_cap = 0; uint32_t pb_stale_cap = 0;
    std::vector<uint32_t> pb_wave_lanes;          // per launch: lanes per wave of the plan kernel
    uint32_t ev_words = 0;
    uint64_t nuc_base = 0;     // offset of the haplotypes' nucleotide blob inside d_seq
    double pool_scale = 1.25;  // PacBio: scratch capacity (event masks, stale characters) relative to the expected need (grown on overflow)
    double image_scale = 1.0;  // PacBio: image capacity relative to the expected bytes (grown when pb_emit_kernel ran out of image)
    uint32_t retries = 0;      // re-plans of the last generate() (pool or image too small)
    std::function<void()> replan;   // PacBio: re-plan pools after pool_scale changed
    DevBuf d_bucket_off, d_bucket, d_cell_off, d_mut, d_cell_size, d_bc_blob, d_bc_len;
    // lanes of this shard
    uint64_t n_lanes_total = 0, lane_begin = 0, lane_end = 0, n_shard = 0;
    std::vector<uint64_t> pool_off_host;          // per batch-relative offsets, concatenated per batch (n+1 each)
    DevBuf d_seeds, d_lane_reads, d_chrom_reads, d_pool_off;
    std::vector<Batch> batches;
    std::vector<uint64_t> batch_pool_off_index;   // index into d_pool_off of each batch's first entry
    int n_pool_sets = 2;              // pool sets in rotation (3 when memory allows: see plan_pools_common)
    DevBuf d_img[2][2] /* streaming: [slot][end] one batch's compacted image */, d_zero;
    uint64_t img_cap = 0;
    std::atomic<uint64_t> progress_done{0};       // reads whose FASTQ has left the device (streaming) / been generated
    uint64_t progress_total = 0;
    bool streamed = false;                        // a streaming run has completed
    DevBuf d_pool[3][2] /* [set][end] */, d_out[2], d_lane_bytes[2], d_lane_off[2], d_block_sums, d_base[2], d_lane_made, d_evw, d_err, d_result;
    uint64_t out_cap = 0;
    IlluminaKernelParams kp{};                    // template, per-batch fields filled at launch
    // results of the last generate()
    uint64_t bytes[2] = {0, 0};
    uint64_t reads_made = 0;
    double ms[3] = {0, 0, 0};
    bool generated = false;
    uint64_t seed_words_used = 0, shard_seed_begin = 0, shard_seed_end = 0;
    const volatile int32_t* abort_flag = nullptr;
    std::vector<hipEvent_t> events;       // [0] start, [1+2b] / [2+2b] around generator b, last = end
    std::vector<hipEvent_t> gen_done, cp_done;   // per batch, for the two-stream hand-off
    // steps (passes over all batches) in flight: at most two (jk_session_generate_async / jk_session_wait)
    int inflight = 0, next_slot = 0;
    bool setup_pending = true;             // set-up work (memsets, small kernels) was queued on the null stream since the last launch
    hipEvent_t step_end[2] = {nullptr, nullptr};
    size_t pending_ev[2] = {0, 0};

    ~jk_session() {
        // (the buffers below go to the device arena, not to hipFree, which would wait for the device: nothing of this session
        //  may still be running when the next session is handed them)
        if (stream) (void)hipStreamSynchronize(stream);
        if (cp_stream) (void)hipStreamSynchronize(cp_stream);
        if (stream2) (void)hipStreamSynchronize(stream2);
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        for (hipEvent_t e : gen_done) (void)hipEventDestroy(e);
        for (hipEvent_t e : cp_done) (void)hipEventDestroy(e);
        for (hipEvent_t e : step_end) if (e) (void)hipEventDestroy(e);
        if (stream) (void)hipStreamDestroy(stream);
        if (cp_stream) (void)hipStreamDestroy(cp_stream);
        if (stream2) (void)hipStreamDestroy(stream2);
    }
};


namespace jk {
// The generator stream gets the highest stream priority: besides the obvious (the compaction of the previous
// batch should not delay the generator), HIP keeps streams of different priorities on different hardware queues.
// With equal priorities the runtime deals streams round-robin onto GPU_MAX_HW_QUEUES (4) queues, and in a
// process that also runs RCCL (torch.distributed) the generator and compaction streams were seen to land on
// one queue, which serialises them (21.8 -> 24.0 ms per step).
static void create_generator_stream(jk_session& s) {
    int least = 0, greatest = 0;
    JK_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    JK_HIP(hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, greatest));
}
// The scan + compaction stream at the lowest priority: when a generator launch and the previous launch's compaction
// become ready together, the generator's workgroups should be placed first (see also pb_delay_kernel).
static void create_compaction_stream(jk_session& s) {
    int least = 0, greatest = 0;
    JK_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    JK_HIP(hipStreamCreateWithPriority(&s.cp_stream, hipStreamNonBlocking, least));
}
}  // namespace jk
