// api_illumina.h -- host set-up of the Illumina sessions: genome upload, error model tables, lanes, pools
// (part of the one translation unit jk_api.hip; see the include list there)
#pragma once

namespace jk {

// Chromosomes (+ optionally the haplotypes' nucleotide blob) into one encoded device buffer.
static void upload_genome(jk_session& s, const jk_ref_genome& g, const char* blob_bytes, uint64_t blob_len) {
    PhaseTimer pt("upload_genome");
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
    PhaseTimer pt("model tables");
    set_compression(s, a.compress, a.comp_method);
    if (!(a.frag_len_shape > 0)) throw Error(JK_ERR_ARG, "frag_len_shape must be > 0");
    if (!(a.frag_len_scale > 0)) throw Error(JK_ERR_ARG, "frag_len_scale must be > 0");
    s.paired = a.paired != 0;
    s.n_ends = s.paired ? 2 : 1;
    s.out_prefix = a.out_prefix ? a.out_prefix : "";
    s.abort_flag = a.abort_flag;
    s.device = a.device;
    if (a.stream_output) s.streaming = true;
    JK_HIP(hipSetDevice(s.device));
    create_generator_stream(s);
    create_compaction_stream(s);
    JK_HIP(hipStreamCreateWithFlags(&s.stream2, hipStreamNonBlocking));
    if (const char* e = std::getenv("JK_TWO_GEN_STREAMS")) s.two_gen_streams = std::atoi(e) != 0;

    s.tables = build_illumina_tables(a);
    const uint32_t L = s.tables.read_length;
    s.ev_words = (2 * L + 63) / 64 + 1;
    if (s.ev_words > (uint32_t)JK_MAX_EVW_LONG) throw Error(JK_ERR_UNSUPPORTED, "read lengths above 992 are not implemented on the GPU path");

    IlluminaKernelParams& P = s.kp;
    P.read_len = L; P.n_ends = s.n_ends; P.paired = s.paired; P.matepair = (s.paired && a.matepair) ? 1 : 0;
    P.ev_words = s.ev_words;
    P.frag_min = a.frag_len_min; P.frag_max = a.frag_len_max;
    P.gp = jk_gamma_make(a.frag_len_shape, a.frag_len_scale);     // gamma_distribution<double>::param_type::_M_initialize (random.tcc:2330-2346)
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

// The per-lane, per-cell read quotas into d_chrom_reads: uploaded when the planner made them on the host, else made
// on the device from the deferred reads_per_group tasks (chrom_split_kernel; tasks it hands back are redone here).
static void upload_quotas(jk_session& s, const LanePlan& lp, const QuotaModel& Q) {
    if (!lp.deferred) { s.d_chrom_reads.upload(lp.quotas.data(), lp.quotas.size()); return; }
    PhaseTimer pt("quotas on the device");
    const uint64_t nh = Q.hap ? Q.n_haps : 1, G = Q.n_chroms, n_cells = nh * G;
    s.d_chrom_reads.alloc(std::max<uint64_t>(n_cells * s.n_shard, 1) * 4);
    JK_HIP(hipMemset(s.d_chrom_reads.p, 0, std::max<uint64_t>(n_cells * s.n_shard, 1) * 4));
    const uint64_t nt = lp.n_tasks();
    if (nt == 0 || G == 0) return;
    std::vector<double> cp(nh * (G - 1) + 1, 0.0), cq(nh * (G - 1) + 1, 0.0);
    std::vector<uint8_t> ck(nh * (G - 1) + 1, 1);
    for (uint64_t h = 0; h < nh; h++) {
        const GroupChain& ch = Q.chrom_chain[h];
        for (uint64_t g = 0; g + 1 < G; g++) {
            cp[h * (G - 1) + g] = ch.p[g]; ck[h * (G - 1) + g] = ch.kind[g];
            const double p12 = ch.p[g] <= 0.5 ? ch.p[g] : 1.0 - ch.p[g];
            cq[h * (G - 1) + g] = -std::log(1 - p12);           // BinomDraw's q, by the host's libm
        }
    }
    DevBuf d_words, d_n, d_lane, d_hap, d_p, d_q, d_k, d_redo;
    d_words.upload(lp.task_words.data(), lp.task_words.size()); d_n.upload(lp.task_n.data(), lp.task_n.size());
    d_lane.upload(lp.task_lane.data(), lp.task_lane.size()); d_hap.upload(lp.task_hap.data(), lp.task_hap.size());
    d_p.upload(cp); d_q.upload(cq); d_k.upload(ck);
    d_redo.alloc(nt * 4);
    JK_HIP(hipMemset(d_redo.p, 0, nt * 4));
    SplitChainDev C{d_p.as<double>(), d_q.as<double>(), d_k.as<uint8_t>(), (uint32_t)G};
    hipLaunchKernelGGL(chrom_split_kernel, dim3((uint32_t)((nt + 255) / 256)), dim3(256), 0, 0, nt, d_words.as<uint32_t>(), d_n.as<uint32_t>(),
                       d_lane.as<uint32_t>(), d_hap.as<uint32_t>(), C, Q.n_ends, (uint64_t)s.n_shard, s.d_chrom_reads.as<uint32_t>(),
                       d_redo.as<uint32_t>());
    JK_HIP(hipGetLastError());
    std::vector<uint32_t> redo(nt);
    JK_HIP(hipMemcpy(redo.data(), d_redo.p, nt * 4, hipMemcpyDeviceToHost));
    std::vector<uint64_t> idx;
    for (uint64_t k = 0; k < nt; k++) if (redo[k]) idx.push_back(k);
    if (idx.empty()) return;
    // binomials outside the waiting-time branch (many reads per lane and chromosome): libstdc++ itself, on host threads
    std::vector<uint32_t> vals(idx.size() * G);
    run_tasks_on_host(Q, lp, idx.data(), idx.size(), vals.data());
    DevBuf d_idx, d_vals;
    d_idx.upload(idx); d_vals.upload(vals);
    hipLaunchKernelGGL(chrom_split_patch_kernel, dim3((uint32_t)((idx.size() * G + 255) / 256)), dim3(256), 0, 0, (uint64_t)idx.size(),
                       d_idx.as<uint64_t>(), d_vals.as<uint32_t>(), d_lane.as<uint32_t>(), d_hap.as<uint32_t>(), (uint32_t)G,
                       (uint64_t)s.n_shard, s.d_chrom_reads.as<uint32_t>());
    JK_HIP(hipGetLastError());
    JK_HIP(hipDeviceSynchronize());
}

// Pools: tiles of 64 lanes (one wave), every lane of a tile gets the capacity of the tile's largest
// quota of maximal records; a batch is a run of whole tiles.  Then all device buffers.
// lane_cap[l] = pool bytes lane l may need.  Plans batches/tiles and allocates everything that does not
// depend on the sequencer model.  Returns the largest number of lanes in a batch.
// `image_hint` (PacBio): expected size of the compacted image; when the pools' total capacity (sized for the longest
// reads) is far above it, the image buffer is allocated for the hint with headroom instead, and the compaction
// refuses to write past it (JK_KERR_IMAGE_FULL).
static uint32_t plan_pools_common(jk_session& s, uint64_t max_batch_bytes, uint64_t lanes_per_batch,
                                  const std::vector<uint64_t>& lane_cap, const LanePlan& lp, const QuotaModel& Q,
                                  uint64_t image_hint = 0, uint64_t first_batch_lanes = 0) {
    const std::vector<uint64_t>& lane_reads = lp.lane_reads;
    const std::vector<uint32_t>& lane_seeds = lp.lane_seeds;
    PhaseTimer pt("pools: plan, alloc, upload");
    uint64_t max_batch_lanes = lanes_per_batch;
    if (const char* e = std::getenv("JK_BATCH_LANES")) { const long long v = std::atoll(e); if (v >= 64) max_batch_lanes = (uint64_t)v / 64 * 64; }
    std::vector<uint64_t> pool_off;
    uint64_t out_cap = 0, max_pool = 0;
    uint32_t max_lanes = 0;
    auto plan = [&](uint64_t max_batch) {
        s.batches.clear(); s.batch_pool_off_index.clear(); pool_off.clear();
        out_cap = 0; max_pool = 0; max_lanes = 0;
        uint64_t l = 0;
        while (l < s.n_shard) {
            Batch b{l, 0, 0, 0};
            s.batch_pool_off_index.push_back(pool_off.size());
            pool_off.push_back(0);
            uint64_t used = 0;
            const uint64_t lane_limit = (l == 0 && first_batch_lanes) ? first_batch_lanes : max_batch_lanes;
            while (l < s.n_shard && b.n_lanes < lane_limit) {
                const uint64_t tl = std::min<uint64_t>(64, s.n_shard - l);
                uint64_t mx = 0;
                for (uint64_t k = 0; k < tl; k++) mx = std::max(mx, lane_cap[l + k]);
                const uint64_t cap = align_up(mx, 4) * 64;
                if (b.n_lanes > 0 && used + cap > max_batch) break;
                used += cap; pool_off.push_back(used); b.n_lanes += (uint32_t)tl;
                for (uint64_t k = 0; k < tl; k++) b.n_reads += lane_reads[l + k];
                l += tl;
            }
            b.pool_bytes = used;
            out_cap += used;
            max_pool = std::max(max_pool, used);
            max_lanes = std::max(max_lanes, b.n_lanes);
            s.batches.push_back(b);
        }
    };
    if (max_batch_bytes) plan(max_batch_bytes);
    else {
        // No cap given: whole launches (max_batch_lanes lanes) if two pool sets of that size fit next to the image with
        // room to spare, else launches of at most 8 GB of pool per read end (a launch below one workgroup per CU is
        // slower: BASELINE configs[2] at 143 pairs per lane runs 447 instead of 373 M pairs/s with whole launches)
        plan(~0ULL);
        if (max_pool > (8ULL << 30)) {
            size_t free_b = 0, total_b = 0;
            dev_mem_info(&free_b, &total_b);
            uint64_t image = image_hint ? std::min<uint64_t>(out_cap, image_hint + image_hint / 8 + (64ULL << 20)) : out_cap;
            if (s.streaming) image = 2 * max_pool;                 // two per-batch images instead of the whole run's
            const uint64_t need = (2 * max_pool + image) * s.n_ends + (16ULL << 30);
            if (need > free_b) plan(8ULL << 30);
        }
    }
    if (image_hint) out_cap = std::min<uint64_t>(out_cap, image_hint + image_hint / 8 + (64ULL << 20));
    s.out_cap = out_cap;
    {   // say what does not fit before hipMalloc does (a tile's pools are 64 x its largest lane: few lanes with many
        // reads each need far more pool than their FASTQ)
        size_t free_b = 0, total_b = 0;
        dev_mem_info(&free_b, &total_b);
        const uint64_t sets = s.batches.size() > 1 ? 2 : 1;
        const uint64_t image = s.streaming ? std::min<uint64_t>(sets, 2) * (max_pool + 64) : out_cap;
        const uint64_t need = (sets * (max_pool + 64 + CP_SLACK) + image + 64) * s.n_ends + s.n_shard * 64;
        if (need > free_b)
            throw Error(JK_ERR_DEVICE, "this run needs " + std::to_string(need >> 20) + " MiB of device memory (" +
                        std::to_string((sets * max_pool * s.n_ends) >> 20) + " MiB of read pools for " + std::to_string(max_lanes) +
                        " lanes per launch, " + std::to_string((image * s.n_ends) >> 20) + " MiB of FASTQ image) and " +
                        std::to_string(free_b >> 20) + " MiB are free: raise n_threads (more, shorter lanes), lower max_batch_bytes, "
                        "or split the job over more GPUs / calls");
    }
    s.d_seeds.upload(lane_seeds);
    s.d_lane_reads.upload(lane_reads);
    upload_quotas(s, lp, Q);
    s.d_pool_off.upload(pool_off);
    for (uint32_t e = 0; e < s.n_ends; e++) {
        s.d_pool[0][e].alloc(max_pool + 64 + CP_SLACK);
        if (s.batches.size() > 1) s.d_pool[1][e].alloc(max_pool + 64 + CP_SLACK);
        if (s.streaming) {
            // a batch's compacted image is at most its pools' capacity (PacBio: far less; the pools hold worst-case reads)
            s.img_cap = max_pool + 64;
            s.d_out[e].release();
            s.d_img[0][e].alloc(s.img_cap);
            if (s.batches.size() > 1) s.d_img[1][e].alloc(s.img_cap); else s.d_img[1][e].release();
        } else {
            s.d_out[e].alloc(out_cap + 64);
        }
        s.d_lane_bytes[e].alloc(s.n_shard * 8);
        s.d_lane_off[e].alloc(s.n_shard * 8);
        s.d_base[e].alloc((s.batches.size() + 1) * 8);
    }
    s.d_lane_made.alloc(s.n_shard * 8);
    s.d_zero.alloc(16);
    JK_HIP(hipMemset(s.d_zero.p, 0, 16));
    s.progress_total = 0;
    for (const Batch& b : s.batches) s.progress_total += b.n_reads;
    // Pool sets in rotation.  The generator fills the whole register file of every SIMD it runs on, so the compaction
    // of batch b never runs beside generator b+1 on a CU: it goes in between launches.  Measured on the headline
    // workload: 1 set 15.1 ms per step, 2 sets 14.7, 3 sets 14.7 -- two is the default (JK_POOL_SETS=1/3 to change).
    s.n_pool_sets = 2;
    for (uint32_t e = 0; e < s.n_ends; e++) s.d_pool[2][e].release();
    int want_sets = s.pacbio ? 3 : 2;      // (PacBio launches overlap: the third set lets launch b + 1 start while the compaction of b - 1 waits for slots)
    if (const char* e = std::getenv("JK_POOL_SETS")) want_sets = std::atoi(e);
    if (want_sets <= 1) s.n_pool_sets = 1;
    if (s.batches.size() > 2 && want_sets >= 3) {
        size_t free_b = 0, total_b = 0;
        dev_mem_info(&free_b, &total_b);
        if (free_b > (max_pool + 64 + CP_SLACK) * s.n_ends + total_b / 16) {
            for (uint32_t e = 0; e < s.n_ends; e++) s.d_pool[2][e].alloc(max_pool + 64 + CP_SLACK);
            s.n_pool_sets = 3;
        }
    }
    s.d_block_sums.alloc((max_lanes / SCAN_BLOCK + 2) * 8);
    s.d_err.alloc(8 + 4 * (1 + JK_RARE_LOG_CAP));          // two steps' error words, then the generator's rare-branch log
    JK_HIP(hipMemset(s.d_err.p, 0, s.d_err.n));
    s.setup_pending = true;
    for (hipEvent_t& e : s.step_end) if (!e) JK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
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

// The 2-bit copy of the (final) sequence buffer and its flags: GenomeDev::packed.  On by default since every source access
// of an N-free read window goes through it (a first version that served only the 8-base gear was 2-5 % slower and fetched
// more): same speed as the byte path at 100 Mbp / 1 Gbp / 3 Gbp and on the haplotype workload (751 / 737 / 716 / 669 against
// 753 / 744 / 717 / 673 M pairs/s, tools/packed_probe.sh) with 37 % less fetched per launch (FETCH_SIZE 1.44 against 2.27 GB,
// profiles/r02_pmc_packed_vs_bytes.txt).  JK_PACKED_REF=0: bytes only.
static void pack_reference(jk_session& s) {
    if (s.d_packed.p) return;
    if (const char* e = std::getenv("JK_PACKED_REF")) if (std::atoi(e) == 0) return;
    PhaseTimer pt("packed reference");
    const uint64_t n = s.d_seq.n;
    const uint64_t n_threads = (n + 15) / 16, blocks = (n_threads + 255) / 256;
    if (blocks > 0x7fffffffULL) return;
    {   // an extra, not a need: a run that has filled the device with pools and image goes without it
        size_t free_b = 0, total_b = 0;
        dev_mem_info(&free_b, &total_b);
        if ((uint64_t)free_b < blocks * 1024 + (2ULL << 30)) return;
    }
    s.d_packed.alloc(blocks * 1024 + 64);
    s.d_nflags.alloc((blocks / 32 + 3) * 4);
    JK_HIP(hipMemset(s.d_nflags.p, 0, s.d_nflags.n));
    // the cells of the buffer: chromosomes, or the materialised (haplotype, chromosome) sequences
    const bool cells = s.hap && s.hap_materialised;
    const uint64_t* len = cells ? s.d_cell_size.as<uint64_t>() : s.d_chrom_len.as<uint64_t>();
    const uint32_t n_cells = (uint32_t)((cells ? s.d_cell_size.n : s.d_chrom_len.n) / 8);
    hipLaunchKernelGGL(pack_reference_kernel, dim3((uint32_t)blocks), dim3(256), 0, 0, s.d_seq.as<uint8_t>(), n,
                       s.d_packed.as<uint32_t>(), s.d_nflags.as<uint32_t>(), s.d_chrom_off.as<uint64_t>(), len, n_cells);
    JK_HIP(hipGetLastError());
    JK_HIP(hipDeviceSynchronize());
}

static void plan_pools_and_alloc(jk_session& s, const jk_illumina_args& a, const LanePlan& lp, const QuotaModel& Q, uint64_t rec_max) {
    const std::vector<uint64_t>& lane_reads = lp.lane_reads;
    // A batch is one generator launch.  Default: 2^18 lanes = one 1024-thread workgroup on each of the
    // 256 CUs, so every launch is a single full wave of workgroups and the pool compaction of batch b
    // (HBM-bound, second stream) runs under the generator of batch b+1 (ALU-bound).
    std::vector<uint64_t> lane_cap(s.n_shard);
    for (uint64_t l = 0; l < s.n_shard; l++) lane_cap[l] = (lane_reads[l] / s.n_ends) * rec_max;
    // a lane's position in its pool is a 32-bit byte offset in the generator and the compaction
    if (!lane_cap.empty() && *std::max_element(lane_cap.begin(), lane_cap.end()) >= (1ULL << 32))
        throw Error(JK_ERR_UNSUPPORTED, "a lane would write 4 GiB or more of FASTQ per read end (" +
                    std::to_string(lane_reads.empty() ? 0 : lane_reads[0] / s.n_ends) + " reads of up to " + std::to_string(rec_max) +
                    " bytes): raise n_threads -- on the GPU n_threads is the number of generator lanes, 2^16..2^20 per device");
    // Launch size.  A generator workgroup (1024 lanes) owns its CU -- 128 VGPRs x 16 waves are the whole register file --
    // so the scan + compaction of the previous launch can only run on CUs the generator does not use.  A run of several
    // launches therefore leaves an eighth of the CUs to it (224 + 32 on an MI355X: the two compactions of a launch,
    // 1.6 ms each on 32 CUs, then end with the 3.2 ms generator launch they run beside, kernel trace in
    // profiles/r02_ktrace_g224.txt; with 24 CUs they do not keep up, with 40 the generator loses more than it gains).
    // A job that fits one launch takes every CU.
    int n_cu = 256;
    JK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, s.device));
    uint64_t reserve = (uint64_t)n_cu / 8;
    if (const char* e = std::getenv("JK_COMPACT_CUS")) { const long v = std::atol(e); if (v >= 0 && v < n_cu) reserve = (uint64_t)v; }
    uint64_t launch_lanes = (uint64_t)n_cu * JK_ILL_BLOCK;
    if (s.n_shard > launch_lanes) launch_lanes = ((uint64_t)n_cu - reserve) * JK_ILL_BLOCK;
    // The first launch of a run of several has no compaction beside it: it takes every CU (and the third pool set below
    // absorbs its larger compaction).  JK_FIRST_LAUNCH_FULL=0: all launches alike.
    uint64_t first_lanes = 0;
    if (s.n_shard > launch_lanes && launch_lanes < (uint64_t)n_cu * JK_ILL_BLOCK) {
        first_lanes = (uint64_t)n_cu * JK_ILL_BLOCK;
        if (const char* e = std::getenv("JK_FIRST_LAUNCH_FULL")) if (std::atoi(e) == 0) first_lanes = 0;
        if (std::getenv("JK_BATCH_LANES")) first_lanes = 0;
    }
    const uint32_t max_lanes = plan_pools_common(s, a.max_batch_bytes, launch_lanes, lane_cap, lp, Q, 0, first_lanes);
    // after the tables: haplotype runs served from the mutation tables add the per-lane segment table (JK_HAP_SEGS segments
    // x 12 bytes x 1024 lanes); every other run the 2 KB expansion table of the packed reference
    const bool seg_run = s.hap && !s.hap_materialised;
    const size_t seg_bytes = seg_run ? (size_t)JK_HAP_SEGS * 12 * JK_ILL_BLOCK : 0;
    const size_t lut_bytes = seg_run ? 0 : 2048;
    // 8 bytes per alias entry when the tables fit LDS that way (or do not fit either way: they then stay in global memory),
    // 6 bytes per entry when that is what lets them in (JK_ALIAS6=0/1 to force)
    const size_t n_info0 = s.tables.info.size(), n_ent0 = s.tables.thresh.size();
    const size_t bytes8 = n_info0 * 8 + n_ent0 * 8, bytes6 = n_info0 * 8 + n_ent0 * 4 + (n_ent0 + 1) / 2 * 4;
    const size_t lds_room = 156 * 1024;
    s.ent6 = bytes8 + seg_bytes + lut_bytes > lds_room && bytes6 + seg_bytes + lut_bytes <= lds_room && s.ev_words <= (uint32_t)JK_MAX_EVW;
    if (const char* e = std::getenv("JK_ALIAS6")) s.ent6 = std::atoi(e) != 0 && bytes6 + seg_bytes + lut_bytes <= lds_room && s.ev_words <= (uint32_t)JK_MAX_EVW;
    const IlluminaPacked packed = s.ent6 ? pack_illumina_tables6(s.tables) : pack_illumina_tables(s.tables);
    s.d_tab.upload(packed.tab);
    s.d_tab_lo.upload(packed.lo);
    s.d_mm2.upload(packed.mm2);
    s.evw_set = (size_t)s.n_ends * 4 * s.ev_words * std::max<uint32_t>(max_lanes, 1);      // u64 words per generator in flight
    s.d_evw.alloc(2 * s.evw_set * 8);

    s.lds_bytes = packed.tab.size() * 4;        // dynamic LDS; the kernel keeps mm2 (2 KB) in static LDS on top
    // (reads above 480 need the 64-bit event masks, which only the kernels with the tables in global memory have)
    s.lds_tables = s.lds_bytes + seg_bytes + lut_bytes <= 156 * 1024 && s.ev_words <= (uint32_t)JK_MAX_EVW;
    // tables in global memory: their {entry offset, entry count} part (8 bytes per end, position and nucleotide) goes to
    // LDS all the same when it fits beside the rest
    const size_t info_bytes = (size_t)s.tables.info.size() * 8;
    const bool info_lds = !s.lds_tables && info_bytes + seg_bytes + lut_bytes + 2304 <= 120 * 1024;
    const size_t front = s.lds_tables ? align_up(s.lds_bytes, 16) : (info_lds ? align_up(info_bytes, 16) : 0);
    s.lds_seg_off = (uint32_t)front;
    s.lds_lut_off = s.lds_seg_off;
    s.lds_launch = front + seg_bytes + lut_bytes;
    s.kp.info_in_lds = info_lds ? 1u : 0u;
    // the per-lane chromosome cache (32 bytes per lane), when there is room beside the tables and the 2.1 KB of static LDS
    s.lds_cell_off = 0xffffffffu;
    if (!seg_run && s.lds_launch + 32 * JK_ILL_BLOCK + 2304 <= 160 * 1024) {
        s.lds_cell_off = (uint32_t)s.lds_launch;
        s.lds_launch += 32 * JK_ILL_BLOCK;
    }
    if (!seg_run) pack_reference(s);

    IlluminaKernelParams& P = s.kp;
    P.g.seq = s.d_seq.as<uint8_t>();
    P.g.chrom_off = s.d_chrom_off.as<uint64_t>();
    P.g.chrom_len = s.d_chrom_len.as<uint64_t>();
    P.g.hdr_blob = s.d_hdr_blob.as<uint8_t>();
    P.g.hdr_off = s.d_hdr_off.as<uint32_t>();
    P.g.n_chroms = s.n_chroms;
    P.g.packed = s.d_packed.p ? s.d_packed.as<uint8_t>() : nullptr;
    P.g.nflags = s.d_nflags.p ? s.d_nflags.as<uint8_t>() : nullptr;
    P.evw = s.d_evw.as<uint64_t>();
    P.err = s.d_err.as<uint32_t>();
    P.tab = s.d_tab.as<uint32_t>(); P.mm2 = s.d_mm2.as<uint64_t>(); P.tab_lo = s.d_tab_lo.as<uint32_t>();
    P.n_info = (uint32_t)s.tables.info.size(); P.n_entries = (uint32_t)s.tables.thresh.size();

    P.lds_seg_off = s.lds_seg_off;
    P.lds_lut_off = s.lds_lut_off;
    P.lds_cell_off = s.lds_cell_off;
    {
        const int lb = (int)s.lds_launch;
        auto allow = [&](const void* k) { JK_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lb)); };
#define JK_K(LDS, NE, HAP, SEG) reinterpret_cast<const void*>(&illumina_kernel<LDS, NE, JK_ILL_BLOCK, HAP, SEG>)
#define JK_K6(NE, HAP, SEG) reinterpret_cast<const void*>(&illumina_kernel<true, NE, JK_ILL_BLOCK, HAP, SEG, true>)
        if (s.lds_tables && s.ent6) {
            allow(JK_K6(1, false, false)); allow(JK_K6(2, false, false));
            allow(JK_K6(1, true, true)); allow(JK_K6(2, true, true));
            allow(JK_K6(1, true, false)); allow(JK_K6(2, true, false));
        } else if (s.lds_tables) {
            allow(JK_K(true, 1, false, false)); allow(JK_K(true, 2, false, false));
            allow(JK_K(true, 1, true, true)); allow(JK_K(true, 2, true, true));
            allow(JK_K(true, 1, true, false)); allow(JK_K(true, 2, true, false));
        } else if (lb) {
            allow(JK_K(false, 1, true, true)); allow(JK_K(false, 2, true, true));
            allow(JK_K(false, 1, true, false)); allow(JK_K(false, 2, true, false));
            allow(JK_K(false, 1, false, false)); allow(JK_K(false, 2, false, false));
        }
#undef JK_K6
#undef JK_K
    }
}

static uint64_t record_max(size_t max_hdr, uint64_t max_chrom, bool paired, uint32_t L) {
    return max_hdr + n_digits(max_chrom) + 2 + (paired ? 2 : 0) + 1 + (uint64_t)L + 3 + L + 1;
}

// ---- illumina_ref_cpp (src/hts_illumina.cpp:589-649): everything the reference does on the calling
// thread before the parallel region, plus device set-up.
static QuotaModel quota_model_ref(const jk_ref_genome& g, uint32_t n_ends) {
    QuotaModel Q;
    Q.n_ends = n_ends; Q.n_chroms = g.n_chroms;
    Q.chrom_chain.emplace_back(std::vector<double>(g.chrom_lens, g.chrom_lens + g.n_chroms));
    return Q;
}
// IlluminaHaplotypes::add_n_reads (src/hts_illumina.h:620-644) / PacBioHaplotypes::add_n_reads (src/hts_pacbio.h:683-700)
static QuotaModel quota_model_hap(const jk_hap_set& hs, const std::vector<double>& hap_probs, uint32_t n_ends, bool maker_halves) {
    const uint64_t nh = hs.n_haps, nc = hs.ref.n_chroms;
    QuotaModel Q;
    Q.hap = true; Q.n_ends = n_ends; Q.maker_halves = maker_halves; Q.n_haps = nh; Q.n_chroms = nc;
    Q.hap_chain = GroupChain(hap_probs);
    for (uint64_t h = 0; h < nh; h++) {
        std::vector<double> cp(nc);
        for (uint64_t c = 0; c < nc; c++) cp[c] = (double)hs.chrom_size[h * nc + c];
        Q.chrom_chain.emplace_back(cp);
    }
    return Q;
}
// chromosome-level splits on the device unless JK_HOST_SPLITS=1
static bool defer_splits() { const char* e = std::getenv("JK_HOST_SPLITS"); return !(e && std::atoi(e) != 0); }
// this session's lanes: planned here, or cut out of a plan made once for all devices of a one-shot call
static LanePlan session_plan(jk_session& s, const QuotaModel& Q, const std::vector<uint64_t>& per_lane, SeedReader& seeds,
                             bool offset_given, uint64_t offset_words, const LanePlan* full) {
    PhaseTimer pt("lane plan (host)");
    LanePlan lp = full ? slice_plan(*full, s.n_lanes_total, s.lane_begin, s.lane_end)
                       : plan_lane_quotas(Q, per_lane, s.lane_begin, s.lane_end, seeds, offset_given, offset_words, defer_splits());
    s.seed_words_used = lp.words_used; s.shard_seed_begin = lp.shard_begin_word; s.shard_seed_end = lp.shard_end_word;
    return lp;
}

static void open_illumina_ref(jk_session& s, const jk_ref_genome& g, const jk_illumina_args& a, SeedReader& seeds,
                              const LanePlan* full = nullptr) {
    setup_model(s, a);
    const uint32_t L = s.tables.read_length;
    const std::string barcode = (a.barcodes && a.n_barcodes > 0 && a.barcodes[0]) ? a.barcodes[0] : "";
    check_barcode(barcode, L);
    upload_genome(s, g, nullptr, 0);
    uint64_t min_chrom = ~0ULL, max_chrom = 0;
    size_t max_hdr = 0;
    const std::string gname = g.name ? g.name : "REF";
    for (uint64_t i = 0; i < g.n_chroms; i++) {
        // (an empty chromosome has probability 0 in reads_per_group, src/hts.h:78: it gets no reads and does not count here)
        if (g.chrom_lens[i]) min_chrom = std::min<uint64_t>(min_chrom, g.chrom_lens[i]);
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
        blob.resize(blob.size() + 16, 0);        // the kernel reads the id prefix in 4-byte pieces, four of them unconditionally
        s.d_hdr_blob.upload(blob);
        s.d_hdr_off.upload(hoff);
    }
    const uint64_t frag_lb = std::min<uint64_t>(a.frag_len_min <= a.frag_len_max ? a.frag_len_min : a.frag_len_max, min_chrom);
    if (max_chrom == 0) throw Error(JK_ERR_ARG, "the genome holds no bases");
    // a fragment shorter than the barcode: the reference's `read_chrom_spaces[r] -= barcode.size()` wraps below zero and
    // `read[i] = barcode[i]` writes past the string (src/hts_illumina.cpp:177-182, :391) -- undefined there, refused here
    if (frag_lb == 0)            // (R's illumina() demands frag_len_min >= 1, R/hts_illumina.R:327-331; the reference would write records without bases)
        throw Error(JK_ERR_UNSUPPORTED, "frag_len_min = 0 (empty fragments) is not implemented on the GPU path");
    if (frag_lb < barcode.size())
        throw Error(JK_ERR_UNSUPPORTED, "fragments can be shorter than the barcode (frag_len_min or a chromosome of " + std::to_string(frag_lb) +
                    " bases against a barcode of " + std::to_string(barcode.size()) + "): undefined in the reference, refused here");

    // ---- lanes, quotas, seeds: same order of seed consumption as src/hts.h:334-353 (mt_seeds, then per lane
    // IlluminaOneGenome::add_n_reads, src/hts_illumina.h:410-418); see jk_plan.h
    std::vector<uint64_t> per_lane = plan_lanes(s, a.n_threads, a.lane_begin, a.lane_end, a.n_reads);
    const QuotaModel Q = quota_model_ref(g, s.n_ends);
    LanePlan lp = session_plan(s, Q, per_lane, seeds, a.seed_offset_given != 0, a.seed_offset_words, full);

    IlluminaKernelParams& P = s.kp;
    P.bc_len = (uint32_t)barcode.size();
    std::memset(P.barcode, 0, sizeof(P.barcode));
    for (size_t k = 0; k < barcode.size(); k++) P.barcode[k] = encode_base(barcode[k]);
    plan_pools_and_alloc(s, a, lp, Q, record_max(max_hdr, max_chrom, s.paired, L));
}

// Mutation tables of a haplotype set -> device form (see HapDev); also uploads the genome + nucleotide blob.
static void upload_hap_tables(jk_session& s, const jk_hap_set& hs, uint64_t& min_chrom, uint64_t& max_chrom,
                              std::vector<uint64_t>& cell_size) {
    PhaseTimer pt("upload_hap_tables (all)");
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
    ZeroArray<HapMut> mut;                   // (filled by host threads, cell by cell: tens of millions of records)
    mut.assign_zero(n_mut);
    const uint64_t* new_pos = hs.new_pos;
    cell_size.assign(hs.chrom_size, hs.chrom_size + n_cells);
    for (uint64_t k = 0; k < n_cells; k++) {
        if (cell_size[k]) min_chrom = std::min(min_chrom, cell_size[k]);      // (an empty cell gets no reads, as an empty chromosome)
        max_chrom = std::max(max_chrom, cell_size[k]);
    }
    parallel_for(n_cells, 1, [&](size_t ka, size_t kb, unsigned) {
      for (uint64_t k = ka; k < kb; k++) {
        const uint64_t ref_len = hs.ref.chrom_lens[k % nc];
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
            HapMut& mu = mut[m];
            mu.new_pos = hs.new_pos[m];
            mu.nuc_len = smod >= 0 ? (uint32_t)(smod + 1) : 0u;
            mu.nuc_off = s.nuc_base + hs.nuc_off[m];
            mu.ref_shift = (int64_t)hs.old_pos[m] - smod - (int64_t)hs.new_pos[m];
            mu.pad = 0;
            // the reference run after this mutation must stay inside the chromosome
            const uint64_t run_end = (m + 1 < cell_off[k + 1]) ? hs.new_pos[m + 1] : cell_size[k];
            const int64_t last_ref = (int64_t)run_end - 1 + mu.ref_shift;
            if (run_end > hs.new_pos[m] + mu.nuc_len && (last_ref < 0 || (uint64_t)last_ref >= ref_len))
                throw Error(JK_ERR_ARG, "mutation table points outside the reference chromosome");
        }
      }
    });
    {   // bucket index for hap_search (HapDev::bucket): counts of mutations below each bucket boundary, + one closing entry
        std::vector<uint64_t> bucket_off(n_cells + 1, 0);
        for (uint64_t k = 0; k < n_cells; k++) bucket_off[k + 1] = bucket_off[k] + (cell_size[k] >> JK_HAP_BUCKET_SHIFT) + 2;
        ZeroArray<uint32_t> bucket;
        bucket.assign_zero(bucket_off[n_cells]);
        parallel_for(n_cells, 1, [&](size_t ka, size_t kb, unsigned) {
            for (uint64_t k = ka; k < kb; k++) {
                const uint64_t nb = bucket_off[k + 1] - bucket_off[k];
                uint32_t* b = bucket.data() + bucket_off[k];
                uint64_t m = cell_off[k];
                for (uint64_t j = 0; j < nb; j++) {
                    const uint64_t bound = j << JK_HAP_BUCKET_SHIFT;
                    while (m < cell_off[k + 1] && new_pos[m] < bound) m++;
                    b[j] = (uint32_t)(m - cell_off[k]);
                }
            }
        });
        s.d_bucket_off.upload(bucket_off);
        s.d_bucket.upload(bucket.data(), bucket.size());
    }
    s.d_cell_off.upload(cell_off);
    s.d_mut.upload(mut.data(), mut.size());
    s.d_cell_size.upload(cell_size);
}

static void set_hap_params(const jk_session& s, HapDev& h, uint32_t n_haps) {
    h.cell_mut_off = s.d_cell_off.as<uint64_t>();
    h.mut = s.d_mut.as<HapMut>();
    h.cell_size = s.d_cell_size.as<uint64_t>();
    h.bucket_off = s.d_bucket_off.as<uint64_t>();
    h.bucket = s.d_bucket.as<uint32_t>();
    h.bc_blob = s.d_bc_blob.as<uint8_t>();
    h.bc_len = s.d_bc_len.as<uint32_t>();
    h.n_haps = n_haps;
}

// Every haplotype chromosome written out once in device memory (materialise_haps_kernel); afterwards the session's
// genome buffer holds the n_haps x n_chroms materialised sequences and the mutation tables are released.
static void materialise_haplotypes(jk_session& s, uint64_t n_cells, const std::vector<uint64_t>& cell_size) {
    PhaseTimer pt("materialise_haplotypes");
    std::vector<uint64_t> out_off(n_cells), tile0(n_cells + 1, 0);
    uint64_t total = 64;
    for (uint64_t k = 0; k < n_cells; k++) {
        out_off[k] = total;
        total = align_up(total + cell_size[k], 64) + 64;
        tile0[k + 1] = tile0[k] + (cell_size[k] + JK_MAT_TILE - 1) / JK_MAT_TILE;
    }
    if (tile0[n_cells] > 0x7fffffffULL) throw Error(JK_ERR_UNSUPPORTED, "haplotype set too large to materialise");
    DevBuf d_hap, d_out_off, d_tile0;
    d_hap.alloc(total);
    JK_HIP(hipMemset(d_hap.p, 'N', total));
    d_out_off.upload(out_off); d_tile0.upload(tile0);
    HapDev H;
    set_hap_params(s, H, (uint32_t)(n_cells / s.n_chroms));
    if (tile0[n_cells])
        hipLaunchKernelGGL(materialise_haps_kernel, dim3((uint32_t)tile0[n_cells]), dim3(256), 0, 0, s.d_seq.as<uint8_t>(),
                           s.d_chrom_off.as<uint64_t>(), s.n_chroms, H, (uint32_t)n_cells, d_tile0.as<uint64_t>(),
                           d_out_off.as<uint64_t>(), d_hap.as<uint8_t>());
    JK_HIP(hipGetLastError());
    JK_HIP(hipDeviceSynchronize());
    s.d_seq.release(); s.d_mut.release(); s.d_bucket.release(); s.d_bucket_off.release(); s.d_cell_off.release();
    s.d_seq.swap(d_hap);
    s.d_chrom_off.upload(out_off);           // indexed by cell from here on
    s.hap_materialised = true;
}

// ---- illumina_hap_cpp (src/hts_illumina.cpp:662-739), one set of output files (sep_files handled by
// the caller: it opens one session per haplotype with one-hot probabilities, src/hts.h:512-552).
static void open_illumina_hap(jk_session& s, const jk_hap_set& hs, const jk_illumina_args& a,
                              const std::vector<double>& hap_probs, uint64_t n_reads, SeedReader& seeds, const LanePlan* full = nullptr) {
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
        blob.resize(blob.size() + 16, 0);
        s.d_hdr_blob.upload(blob);
        s.d_hdr_off.upload(hoff);
    }
    const uint64_t frag_lb = std::min<uint64_t>(a.frag_len_min <= a.frag_len_max ? a.frag_len_min : a.frag_len_max, min_chrom);
    if (max_chrom == 0) throw Error(JK_ERR_ARG, "the haplotypes hold no bases");
    if (frag_lb == 0) throw Error(JK_ERR_UNSUPPORTED, "frag_len_min = 0 (empty fragments) is not implemented on the GPU path");
    if (frag_lb < max_bc)         // (undefined in the reference: see open_illumina_ref)
        throw Error(JK_ERR_UNSUPPORTED, "fragments can be shorter than the barcode (frag_len_min or a chromosome of " + std::to_string(frag_lb) +
                    " bases against a barcode of " + std::to_string(max_bc) + "): undefined in the reference, refused here");

    // ---- lanes, quotas, seeds.  IlluminaHaplotypes::add_n_reads (src/hts_illumina.h:620-644) per lane:
    // reads_per_group over haplotypes, then per haplotype reads_per_group over its chromosomes, then
    // each read maker's own add_n_reads (halves the pair count again when paired; its result is never
    // read by the haplotype path, but it consumes 8 seed words when it has reads).  See jk_plan.h.
    std::vector<uint64_t> per_lane = plan_lanes(s, a.n_threads, a.lane_begin, a.lane_end, n_reads);
    const QuotaModel Q = quota_model_hap(hs, hap_probs, s.n_ends, s.paired);
    LanePlan lp = session_plan(s, Q, per_lane, seeds, a.seed_offset_given != 0, a.seed_offset_words, full);
    const std::vector<uint64_t>& lane_reads = lp.lane_reads;

    {   // Materialise the haplotypes when device memory allows (the kernel then reads plain sequences at the speed of
        // a reference-genome run; through the tables a read end pays a chain of dependent cache misses and the 4-base
        // gear: 500 against 700 M pairs/s on BASELINE configs[3]'s share of one GPU).  Needs sum(cell sizes) bytes next
        // to the run's pools and image; otherwise (hundreds of haplotypes of a large genome) the tables are used.
        uint64_t mat = 64;
        for (uint64_t k = 0; k < n_cells; k++) mat += align_up(cell_size[k], 64) + 64;
        const uint64_t rec = record_max(max_hdr, max_chrom, s.paired, L);
        uint64_t reads_shard = 0, reads_lane_max = 0;
        for (uint64_t v : lane_reads) { reads_shard += v; reads_lane_max = std::max(reads_lane_max, v); }
        const uint64_t launch = std::min<uint64_t>(s.n_shard, 256ULL * JK_ILL_BLOCK);
        const uint64_t pools = 2 * s.n_ends * std::min<uint64_t>(launch * (reads_lane_max / s.n_ends) * rec, a.max_batch_bytes ? a.max_batch_bytes : ~0ULL);
        const uint64_t image = s.streaming ? 0 : (reads_shard / s.n_ends) * rec * s.n_ends;
        size_t free_b = 0, total_b = 0;
        dev_mem_info(&free_b, &total_b);
        bool want = mat + pools + image + (12ULL << 30) <= free_b;
        if (const char* e = std::getenv("JK_HAP_MATERIALISE")) want = std::atoi(e) != 0;
        if (want) materialise_haplotypes(s, n_cells, cell_size);
    }
    IlluminaKernelParams& P = s.kp;
    P.bc_len = 0;
    std::memset(P.barcode, 0, sizeof(P.barcode));
    set_hap_params(s, P.h, (uint32_t)nh);
    plan_pools_and_alloc(s, a, lp, Q, record_max(max_hdr, max_chrom, s.paired, L));
}

}  // namespace jk
