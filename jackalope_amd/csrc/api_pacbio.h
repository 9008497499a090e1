// api_pacbio.h -- host set-up of the PacBio sessions
// (part of the one translation unit jk_api.hip; see the include list there)
#pragma once

namespace jk {

// ---- pacbio_ref_cpp / pacbio_hap_cpp (src/hts_pacbio.cpp:579-715): host set-up -----------------------
// Everything that depends only on the run's parameters or on an integer is tabulated here with the host's
// libm (exactly what the reference calls) and the nmath restatements of jk_nmath.h.
struct PacbioHostModel {
    std::vector<uint64_t> len_thresh; std::vector<uint32_t> len_alias; std::vector<uint64_t> lens;
    std::vector<double> thr_tab; std::vector<PassEntry> pass_tab;
    double min_exp = 0;
    uint64_t len_hi = 0;      // a read length few reads exceed
    uint64_t len_cap = 0;     // the longest read there can be
    double len_mean = 0;      // expected read length (sizes the FASTQ image)
};

static PacbioHostModel setup_pacbio_model(jk_session& s, const jk_pacbio_args& a, uint64_t max_chrom) {
    set_compression(s, a.compress, a.comp_method);
    if (!a.chi2_params_n || !a.chi2_params_s || !a.sqrt_params || !a.norm_params) throw Error(JK_ERR_ARG, "PacBio parameter vectors must not be NULL");
    s.pacbio = true; s.paired = false; s.n_ends = 1;
    // (tests: start with too little per-launch scratch / image so that the run has to be planned again)
    if (const char* e = std::getenv("JK_PB_POOL_SCALE")) { const double v = std::atof(e); if (v > 0) s.pool_scale = v; }
    if (const char* e = std::getenv("JK_PB_IMAGE_SCALE")) { const double v = std::atof(e); if (v > 0) s.image_scale = v; }
    s.out_prefix = a.out_prefix ? a.out_prefix : "";
    s.abort_flag = a.abort_flag;
    s.device = a.device;
    if (a.stream_output) s.streaming = true;
    JK_HIP(hipSetDevice(s.device));
    create_generator_stream(s);
    create_compaction_stream(s);

    PacbioHostModel M;
    PacbioKernelParams& P = s.kpb;
    // read lengths (PacBioReadLenSampler, src/hts_pacbio.h:45-109)
    if (a.n_read_lens == 0) {
        P.use_lognormal = 1;
        P.ln_mu = std::log(a.scale); P.ln_sigma = a.sigma; P.ln_loc = a.loc;
        P.min_read_len = std::ceil(a.min_read_len);
        if (P.min_read_len < 1) P.min_read_len = 1;
        const double hi = std::exp(P.ln_mu + 4.0 * a.sigma) + a.loc;
        M.len_hi = (uint64_t)std::max(hi, P.min_read_len + 1.0);
        M.len_cap = ~0ULL;        // a log-normal read is as long as it is drawn (and at most its chromosome, below)
        M.len_mean = std::max(std::exp(P.ln_mu + 0.5 * a.sigma * a.sigma) + a.loc, P.min_read_len);
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
        {
            double ps = 0, ls = 0;
            for (uint64_t i = 0; i < a.n_read_lens; i++) { ps += a.read_probs[i]; ls += a.read_probs[i] * (double)a.read_lens[i]; }
            M.len_mean = ps > 0 ? ls / ps : (double)M.len_hi;
        }
        if (a.n_read_lens >= (1ULL << 31)) throw Error(JK_ERR_UNSUPPORTED, "too many custom read lengths");
        P.n_lens = (uint32_t)a.n_read_lens;
    }
    M.len_hi = std::min(M.len_hi, max_chrom);
    M.len_cap = std::min(M.len_cap, max_chrom);
    M.len_mean = std::min(M.len_mean, (double)max_chrom);
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

// Launches (batches of whole 64-lane tiles), the image, and the scratch the two kernels hand over: read records, event
// masks, stale characters.  Everything is sized from the EXPECTED read length; a run that outgrows its scratch or its
// image is planned again, larger (s.pool_scale / s.image_scale, with_replan).
static void finish_pacbio(jk_session& s, uint64_t max_batch_bytes, const PacbioHostModel& M, size_t max_hdr, uint64_t max_chrom,
                          const LanePlan& lp, const QuotaModel& Q) {
    PhaseTimer pt("pacbio: plan, alloc, upload");
    const std::vector<uint64_t>& lane_reads = lp.lane_reads;
    const uint64_t rec_mean = max_hdr + n_digits(max_chrom) + 3 + 2 * (uint64_t)std::ceil(M.len_mean) + 8;
    // A launch = whole lanes with all their reads.  The plan kernel is a dependent chain per wave and wants its eight waves
    // on every SIMD (`slots`), no more (a ninth runs after the others: the launch then takes twice as long) and not many
    // fewer; its waves carry wl = 1..64 lanes each.  So a launch takes slots x wl lanes, with wl the largest power of two
    // that keeps its expected FASTQ within the batch size -- and when the run has few lanes with many reads each, a launch
    // still takes `slots` of them if memory allows (up to 96 GB of image per launch: the scratch is an eighth of that),
    // since a lane's reads cannot be spread over launches.
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, s.device);
    uint64_t waves_per_cu = 4 * JK_PB_PLAN_WAVES;
    if (const char* e = std::getenv("JK_PB_WAVES_PER_CU")) { const int v = std::atoi(e); if (v >= 4 && v <= 32 && v % 4 == 0) waves_per_cu = (uint64_t)v; }
    const uint64_t slots = (uint64_t)n_cu * waves_per_cu;
    // (default batch: 32 GB of FASTQ -- a full launch of 10-kb reads at three reads per lane -- when the device has room for
    //  two such image slots and their scratch beside everything else, less on a device that is already in use)
    uint64_t batch_bytes = max_batch_bytes;
    if (!batch_bytes) {
        size_t free0 = 0, total0 = 0;
        dev_mem_info(&free0, &total0);
        batch_bytes = free0 >= (96ULL << 30) ? (32ULL << 30) : free0 >= (48ULL << 30) ? (16ULL << 30) : (8ULL << 30);
    }
    if (const char* e = std::getenv("JK_PB_BATCH_GB")) { const int v = std::atoi(e); if (v >= 1 && v <= 128 && !max_batch_bytes) batch_bytes = (uint64_t)v << 30; }     // (experiments)
    uint64_t max_batch_lanes = 1ULL << 18;
    uint32_t wave_lanes = 64;
    {
        uint64_t reads_shard = 0;
        for (uint64_t v : lane_reads) reads_shard += v;
        const double per_lane = s.n_shard ? (double)reads_shard / (double)s.n_shard * (double)rec_mean : 1.0;   // expected bytes of a lane
        if (!max_batch_bytes) {
            const double want = per_lane * (double)std::min<uint64_t>(s.n_shard, slots);
            const double cap = s.streaming ? 40e9 : 96e9;
            if (want > (double)batch_bytes) batch_bytes = (uint64_t)std::min(want, cap);
        }
        const uint64_t by_bytes = std::max<uint64_t>(64, (uint64_t)((double)batch_bytes / std::max(per_lane, 1.0)));
        while (wave_lanes > 1 && slots * wave_lanes > by_bytes) wave_lanes >>= 1;
        max_batch_lanes = std::max<uint64_t>(64, std::min<uint64_t>(slots * wave_lanes, by_bytes) / 64 * 64);
    }
    if (const char* e = std::getenv("JK_BATCH_LANES")) { const long long v = std::atoll(e); if (v >= 64) max_batch_lanes = (uint64_t)v / 64 * 64; }
    if (const char* e = std::getenv("JK_PB_WAVE_LANES")) { const int v = std::atoi(e); if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) wave_lanes = (uint32_t)v; }
    s.batches.clear(); s.batch_pool_off_index.clear();
    std::vector<uint64_t> rec_off(std::max<uint64_t>(s.n_shard, 1), 0);
    uint64_t max_reads = 0, total_reads = 0;
    uint32_t max_lanes = 0;
    for (uint64_t l = 0; l < s.n_shard;) {
        Batch b{l, 0, 0, 0};
        while (l < s.n_shard && b.n_lanes < max_batch_lanes) {
            const uint64_t tl = std::min<uint64_t>(64, s.n_shard - l);
            uint64_t r = 0;
            for (uint64_t k = 0; k < tl; k++) r += lane_reads[l + k];
            if (b.n_lanes > 0 && (b.n_reads + r) * rec_mean > batch_bytes) break;
            for (uint64_t k = 0, at = b.n_reads; k < tl; k++) { rec_off[l + k] = at; at += lane_reads[l + k]; }
            b.n_reads += r; b.n_lanes += (uint32_t)tl; l += tl;
        }
        if (b.n_reads > 0x7fffffffULL) throw Error(JK_ERR_UNSUPPORTED, "more than 2^31 reads in one launch: lower max_batch_bytes");
        max_reads = std::max(max_reads, b.n_reads); max_lanes = std::max(max_lanes, b.n_lanes);
        total_reads += b.n_reads;
        s.batches.push_back(b);
    }
    // the image: expected bytes + 12.5 % + 64 MB (pb_emit_kernel refuses to write past it: JK_KERR_IMAGE_FULL)
    auto image_for = [&](uint64_t reads) { const uint64_t v = reads * rec_mean; return (uint64_t)((double)(v + v / 8 + (64ULL << 20)) * s.image_scale); };
    s.out_cap = image_for(total_reads);
    s.img_cap = image_for(max_reads);
    // event masks: 16 bytes per 64 positions of a read's walk (about its length), taken from the arena in chunks per wave
    const uint64_t blocks_per_read = (uint64_t)(M.len_mean * 1.15) / 64 + 2;
    // lanes per wave of each launch's plan kernel (a launch with fewer lanes than planned -- the last one -- spreads them thinner)
    uint64_t max_waves = 1;
    s.pb_wave_lanes.clear();
    for (const Batch& b : s.batches) {
        uint32_t wl = wave_lanes;
        while (wl > 1 && (uint64_t)b.n_lanes / wl < slots && !std::getenv("JK_PB_WAVE_LANES")) wl >>= 1;
        while (wl < 64 && ((uint64_t)b.n_lanes + wl - 1) / wl > slots) wl <<= 1;
        s.pb_wave_lanes.push_back(wl);
        max_waves = std::max<uint64_t>(max_waves, (b.n_lanes + wl - 1) / wl);
    }
    // (every wave may leave most of its last chunk unused)
    s.pb_mask_cap = (uint64_t)((double)(max_reads * blocks_per_read + (max_waves + 1) * PB_MASK_CHUNK) * s.pool_scale);
    s.pb_stale_cap = (uint32_t)std::min<double>((double)(1u << 30), (double)(1u << 20) * s.pool_scale);
    const uint64_t sets = s.batches.size() > 1 ? 2 : 1;
    {
        size_t free_b = 0, total_b = 0;
        dev_mem_info(&free_b, &total_b);
        const uint64_t image = s.streaming ? sets * s.img_cap : s.out_cap;
        const uint64_t scratch = sets * (s.pb_mask_cap * 16 + max_reads * sizeof(PbRead) + s.pb_stale_cap);
        // (buffers of an earlier plan of this session are released below before the new ones are made)
        uint64_t held = 0;
        for (int k = 0; k < 2; k++) held += s.d_pb_masks[k].n + s.d_pb_recs[k].n + s.d_pb_stale[k].n + s.d_img[k][0].n;
        held += s.d_out[0].n;
        if (image + scratch + s.n_shard * 64 > (uint64_t)free_b + held)
            throw Error(JK_ERR_DEVICE, "this run needs " + std::to_string((image + scratch) >> 20) + " MiB of device memory (" +
                        std::to_string(scratch >> 20) + " MiB of per-launch scratch for up to " + std::to_string(max_reads) + " reads per launch, " +
                        std::to_string(image >> 20) + " MiB of FASTQ image) and " + std::to_string(((uint64_t)free_b + held) >> 20) +
                        " MiB are free: lower max_batch_bytes, or split the job over more GPUs / calls");
    }
    for (int k = 0; k < 2; k++) { s.d_pb_masks[k].release(); s.d_pb_recs[k].release(); s.d_pb_stale[k].release(); s.d_img[k][0].release(); }
    s.d_out[0].release();
    s.d_seeds.upload(lp.lane_seeds);
    s.d_lane_reads.upload(lane_reads);
    upload_quotas(s, lp, Q);
    s.d_pb_rec_off.upload(rec_off);
    for (uint64_t k = 0; k < sets; k++) {
        s.d_pb_masks[k].alloc(s.pb_mask_cap * 16);
        s.d_pb_recs[k].alloc(std::max<uint64_t>(max_reads, 1) * sizeof(PbRead));
        s.d_pb_stale[k].alloc(s.pb_stale_cap);
        s.d_pb_ctr[k].alloc(16);
    }
    if (s.streaming) { for (uint64_t k = 0; k < sets; k++) s.d_img[k][0].alloc(s.img_cap); }
    else s.d_out[0].alloc(s.out_cap + 64);
    s.d_lane_bytes[0].alloc(std::max<uint64_t>(s.n_shard, 1) * 8);
    s.d_lane_off[0].alloc(std::max<uint64_t>(s.n_shard, 1) * 8);
    s.d_base[0].alloc((s.batches.size() + 1) * 8);
    s.d_lane_made.alloc(std::max<uint64_t>(s.n_shard, 1) * 8);
    JK_HIP(hipMemset(s.d_lane_made.p, 0, std::max<uint64_t>(s.n_shard, 1) * 8));
    s.d_zero.alloc(16);
    JK_HIP(hipMemset(s.d_zero.p, 0, 16));
    s.progress_total = total_reads;
    s.n_pool_sets = 2;
    s.two_gen_streams = false;
    s.d_block_sums.alloc((max_lanes / SCAN_BLOCK + 2) * 8);
    s.d_err.alloc(8);
    JK_HIP(hipMemset(s.d_err.p, 0, 8));
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

    s.d_pb_hist.alloc((size_t)2 * PB_HIST * std::max<uint32_t>(max_lanes, 1) * 8);
    {   // jump constants of the engine: the state j + 1 steps ahead is A s + G inc (engine::advance, pcg_random.hpp:419-434)
        std::vector<uint64_t> jump(64 * 4);
        jk_u128 a = 1, g = 0;
        for (int j = 0; j < 64; j++) {
            g += a; a *= PB_M;                      // a = M^(j+1), g = 1 + M + ... + M^j
            jump[4 * j] = (uint64_t)a; jump[4 * j + 1] = (uint64_t)(a >> 64);
            jump[4 * j + 2] = (uint64_t)g; jump[4 * j + 3] = (uint64_t)(g >> 64);
        }
        s.d_pb_jump.upload(jump);
    }
    s.d_len_thresh.upload(M.len_thresh); s.d_len_alias.upload(M.len_alias); s.d_lens.upload(M.lens);
    s.d_thr_tab.upload(M.thr_tab); s.d_pass_tab.upload(M.pass_tab);
    PacbioKernelParams& P = s.kpb;
    P.g.seq = s.d_seq.as<uint8_t>();
    P.g.chrom_off = s.d_chrom_off.as<uint64_t>();
    P.g.chrom_len = s.d_chrom_len.as<uint64_t>();
    P.g.hdr_blob = s.d_hdr_blob.as<uint8_t>();
    P.g.hdr_off = s.d_hdr_off.as<uint32_t>();
    P.g.n_chroms = s.n_chroms;
    P.g.packed = nullptr; P.g.nflags = nullptr;
    P.hap_seg = (s.hap && !s.hap_materialised) ? 1u : 0u;
    P.hist = s.d_pb_hist.as<uint64_t>();
    P.jump = s.d_pb_jump.as<uint64_t>();
    P.mask_cap = s.pb_mask_cap; P.stale_cap = s.pb_stale_cap;
    // Outside the reference's `read` string its behaviour is undefined; by default such a read ends the run
    // (JK_ERR_UNSUPPORTED).  Opt-in: treat the byte as NUL, which is what freshly allocated string capacity holds.
    if (const char* e = std::getenv("JK_PB_UNDEFINED_AS_NUL")) P.undefined_as_nul = std::atoi(e) != 0;
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

static void open_pacbio_ref(jk_session& s, const jk_ref_genome& g, const jk_pacbio_args& a, SeedReader& seeds, const LanePlan* full = nullptr) {
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
    const QuotaModel Q = quota_model_ref(g, 1);
    LanePlan lp = session_plan(s, Q, per_lane, seeds, a.seed_offset_given != 0, a.seed_offset_words, full);
    std::shared_ptr<LanePlan> plan = std::make_shared<LanePlan>(std::move(lp));
    const uint64_t mbb = a.max_batch_bytes;
    jk_session* sp = &s;
    s.replan = [=]() { finish_pacbio(*sp, mbb, M, max_hdr, max_chrom, *plan, Q); };
    s.replan();
}

static void open_pacbio_hap(jk_session& s, const jk_hap_set& hs, const jk_pacbio_args& a,
                            const std::vector<double>& hap_probs, uint64_t n_reads, SeedReader& seeds, const LanePlan* full = nullptr) {
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
    const QuotaModel Q = quota_model_hap(hs, hap_probs, 1, false);
    LanePlan lp = session_plan(s, Q, per_lane, seeds, a.seed_offset_given != 0, a.seed_offset_words, full);
    std::shared_ptr<LanePlan> plan = std::make_shared<LanePlan>(std::move(lp));
    {   // Materialise the haplotypes when device memory allows (PacBioHaplotypes::one_read does it per thread and cell,
        // src/hts_pacbio.cpp:523): pb_emit_kernel then reads plain sequences, one coalesced byte per position; through
        // the tables every position is a search of its own.  Needs sum(cell sizes) bytes next to the image and scratch.
        uint64_t mat = 64, reads_shard = 0;
        for (uint64_t k = 0; k < n_cells; k++) mat += align_up(cell_size[k], 64) + 64;
        for (uint64_t v : plan->lane_reads) reads_shard += v;
        const uint64_t rec = max_hdr + 24 + 2 * (uint64_t)std::ceil(M.len_mean);
        const uint64_t image = std::min<uint64_t>(reads_shard * rec, s.streaming ? (40ULL << 30) : ~0ULL);
        size_t free_b = 0, total_b = 0;
        dev_mem_info(&free_b, &total_b);
        bool want = mat + image + image / 4 + (12ULL << 30) <= free_b;
        if (const char* e = std::getenv("JK_HAP_MATERIALISE")) want = std::atoi(e) != 0;
        if (want) materialise_haplotypes(s, n_cells, cell_size);
    }
    set_hap_params(s, s.kpb.h, (uint32_t)nh);
    const uint64_t mbb = a.max_batch_bytes;
    jk_session* sp = &s;
    s.replan = [=]() { finish_pacbio(*sp, mbb, M, max_hdr, max_chrom, *plan, Q); };
    s.replan();
}

}  // namespace jk
