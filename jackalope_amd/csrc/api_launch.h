// api_launch.h -- one generate(): generator launches, scans and compaction on two streams
// (part of the one translation unit jk_api.hip; see the include list there)
#pragma once

namespace jk {

// Sink side of a streaming run: one thread takes the batches' images in order as their compaction completes and
// hands them to the files (device BGZF where asked for, pinned double-buffered D2H, file writes on writer threads),
// while the calling thread keeps the generator launches coming.  An image slot is reused two batches later, once its
// bytes have left the device.
struct StreamCtx {
    jk_session& s;
    HostPipe& pipe;
    std::vector<std::unique_ptr<FastqFile>> files;
    std::mutex m; std::condition_variable cv;
    std::deque<int> q; bool closed = false;
    int consumed = -1;                   // last batch whose image has left the device
    int err_code = 0; std::string err;
    std::thread th;
    StreamCtx(jk_session& s_, const std::string& suffix) : s(s_), pipe(session_pipe(s_)) {
        for (uint32_t e = 0; e < s.n_ends; e++) files.emplace_back(new FastqFile(s, e, suffix));
        th = std::thread([this] { run(); });
    }
    // (the files close after this body: no writer thread may still hold a task with their descriptors -- an error path
    //  gets here without finish(), and a re-plan reopens the same names right afterwards)
    ~StreamCtx() { close(); pipe.quiesce(); }
    void push(int b) { { std::lock_guard<std::mutex> l(m); q.push_back(b); } cv.notify_all(); }
    void close() {
        { std::lock_guard<std::mutex> l(m); closed = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
    // wait until batch `b`'s image slot may be overwritten; false = the sink failed
    bool wait_consumed(int b) {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return consumed >= b || err_code != 0; });
        return err_code == 0;
    }
    void fail(int code, const std::string& what) {
        { std::lock_guard<std::mutex> l(m); if (!err_code) { err_code = code; err = what; } }
        cv.notify_all();
    }
    void run() {
        try {
            JK_HIP(hipSetDevice(s.device));
            for (;;) {
                int b;
                {
                    std::unique_lock<std::mutex> l(m);
                    cv.wait(l, [&] { return closed || !q.empty() || err_code != 0; });
                    if (err_code != 0 || q.empty()) return;
                    b = q.front(); q.pop_front();
                }
                JK_HIP(hipEventSynchronize(s.cp_done[b]));
                uint32_t kerr2[2] = {0, 0};                  // (the error words of both step slots: a run uses one of them)
                JK_HIP(hipMemcpyAsync(kerr2, s.d_err.p, 8, hipMemcpyDeviceToHost, pipe.stream()));
                uint64_t base[2][2] = {{0, 0}, {0, 0}};
                for (uint32_t e = 0; e < s.n_ends; e++)
                    JK_HIP(hipMemcpyAsync(base[e], s.d_base[e].as<uint64_t>() + b, 16, hipMemcpyDeviceToHost, pipe.stream()));
                JK_HIP(hipStreamSynchronize(pipe.stream()));
                if (kerr2[0] | kerr2[1]) { fail(JK_ERR_DEVICE, "kernel error"); return; }      // (the caller reads the bits and words the message)
                const int slot = b & 1;
                for (uint32_t e = 0; e < s.n_ends; e++)
                    files[e]->add(pipe, s.d_img[slot][e].as<uint8_t>(), base[e][1] - base[e][0]);
                s.progress_done.fetch_add(s.batches[b].n_reads);
                { std::lock_guard<std::mutex> l(m); consumed = b; }
                cv.notify_all();
            }
        } catch (const Error& e) { fail(e.code, e.what()); }
        catch (const std::exception& e) { fail(JK_ERR_IO, e.what()); }
    }
};

// One pass over all batches of the session.  `sc` == nullptr: the images of all batches end up side by side in the
// resident image (d_out).  Otherwise every batch's image goes to one of two per-batch buffers and on to the sink.
static void complete_step(jk_session& s, StreamCtx* sc, int slot, bool stopped, size_t ev);

// `pipelined`: return once the step is queued (jk_session_generate_async); jk_session_wait completes it.  Two steps may
// be in flight: the generator launches of step k + 1 then run beside the last compaction of step k, whose pool set they do
// not use -- the un-overlapped tail of a step (about a tenth of the headline step) is hidden behind the next step's head.
static void launch_batches(jk_session& s, StreamCtx* sc, bool pipelined = false) {
    JK_HIP(hipSetDevice(s.device));
    const int step_slot = s.next_slot;
    // The set-up (open, re-plan) zeroes and fills buffers with hipMemset and small kernels on the null stream, which nothing
    // orders against this session's non-blocking streams except the host having waited: most of that work is followed by
    // a synchronous copy, but the first launch after a set-up does not rely on it.
    if (s.setup_pending) { JK_HIP(hipStreamSynchronize(nullptr)); s.setup_pending = false; }
    uint32_t* const err_ptr = s.d_err.as<uint32_t>() + step_slot;
    JK_HIP(hipMemsetAsync(err_ptr, 0, 4, s.stream));
    if (s.inflight == 0) for (uint32_t e = 0; e < s.n_ends; e++) JK_HIP(hipMemsetAsync(s.d_base[e].p, 0, 8, s.stream));      // (entry 0 stays 0)
    s.progress_done.store(0);
    size_t ev = 0;
    JK_HIP(hipEventRecord(s.events[ev++], s.stream));
    if (s.inflight == 0) JK_HIP(hipStreamWaitEvent(s.cp_stream, s.events[0], 0));
    bool stopped = false;
    for (size_t b = 0; b < s.batches.size(); b++) {
        if (s.abort_flag && *s.abort_flag) { stopped = true; break; }
        const Batch& B = s.batches[b];
        const int pp = (int)(b % (size_t)s.n_pool_sets);       // pool set in rotation
        const size_t ns = (size_t)s.n_pool_sets;
        const int slot = (int)(b & 1);
        // where this batch's compacted image goes
        uint8_t* out_img[2]; const uint64_t* out_base[2]; uint64_t out_cap;
        for (uint32_t e = 0; e < 2; e++) {
            out_img[e] = e < s.n_ends ? (sc ? s.d_img[slot][e].as<uint8_t>() : s.d_out[e].as<uint8_t>()) : nullptr;
            out_base[e] = e < s.n_ends ? (sc ? s.d_zero.as<uint64_t>() : s.d_base[e].as<uint64_t>() + b) : nullptr;
        }
        out_cap = sc ? s.img_cap : s.out_cap;
        if (s.pacbio) {
            // plan kernel on the generator stream, lane scan + emit kernel on the second one: the plan kernel of launch
            // b + 1 runs beside the emit kernel of launch b, each with its own set of records / masks / counters
            const int set = (int)(b & 1);
            PacbioKernelParams Q = s.kpb;
            Q.err = err_ptr;
            Q.n_lanes = B.n_lanes;
            Q.seeds = s.d_seeds.as<uint32_t>() + B.lane0 * 8;
            Q.lane_reads = s.d_lane_reads.as<uint64_t>() + B.lane0;
            Q.chrom_reads = s.d_chrom_reads.as<uint32_t>() + B.lane0;
            Q.chrom_stride = (uint32_t)s.n_shard;
            Q.rec_off = s.d_pb_rec_off.as<uint64_t>() + B.lane0;
            Q.recs = s.d_pb_recs[set].as<PbRead>();
            Q.lane_bytes = s.d_lane_bytes[0].as<uint64_t>() + B.lane0;
            Q.lane_made = s.d_lane_made.as<uint64_t>() + B.lane0;
            Q.masks = s.d_pb_masks[set].as<uint4>();
            Q.mask_ctr = s.d_pb_ctr[set].as<unsigned long long>();
            Q.stale = s.d_pb_stale[set].as<uint8_t>();
            Q.stale_ctr = s.d_pb_ctr[set].as<uint32_t>() + 2;
            // the set is free once its emit kernel is done: that of launch b - 2, or (a step queued behind another) of the
            // last launch of the step before that used the set -- its event still holds that record
            if (b >= 2) JK_HIP(hipStreamWaitEvent(s.stream, s.cp_done[b - 2], 0));
            else if (s.inflight > 0) {
                const size_t nb = s.batches.size();
                for (size_t k = nb; k-- > 0;) if ((int)(k & 1) == set) { JK_HIP(hipStreamWaitEvent(s.stream, s.cp_done[k], 0)); break; }
            }
            // The plan kernel of launch b runs beside the emit kernel of launch b - 1 (8 waves of 64 VGPRs per SIMD leave the
            // emit kernel's waves room); taking turns instead was ahead by 2-4 % while the plan kernel held 4 x 128 VGPRs and
            // its waves carried fewer than 64 lanes, and is behind by up to 8 % now (tools/pb_planwaves_probe.sh: four job
            // shapes).  JK_PB_SERIAL=1 for that schedule.
            bool pb_serial = false;
            if (const char* e = std::getenv("JK_PB_SERIAL")) pb_serial = std::atoi(e) != 0;
            if (pb_serial && b >= 1) JK_HIP(hipStreamWaitEvent(s.stream, s.cp_done[b - 1], 0));
            JK_HIP(hipEventRecord(s.events[ev++], s.stream));
            JK_HIP(hipMemsetAsync(s.d_pb_recs[set].p, 0, std::max<uint64_t>(B.n_reads, 1) * sizeof(PbRead), s.stream));
            JK_HIP(hipMemsetAsync(s.d_pb_ctr[set].p, 0, 16, s.stream));
            const uint32_t wl = s.pb_wave_lanes[b];
            Q.wave_lanes = wl;
            const uint32_t n_waves = (B.n_lanes + wl - 1) / wl;
            const uint32_t pgrid = (n_waves * 64u + PB_PLAN_BLOCK - 1) / PB_PLAN_BLOCK;
            if (s.hap) hipLaunchKernelGGL((pb_plan_kernel<true>), dim3(pgrid), dim3(PB_PLAN_BLOCK), 0, s.stream, Q);
            else hipLaunchKernelGGL((pb_plan_kernel<false>), dim3(pgrid), dim3(PB_PLAN_BLOCK), 0, s.stream, Q);
            JK_HIP(hipGetLastError());
            JK_HIP(hipEventRecord(s.events[ev++], s.stream));
            JK_HIP(hipEventRecord(s.gen_done[b], s.stream));
            if (sc && b >= 2 && !sc->wait_consumed((int)b - 2)) { stopped = true; break; }
            JK_HIP(hipStreamWaitEvent(s.cp_stream, s.gen_done[b], 0));
            const uint32_t nbp = (B.n_lanes + SCAN_BLOCK - 1) / SCAN_BLOCK;
            uint64_t* lb = s.d_lane_bytes[0].as<uint64_t>() + B.lane0;
            uint64_t* lo = s.d_lane_off[0].as<uint64_t>() + B.lane0;
            uint64_t* bs = s.d_block_sums.as<uint64_t>();
            uint64_t* base = s.d_base[0].as<uint64_t>() + b;
            hipLaunchKernelGGL(scan_block_kernel, dim3(nbp), dim3(SCAN_BLOCK), 0, s.cp_stream, lb, lo, bs, B.n_lanes);
            hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_BLOCK), 0, s.cp_stream, bs, nbp, base);
            hipLaunchKernelGGL(scan_add_kernel, dim3(nbp), dim3(SCAN_BLOCK), 0, s.cp_stream, lo, bs, B.n_lanes);
            PbEmitParams E;
            E.g = Q.g; E.h = Q.h; E.n_chroms = s.n_chroms;
            E.recs = Q.recs; E.n_recs = (uint32_t)B.n_reads;
            E.seeds = Q.seeds; E.lane_off = lo;
            E.masks = Q.masks; E.stale = Q.stale; E.jump = Q.jump;
            E.out = out_img[0]; E.out_base = out_base[0]; E.out_cap = out_cap;
            E.err = err_ptr;
            // (JK_PB_FORCE_EXACT=1, tests: every buffer of draws goes through the exact index routine and the general block)
            E.exact_from = (std::getenv("JK_PB_FORCE_EXACT") && std::atoi(std::getenv("JK_PB_FORCE_EXACT")) != 0) ? 0u : 0xfffffff0u;
            if (B.n_reads) {
                if (Q.hap_seg) hipLaunchKernelGGL((pb_emit_kernel<true>), dim3((uint32_t)B.n_reads), dim3(64), 0, s.cp_stream, E);
                else hipLaunchKernelGGL((pb_emit_kernel<false>), dim3((uint32_t)B.n_reads), dim3(64), 0, s.cp_stream, E);
            }
            JK_HIP(hipGetLastError());
            JK_HIP(hipEventRecord(s.cp_done[b], s.cp_stream));
            if (sc) sc->push((int)b);
            continue;
        }
        IlluminaKernelParams P = s.kp;
        P.err = err_ptr;
        P.rare_log = s.d_err.as<uint32_t>() + 2; P.rare_lane0 = (uint32_t)B.lane0;
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
        P.evw = s.d_evw.as<uint64_t>() + (size_t)(b & 1) * s.evw_set;
        P.chrom_stride = (uint32_t)s.n_shard;
        // Two generators may be in flight (each has its own pool set and indel scratch): the next batch's
        // workgroups then take over CUs as the current batch's finish instead of waiting for its slowest one.
        hipStream_t gs = (s.two_gen_streams && (b & 1)) ? s.stream2 : s.stream;
        if (s.two_gen_streams && b == 1) JK_HIP(hipStreamWaitEvent(s.stream2, s.events[0], 0));
        const uint32_t block = JK_ILL_BLOCK;
        const uint32_t grid = (B.n_lanes + block - 1) / block;
        // the pool set is free again once the compaction of batch b-2 has read it -- of this step, or (a step queued
        // behind another) the last batch of the step before that used the set: its event still holds that record
        if (b >= ns) JK_HIP(hipStreamWaitEvent(gs, s.cp_done[b - ns], 0));
        else if (s.inflight > 0) {
            const size_t nb = s.batches.size();
            size_t last = nb;                            // largest b' < nb with b' % ns == pp
            for (size_t k = nb; k-- > 0;) if ((int)(k % ns) == pp) { last = k; break; }
            if (last < nb) JK_HIP(hipStreamWaitEvent(gs, s.cp_done[last], 0));
        }
        JK_HIP(hipEventRecord(s.events[ev++], gs));
#define JK_LAUNCH(LDS, NE, HAP, SEG, SH) hipLaunchKernelGGL((illumina_kernel<LDS, NE, JK_ILL_BLOCK, HAP, SEG>), dim3(grid), dim3(block), SH, gs, P)
        const bool seg = s.hap && !s.hap_materialised;       // bases through the mutation tables (else: plain sequences)
#define JK_LAUNCH6(NE, HAP, SEG, SH) hipLaunchKernelGGL((illumina_kernel<true, NE, JK_ILL_BLOCK, HAP, SEG, true>), dim3(grid), dim3(block), SH, gs, P)
        if (s.lds_tables && s.ent6) {
            if (seg)        { if (s.n_ends == 2) JK_LAUNCH6(2, true, true, s.lds_launch); else JK_LAUNCH6(1, true, true, s.lds_launch); }
            else if (s.hap) { if (s.n_ends == 2) JK_LAUNCH6(2, true, false, s.lds_launch); else JK_LAUNCH6(1, true, false, s.lds_launch); }
            else            { if (s.n_ends == 2) JK_LAUNCH6(2, false, false, s.lds_launch); else JK_LAUNCH6(1, false, false, s.lds_launch); }
        } else if (s.lds_tables) {
            if (seg)        { if (s.n_ends == 2) JK_LAUNCH(true, 2, true, true, s.lds_launch); else JK_LAUNCH(true, 1, true, true, s.lds_launch); }
            else if (s.hap) { if (s.n_ends == 2) JK_LAUNCH(true, 2, true, false, s.lds_launch); else JK_LAUNCH(true, 1, true, false, s.lds_launch); }
            else            { if (s.n_ends == 2) JK_LAUNCH(true, 2, false, false, s.lds_launch); else JK_LAUNCH(true, 1, false, false, s.lds_launch); }
        } else {
            if (seg)        { if (s.n_ends == 2) JK_LAUNCH(false, 2, true, true, s.lds_launch); else JK_LAUNCH(false, 1, true, true, s.lds_launch); }
            else if (s.hap) { if (s.n_ends == 2) JK_LAUNCH(false, 2, true, false, s.lds_launch); else JK_LAUNCH(false, 1, true, false, s.lds_launch); }
            else            { if (s.n_ends == 2) JK_LAUNCH(false, 2, false, false, s.lds_launch); else JK_LAUNCH(false, 1, false, false, s.lds_launch); }
        }
#undef JK_LAUNCH6
#undef JK_LAUNCH
        JK_HIP(hipGetLastError());
        JK_HIP(hipEventRecord(s.events[ev++], gs));
        JK_HIP(hipEventRecord(s.gen_done[b], gs));
        if (sc && b >= 2 && !sc->wait_consumed((int)b - 2)) { stopped = true; break; }
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
            hipLaunchKernelGGL(compact_pools_kernel, dim3((B.n_lanes + 63) / 64), dim3(CP_THREADS), 0, s.cp_stream,
                               s.d_pool[pp][e].as<uint8_t>(), P.pool_off, lb, lo, out_img[e], out_base[e], B.n_lanes);
            JK_HIP(hipGetLastError());
        }
        JK_HIP(hipEventRecord(s.cp_done[b], s.cp_stream));
        if (sc) sc->push((int)b);
    }
    if (stopped) {
        (void)hipStreamSynchronize(s.stream); (void)hipStreamSynchronize(s.cp_stream);
        if (s.stream2) (void)hipStreamSynchronize(s.stream2);
        if (sc) sc->close();
        s.inflight = 0;
        if (s.abort_flag && *s.abort_flag) throw Error(JK_ERR_ABORTED, "aborted");
    }
    // the step's summary, behind its last compaction: {error bits, bytes per end, reads} in the step's result slot
    if (!s.d_result.p) { s.d_result.alloc(64); }
    uint64_t* const res = s.d_result.as<uint64_t>() + 4 * step_slot;
    JK_HIP(hipMemsetAsync(res, 0, 32, s.cp_stream));
    hipLaunchKernelGGL(finish_kernel, dim3(64), dim3(256), 0, s.cp_stream, s.d_lane_made.as<uint64_t>(), (uint64_t)s.n_shard, err_ptr,
                       s.d_base[0].as<uint64_t>() + s.batches.size(),
                       s.n_ends > 1 ? s.d_base[1].as<uint64_t>() + s.batches.size() : (const uint64_t*)nullptr, res);
    JK_HIP(hipGetLastError());
    if (!stopped) JK_HIP(hipEventRecord(s.events[ev++], s.cp_stream));
    JK_HIP(hipEventRecord(s.step_end[step_slot], s.cp_stream));
    s.inflight++;
    s.next_slot ^= 1;
    s.pending_ev[step_slot] = ev;
    if (pipelined && !stopped && !sc) return;
    complete_step(s, sc, step_slot, stopped, ev);
}

// wait for the step in `slot` and take its results
static void complete_step(jk_session& s, StreamCtx* sc, int slot, bool stopped, size_t ev) {
    JK_HIP(hipEventSynchronize(s.step_end[slot]));
    if (s.inflight == 1) { JK_HIP(hipStreamSynchronize(s.stream)); JK_HIP(hipStreamSynchronize(s.cp_stream)); }
    s.inflight--;
    if (sc) sc->close();               // the sink has taken every batch it was given (or failed)

    uint64_t result[4] = {0, 0, 0, 0};
    JK_HIP(hipMemcpy(result, s.d_result.as<uint64_t>() + 4 * slot, sizeof(result), hipMemcpyDeviceToHost));
    const uint32_t err = (uint32_t)result[0];
    if (err & JK_KERR_GAMMA_MATH) throw Error(JK_ERR_UNSUPPORTED, "a fragment-length draw with frag_len_shape < 1 needed pow() beyond the range implemented on the GPU (|log(u) / shape| >= 512)");
    if (err & JK_KERR_EMPTY_CHROM) throw Error(JK_ERR_UNSUPPORTED, "a lane was given reads for an empty chromosome (the reference hands reads_per_group's remainder to the last chromosome whatever its length and writes records without bases)");
    if (err & JK_KERR_PB_MATH) throw Error(JK_ERR_UNSUPPORTED, "a PacBio parameter led to an exp/pow argument outside the range implemented on the GPU");
    if (err & JK_KERR_PB_TOO_LONG) throw Error(JK_ERR_UNSUPPORTED, "a read was longer than 2^30 bases or needed more than twice its length in reference positions (deletion probability too high for the GPU path)");
    if (err & JK_KERR_PB_SPACE) throw Error(JK_ERR_UNSUPPORTED, "a read position lies outside the reference's read buffer (undefined there: a read as long as its chromosome or a clipped duplicate with no earlier, longer read on its thread) or needs more chromosome than there is");
    // PacBio images are sized for the expected read length: a length model whose realised mean is above that (e.g.
    // min_read_length cutting off most of the log-normal) gets a larger image
    if ((err & JK_KERR_IMAGE_FULL) && s.pacbio && s.replan && !std::getenv("JK_PB_NO_IMAGE_RETRY")) throw Error(JK_ERR_RETRY_IMAGE, "image full");
    if (err & JK_KERR_IMAGE_FULL) throw Error(JK_ERR_DEVICE, "the FASTQ image of this run does not fit in device memory next to its pools: use more GPUs (lane shards) or fewer reads per call");
    if ((err & JK_KERR_POOL_OVERFLOW) && s.pacbio) throw Error(JK_ERR_RETRY, "pool overflow");
    if (err & JK_KERR_POOL_OVERFLOW) throw Error(JK_ERR_DEVICE, "internal error: a lane overflowed its pool region");
    if (err & JK_KERR_TOO_MANY_DELETIONS) throw Error(JK_ERR_UNSUPPORTED, "a read needed more than 2x read_length reference positions (deletion probability too high for the GPU path)");
    if (sc && sc->err_code) throw Error(sc->err_code, sc->err);
    if (stopped) throw Error(JK_ERR_DEVICE, "the run stopped early");
    for (uint32_t e = 0; e < s.n_ends; e++) s.bytes[e] = result[1 + e];
    s.reads_made = result[3];
    if (s.inflight == 0) {
        // (the timing events are the session's, not the step's: with another step queued behind this one they already hold
        //  that step's records -- the times of a pipelined run are those of its last step)
        float t = 0;
        double gen = 0, rest = 0;
        {   // time with a generator launch running: the union of the launches' spans
            std::vector<std::pair<float, float>> span(s.batches.size());
            for (size_t b = 0; b < s.batches.size(); b++) {
                JK_HIP(hipEventElapsedTime(&span[b].first, s.events[0], s.events[1 + 2 * b]));
                JK_HIP(hipEventElapsedTime(&span[b].second, s.events[0], s.events[2 + 2 * b]));
            }
            std::sort(span.begin(), span.end());
            float hi = -1.0f;
            for (const auto& sp : span) {
                const float a = std::max(sp.first, hi);
                if (sp.second > a) gen += sp.second - a;
                hi = std::max(hi, sp.second);
            }
        }
        JK_HIP(hipEventElapsedTime(&t, s.events[0], s.events[ev - 1]));
        rest = t - gen;
        s.ms[0] = gen; s.ms[1] = rest; s.ms[2] = t;
    }
    s.progress_done.store(s.progress_total);
}

// generate(): all batches into the resident image
static void drain_steps(jk_session& s) {      // complete whatever is queued (oldest first)
    while (s.inflight > 0) {
        const int slot = s.inflight == 2 ? s.next_slot : (s.next_slot ^ 1);
        complete_step(s, nullptr, slot, false, s.pending_ev[slot]);
    }
}
static void launch_generate(jk_session& s) {
    if (s.streaming) throw Error(JK_ERR_ARG, "this session streams its output (stream_output): use jk_session_run");
    drain_steps(s);
    launch_batches(s, nullptr);
    s.generated = true;
}
// generate_async(): queue one more pass over all batches (at most two in flight); wait(): complete the oldest
static void launch_generate_async(jk_session& s) {
    if (s.streaming) throw Error(JK_ERR_ARG, "this session streams its output (stream_output): use jk_session_run");
    if (s.inflight >= 2) throw Error(JK_ERR_ARG, "two steps are in flight already: jk_session_wait first");
    launch_batches(s, nullptr, true);
}
static void launch_wait(jk_session& s) {
    if (s.inflight == 0) throw Error(JK_ERR_ARG, "no step is in flight");
    const int slot = s.inflight == 2 ? s.next_slot : (s.next_slot ^ 1);
    complete_step(s, nullptr, slot, false, s.pending_ev[slot]);
    s.generated = true;
}

// run(): all batches through the sink into <out_prefix>_R<e>.fq[.gz]<suffix>; `with_eof` = 0 leaves a BGZF file
// open-ended (another part is appended behind it)
static void launch_stream(jk_session& s, const std::string& suffix = "", bool with_eof = true, uint64_t* file_bytes = nullptr) {
    if (!s.streaming) throw Error(JK_ERR_ARG, "jk_session_run needs a session opened with stream_output");
    JK_HIP(hipSetDevice(s.device));          // (the pipe's stream and events, made on first use, belong to this device)
    StreamCtx sc(s, suffix);
    launch_batches(s, &sc);
    for (uint32_t e = 0; e < s.n_ends; e++) {
        if (sc.files[e]->plain_bytes() != s.bytes[e]) throw Error(JK_ERR_DEVICE, "internal error: streamed bytes differ from the generated total");
        sc.files[e]->finish(sc.pipe, with_eof);
        if (file_bytes) file_bytes[e] = sc.files[e]->bytes_written();
    }
    s.streamed = true;
}

}  // namespace jk
