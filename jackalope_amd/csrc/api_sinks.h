// api_sinks.h -- output sinks: plain, gzip, BGZF on the host or on the device; pipelined copy-out
// (part of the one translation unit jk_api.hip; see the include list there)
#pragma once

#include <condition_variable>
#include <deque>
#include <mutex>

namespace jk {

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
    static std::mutex mu;
    static std::vector<std::unique_ptr<BgzfDeviceTables>> per_device(64);
    if (device < 0 || device >= 64) throw Error(JK_ERR_ARG, "bad device ordinal");
    std::lock_guard<std::mutex> lock(mu);
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

struct EventPair {       // a timing pair that cannot leak when a HIP call between create and destroy throws
    hipEvent_t a = nullptr, b = nullptr;
    EventPair() {
        JK_HIP(hipEventCreate(&a));
        if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); throw Error(JK_ERR_DEVICE, "hipEventCreate failed"); }
    }
    EventPair(const EventPair&) = delete;
    EventPair& operator=(const EventPair&) = delete;
    ~EventPair() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
    double ms() const { float t = 0; JK_HIP(hipEventElapsedTime(&t, a, b)); return t; }
};

// d_src[0..n) -> BGZF blocks (+ the end-of-file block when `with_eof`) at d_dst; returns the size.
// Works through the input in groups of blocks so that the slot scratch stays at 512 MiB.  A streamed file is
// made of several such pieces (one per generator launch, the last block of each shorter than 0xff00), closed
// by one end-of-file block.
// Scratch of bgzf_deflate_device.  A sink that compresses piece after piece keeps one (hipMalloc / hipFree synchronise the
// whole device and the driver clears fresh VRAM: per piece they would stall the generator the sink runs beside).
struct BgzfScratch {
    DevBuf slots, sizes, offs, sums, base;
    void reserve(uint64_t g_blocks, uint64_t n_groups) {
        if (slots.n < g_blocks * BGZF_SLOT) slots.alloc(g_blocks * BGZF_SLOT);
        if (sizes.n < g_blocks * 8) sizes.alloc(g_blocks * 8);
        if (offs.n < g_blocks * 8) offs.alloc(g_blocks * 8);
        const uint64_t sb = ((g_blocks + SCAN_BLOCK - 1) / SCAN_BLOCK) * 8;
        if (sums.n < sb) sums.alloc(sb);
        if (base.n < (n_groups + 1) * 8) base.alloc((n_groups + 1) * 8);
    }
};
static uint64_t bgzf_deflate_device(int device, hipStream_t stream, const uint8_t* d_src, uint64_t n, uint8_t* d_dst,
                                    uint64_t cap, double* ms, bool with_eof = true, BgzfScratch* keep = nullptr) {
    if (reinterpret_cast<uintptr_t>(d_src) & 15u) throw Error(JK_ERR_ARG, "BGZF input must be 16-byte aligned");
    if (cap < bgzf_bound(n)) throw Error(JK_ERR_ARG, "BGZF destination smaller than jk_bgzf_bound()");
    const BgzfTables T = bgzf_tables(device);
    const uint64_t n_blocks = (n + BGZF_BLOCK_IN - 1) / BGZF_BLOCK_IN;
    const uint64_t GROUP = 8192;
    const uint64_t n_groups = (n_blocks + GROUP - 1) / GROUP;
    BgzfScratch own;
    BgzfScratch& sc = keep ? *keep : own;
    const uint64_t g_blocks = std::min<uint64_t>(GROUP, std::max<uint64_t>(n_blocks, 1));
    sc.reserve(g_blocks, n_groups);
    DevBuf &slots = sc.slots, &sizes = sc.sizes, &offs = sc.offs, &sums = sc.sums, &base = sc.base;
    JK_HIP(hipMemsetAsync(base.p, 0, (n_groups + 1) * 8, stream));
    EventPair ev;
    JK_HIP(hipEventRecord(ev.a, stream));
    for (uint64_t g = 0; g < n_groups; g++) {
        const uint64_t b0 = g * GROUP;
        const uint32_t nb = (uint32_t)std::min<uint64_t>(GROUP, n_blocks - b0);
        const uint64_t off = b0 * BGZF_BLOCK_IN;
        // matches (same column of the previous record, runs) unless JK_BGZF_LZ=0: literals only
        static const bool lz = !(std::getenv("JK_BGZF_LZ") && std::atoi(std::getenv("JK_BGZF_LZ")) == 0);
        if (lz) hipLaunchKernelGGL(bgzf_deflate_lz_kernel, dim3(nb), dim3(BGZF_THREADS), 0, stream, d_src + off, n - off,
                                   slots.as<uint8_t>(), sizes.as<uint64_t>(), T);
        else hipLaunchKernelGGL(bgzf_deflate_kernel, dim3(nb), dim3(BGZF_THREADS), 0, stream, d_src + off, n - off,
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
    JK_HIP(hipEventRecord(ev.b, stream));
    uint64_t total = 0;
    JK_HIP(hipMemcpyAsync(&total, base.as<uint64_t>() + n_groups, 8, hipMemcpyDeviceToHost, stream));
    JK_HIP(hipStreamSynchronize(stream));
    if (with_eof) JK_HIP(hipMemcpy(d_dst + total, kBgzfEof, sizeof(kBgzfEof), hipMemcpyHostToDevice));
    if (ms) *ms = ev.ms();
    return total + (with_eof ? sizeof(kBgzfEof) : 0);
}

// ---- device bytes -> host consumers -------------------------------------------------------------------
// A ring of pinned buffers.  copy() cuts a device range into pieces, keeps two D2H copies in flight and hands every
// finished piece to `consume`: on the calling thread, in order (a gzip stream), or on a small pool of writer threads
// (pwrite at known offsets: the file write of one piece overlaps the copies and writes of the others).
class HostPipe {
public:
    using Consume = std::function<void(const uint8_t*, size_t, uint64_t)>;    // (bytes, count, offset of the piece in the range)
    HostPipe(size_t piece_bytes, int n_buffers, int n_workers) : piece_(piece_bytes), nb_(n_buffers) {
        buf_.assign(nb_, nullptr); ev_.assign(nb_, nullptr); busy_.assign(nb_, 0);
        try {
            JK_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
            for (int k = 0; k < nb_; k++) {
                JK_HIP(hipHostMalloc(&buf_[k], piece_, hipHostMallocDefault));
                JK_HIP(hipEventCreateWithFlags(&ev_[k], hipEventDisableTiming));
            }
        } catch (...) { free_all(); throw; }
        for (int w = 0; w < n_workers; w++) workers_.emplace_back([this] { work(); });
    }
    HostPipe(const HostPipe&) = delete;
    HostPipe& operator=(const HostPipe&) = delete;
    ~HostPipe() {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all();
        for (std::thread& t : workers_) t.join();
        free_all();
    }
    size_t piece() const { return piece_; }
    hipStream_t stream() const { return st_; }
    // `ordered`: consume on this thread, piece after piece; else on the workers (consume must be thread-safe)
    void copy(const uint8_t* d_src, uint64_t n, const Consume& consume, bool ordered) {
        struct Fl { int k; size_t n; uint64_t off; };
        std::deque<Fl> flight;
        auto land = [&]() {
            const Fl f = flight.front(); flight.pop_front();
            // (the buffer goes back to the ring on every path out of here, or to a worker together with its task)
            struct Release { HostPipe* p; int k; bool armed; ~Release() { if (armed) p->release(k); } } rel{this, f.k, true};
            JK_HIP(hipEventSynchronize(ev_[f.k]));
            if (ordered || workers_.empty()) {
                consume(static_cast<const uint8_t*>(buf_[f.k]), f.n, f.off);
            } else {
                { std::lock_guard<std::mutex> l(m_); tasks_.push_back(Task{f.k, f.n, f.off, consume}); rel.armed = false; }
                cv_.notify_all();
            }
        };
        try {
            for (uint64_t off = 0; off < n; off += piece_) {
                const size_t cnt = (size_t)std::min<uint64_t>(piece_, n - off);
                const int k = acquire();
                flight.push_back(Fl{k, cnt, off});
                JK_HIP(hipMemcpyAsync(buf_[k], d_src + off, cnt, hipMemcpyDeviceToHost, st_));
                JK_HIP(hipEventRecord(ev_[k], st_));
                if (flight.size() >= 2) land();
            }
            while (!flight.empty()) land();
        } catch (...) {
            (void)hipStreamSynchronize(st_);
            for (const Fl& f : flight) release(f.k);
            throw;
        }
    }
    // wait until the workers have consumed everything handed to them; rethrows their first error
    void drain() {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return tasks_.empty() && running_ == 0; });
        if (!err_.empty()) { const std::string e = err_; err_.clear(); throw Error(JK_ERR_IO, e); }
    }
    // the same without the error: for the exit paths that are already reporting one (files must not be closed, nor their
    // names reopened, while a writer thread still holds a task with the old descriptor)
    void quiesce() noexcept {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return tasks_.empty() && running_ == 0; });
        err_.clear();
    }
private:
    struct Task { int k; size_t n; uint64_t off; Consume fn; };
    void free_all() {
        for (int k = 0; k < nb_; k++) { if (buf_[k]) (void)hipHostFree(buf_[k]); if (ev_[k]) (void)hipEventDestroy(ev_[k]); }
        if (st_) (void)hipStreamDestroy(st_);
    }
    int acquire() {
        std::unique_lock<std::mutex> l(m_);
        for (;;) {
            if (!err_.empty()) { const std::string e = err_; err_.clear(); throw Error(JK_ERR_IO, e); }     // (reported once: the pipe outlives the run)
            for (int k = 0; k < nb_; k++) if (!busy_[k]) { busy_[k] = 1; return k; }
            cv_.wait(l);
        }
    }
    void release(int k) { { std::lock_guard<std::mutex> l(m_); busy_[k] = 0; } cv_.notify_all(); }
    void work() {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return stop_ || !tasks_.empty(); });
                if (tasks_.empty()) return;
                t = tasks_.front(); tasks_.pop_front(); running_++;
            }
            std::string e;
            try { t.fn(static_cast<const uint8_t*>(buf_[t.k]), t.n, t.off); } catch (const std::exception& ex) { e = ex.what(); }
            { std::lock_guard<std::mutex> l(m_); busy_[t.k] = 0; running_--; if (!e.empty() && err_.empty()) err_ = e; }
            cv_.notify_all();
        }
    }
    size_t piece_; int nb_;
    hipStream_t st_ = nullptr;
    std::vector<void*> buf_; std::vector<hipEvent_t> ev_; std::vector<char> busy_;
    std::mutex m_; std::condition_variable cv_;
    std::deque<Task> tasks_; int running_ = 0; bool stop_ = false; std::string err_;
    std::vector<std::thread> workers_;
};

// One output file of a run (one read end): takes the FASTQ image piece by piece -- the whole resident image of a
// session, or one generator launch after the other of a streaming run -- and writes <prefix>_R<e>.fq[.gz].
//   plain        every piece at its byte offset (pwrite on the pipe's writer threads)
//   bgzip        BGZF blocks made on the device; only compressed bytes cross the host link
//   bgzip-host   BGZF blocks deflated by zlib on host threads at the requested level
//   gzip         one gzip stream (gzwrite), in order
// A session with out_prefix == "" is a null sink: the bytes are brought to the host (compressed first when asked for)
// and dropped (bench.py times the D2H-inclusive rate with it).
class FastqFile {
public:
    FastqFile(const jk_session& s, uint32_t end, const std::string& suffix = "", bool truncate = true, uint64_t base_offset = 0)
        : s_(s), at_(base_offset) {
        null_ = s.out_prefix.empty();
        fn_ = s.out_prefix + "_R" + std::to_string(end + 1) + ".fq" + (s.compress > 0 ? ".gz" : "") + suffix;
        if (null_) return;
        if (s.compress > 0 && !s.bgzip) {
            const std::string mode = "wb" + std::to_string(s.compress);
            gz_ = gzopen(fn_.c_str(), mode.c_str());
            if (!gz_) throw Error(JK_ERR_IO, "gzopen of " + fn_ + " failed.\n");
        } else {
            fd_ = ::open(fn_.c_str(), O_WRONLY | O_CREAT | (truncate ? O_TRUNC : 0), 0644);
            if (fd_ < 0) throw Error(JK_ERR_IO, "Unable to open file " + fn_ + ".\n");
        }
    }
    FastqFile(const FastqFile&) = delete;
    FastqFile& operator=(const FastqFile&) = delete;
    ~FastqFile() {
        // (no thread may hold a task with this file's descriptor when it closes: the copier first, then the pipe's writers)
        stop_copier();
        if (pipe_) pipe_->quiesce();
        if (zstream_) { (void)hipStreamSynchronize(zstream_); (void)hipStreamDestroy(zstream_); }     // (comp_ and the scratch are parked next)
        if (fd_ >= 0) ::close(fd_);
        if (gz_) gzclose(gz_);
    }

    // n bytes of FASTQ at d_img (device); pieces of one file must be added in file order
    void add(HostPipe& pipe, const uint8_t* d_img, uint64_t n) {
        pipe_ = &pipe;
        if (n == 0) return;
        plain_ += n;
        const int fd = fd_; const std::string fn = fn_; const bool null = null_;
        if (s_.compress == 0) {
            const uint64_t at = at_;
            at_ += n;
            if (null) pipe.copy(d_img, n, [](const uint8_t*, size_t, uint64_t) {}, true);
            else pipe.copy(d_img, n, [fd, fn, at](const uint8_t* p, size_t cnt, uint64_t off) { pwrite_all(fd, fn, p, cnt, at + off); }, false);
        } else if (!s_.bgzip) {
            pipe.copy(d_img, n, [&](const uint8_t* p, size_t cnt, uint64_t) {
                if (!null_ && gzwrite(gz_, p, (unsigned)cnt) != (int)cnt) throw Error(JK_ERR_IO, "gzwrite to " + fn_ + " failed");
            }, true);
        } else if (!s_.host_deflate) {
            // Two stages.  This thread compresses the piece where it lies (its own stream; the image slot is free again
            // when add() returns); a copier thread brings the compressed bytes to the host and on to the writer threads
            // while the next piece -- the other read end, the next launch -- is being generated and compressed.  Two
            // compressed buffers in rotation: piece k waits for the copy of piece k - 2.
            if (!zstream_) JK_HIP(hipStreamCreateWithFlags(&zstream_, hipStreamNonBlocking));
            const int kb = (int)(n_pieces_++ & 1u);
            {
                std::unique_lock<std::mutex> l(cm_);
                ccv_.wait(l, [&] { return !cbusy_[kb] || !cerr_.empty(); });
                if (!cerr_.empty()) { const std::string e = cerr_; cerr_.clear(); throw Error(JK_ERR_IO, e); }
            }
            DevBuf& comp = comp_[kb];
            if (comp.n < bgzf_bound(n)) comp.alloc(bgzf_bound(n) + (bgzf_bound(n) >> 3));
            const uint64_t nc = bgzf_deflate_device(s_.device, zstream_, d_img, n, comp.as<uint8_t>(), comp.n, nullptr, false, &bgzf_scratch_);
            const uint64_t at = at_;
            at_ += nc;
            {
                std::lock_guard<std::mutex> l(cm_);
                cbusy_[kb] = true;
                cq_.push_back(CopyTask{kb, nc, at});
                if (!copier_.joinable()) copier_ = std::thread([this, &pipe] { copy_loop(pipe); });
            }
            ccv_.notify_all();
        } else {
            const unsigned n_thr = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
            pipe.copy(d_img, n, [&](const uint8_t* buf, size_t cnt, uint64_t) {
                const size_t n_blocks = (cnt + BGZF_IN - 1) / BGZF_IN;
                std::vector<std::vector<uint8_t>> parts(n_thr);
                std::vector<std::string> errs(n_thr);
                std::vector<std::thread> pool;
                for (unsigned t = 0; t < n_thr; t++) pool.emplace_back([&, t] {
                    try {
                        const size_t b0 = n_blocks * t / n_thr, b1 = n_blocks * (t + 1) / n_thr;
                        for (size_t b = b0; b < b1; b++)
                            bgzf_compress_block(buf + b * BGZF_IN, std::min(BGZF_IN, cnt - b * BGZF_IN), s_.compress, parts[t]);
                    } catch (const std::exception& ex) { errs[t] = ex.what(); }
                });
                for (std::thread& th : pool) th.join();
                for (unsigned t = 0; t < n_thr; t++) {
                    if (!errs[t].empty()) throw Error(JK_ERR_IO, errs[t]);
                    if (!parts[t].empty()) { if (!null_) pwrite_all(fd_, fn_, parts[t].data(), parts[t].size(), at_); at_ += parts[t].size(); }
                }
            }, true);
        }
    }
    // all pieces are in: end-of-file block (unless another part follows in the same file), close
    void finish(HostPipe& pipe, bool with_eof = true) {
        {   // every compressed piece has left the device
            std::unique_lock<std::mutex> l(cm_);
            ccv_.wait(l, [&] { return (cq_.empty() && !cbusy_[0] && !cbusy_[1]) || !cerr_.empty(); });
            if (!cerr_.empty()) { const std::string e = cerr_; cerr_.clear(); throw Error(JK_ERR_IO, e); }
        }
        pipe.drain();
        if (s_.compress > 0 && s_.bgzip && with_eof) { if (!null_) pwrite_all(fd_, fn_, kBgzfEof, sizeof(kBgzfEof), at_); at_ += sizeof(kBgzfEof); }
        if (fd_ >= 0) { const int fd = fd_; fd_ = -1; if (::close(fd) != 0) throw Error(JK_ERR_IO, "error closing " + fn_); }
        if (gz_) { gzFile g = gz_; gz_ = nullptr; if (gzclose(g) != Z_OK) throw Error(JK_ERR_IO, "error closing " + fn_); }
    }
    void set_length(uint64_t n) { if (fd_ >= 0 && ::ftruncate(fd_, (off_t)n) != 0) throw Error(JK_ERR_IO, "ftruncate of " + fn_ + " failed: " + std::strerror(errno)); }
    uint64_t bytes_written() const { return at_; }     // (gzip streams: not tracked)
    uint64_t plain_bytes() const { return plain_; }
    const std::string& name() const { return fn_; }
    static void pwrite_all(int fd, const std::string& fn, const uint8_t* p, size_t n, uint64_t at) {
        while (n) {
            const ssize_t w = ::pwrite(fd, p, n, (off_t)at);
            if (w < 0) { if (errno == EINTR) continue; throw Error(JK_ERR_IO, "write to " + fn + " failed: " + std::strerror(errno)); }
            p += w; n -= (size_t)w; at += (uint64_t)w;
        }
    }
private:
    struct CopyTask { int buf; uint64_t n; uint64_t at; };
    void copy_loop(HostPipe& pipe) {
        (void)hipSetDevice(s_.device);
        for (;;) {
            CopyTask k;
            {
                std::unique_lock<std::mutex> l(cm_);
                ccv_.wait(l, [&] { return cstop_ || !cq_.empty(); });
                if (cq_.empty()) return;
                k = cq_.front(); cq_.pop_front();
            }
            std::string err;
            try {
                const int fd = fd_; const std::string fn = fn_; const uint64_t at = k.at;
                if (null_) pipe.copy(comp_[k.buf].as<uint8_t>(), k.n, [](const uint8_t*, size_t, uint64_t) {}, true);
                else pipe.copy(comp_[k.buf].as<uint8_t>(), k.n, [fd, fn, at](const uint8_t* p, size_t cnt, uint64_t off) { pwrite_all(fd, fn, p, cnt, at + off); }, false);
            } catch (const std::exception& e) { err = e.what(); }
            { std::lock_guard<std::mutex> l(cm_); cbusy_[k.buf] = false; if (!err.empty() && cerr_.empty()) cerr_ = err; }
            ccv_.notify_all();
        }
    }
    void stop_copier() {
        { std::lock_guard<std::mutex> l(cm_); cstop_ = true; cq_.clear(); }
        ccv_.notify_all();
        if (copier_.joinable()) copier_.join();
    }
    const jk_session& s_;
    std::string fn_;
    bool null_ = false;
    int fd_ = -1; gzFile gz_ = nullptr;
    uint64_t at_ = 0;            // next byte of the file
    uint64_t plain_ = 0;         // FASTQ bytes taken so far
    HostPipe* pipe_ = nullptr;   // the pipe this file's pieces went through
    DevBuf comp_[2];             // compressed pieces on their way to the host, in rotation
    BgzfScratch bgzf_scratch_;   // kept from piece to piece
    hipStream_t zstream_ = nullptr;
    uint64_t n_pieces_ = 0;
    std::thread copier_;
    std::mutex cm_; std::condition_variable ccv_;
    std::deque<CopyTask> cq_;
    bool cbusy_[2] = {false, false}, cstop_ = false;
    std::string cerr_;
};

static const size_t PIPE_PIECE = BGZF_IN * 512;      // 33.4 MB pieces, a whole number of BGZF blocks
inline int pipe_writers() {
    if (const char* e = std::getenv("JK_WRITER_THREADS")) { const int v = std::atoi(e); if (v >= 0) return std::min(v, 16); }
    return 4;
}

// The session's host pipe: 200 MB of pinned memory and the writer threads, allocated once per session (hipHostMalloc of
// the ring takes tens of milliseconds -- as long as the D2H copy of a small job's whole output).
static HostPipe& session_pipe(const jk_session& s) {
    jk_session& m = const_cast<jk_session&>(s);
    if (!m.host_pipe) m.host_pipe = std::make_shared<HostPipe>(PIPE_PIECE, 6, pipe_writers());
    return *m.host_pipe;
}

// A lane shard's image at its place in the shared file (see jk_session_write_shard).
static void write_shard(const jk_session& s, const uint64_t* file_offset) {
    if (!s.generated) throw Error(JK_ERR_ARG, "jk_session_write_shard before jk_session_generate");
    if (s.compress > 0) throw Error(JK_ERR_UNSUPPORTED, "lane shards write uncompressed FASTQ only (compress the assembled file, or give each rank its own out_prefix)");
    if (s.out_prefix.empty()) throw Error(JK_ERR_ARG, "out_prefix is empty");
    HostPipe& pipe = session_pipe(s);
    for (uint32_t e = 0; e < s.n_ends; e++) {
        FastqFile f(s, e, "", false, file_offset[e]);
        // the shared file is opened without truncation (the other shards write into it too): the shard that holds the run's
        // last lane knows where the file ends and cuts off what an older, longer file of that name would leave behind
        if (s.lane_end == s.n_lanes_total) f.set_length(file_offset[e] + s.bytes[e]);
        f.add(pipe, s.d_out[e].as<uint8_t>(), s.bytes[e]);
        f.finish(pipe);
    }
}

// The resident image of a session into its files (jk_session_write).
static void write_files(const jk_session& s) {
    if (!s.generated) throw Error(JK_ERR_ARG, "jk_session_write before jk_session_generate");
    if (s.streaming) throw Error(JK_ERR_ARG, "this session streams its output (stream_output): use jk_session_run");
    if (s.n_shard != s.n_lanes_total)
        throw Error(JK_ERR_UNSUPPORTED, "this session holds lanes " + std::to_string(s.lane_begin) + ".." + std::to_string(s.lane_end) + " of " +
                    std::to_string(s.n_lanes_total) + ": writing it as the whole file would drop the other lanes' reads -- use "
                    "jk_session_write_shard with the byte offsets from the ranks' jk_session_sizes");
    HostPipe& pipe = session_pipe(s);
    for (uint32_t e = 0; e < s.n_ends; e++) {
        FastqFile f(s, e);
        f.add(pipe, s.d_out[e].as<uint8_t>(), s.bytes[e]);
        f.finish(pipe);
    }
}

}  // namespace jk
