// api_sinks.h -- output sinks: plain, gzip, BGZF on the host or on the device; pipelined copy-out
// (part of the one translation unit jk_api.hip; see the include list there)
#pragma once

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

// A lane shard's image at its place in the shared file (see jk_session_write_shard).
static void write_shard(const jk_session& s, const uint64_t* file_offset) {
    if (!s.generated) throw Error(JK_ERR_ARG, "jk_session_write_shard before jk_session_generate");
    if (s.compress > 0) throw Error(JK_ERR_UNSUPPORTED, "lane shards write uncompressed FASTQ only (compress the assembled file, or give each rank its own out_prefix)");
    for (uint32_t e = 0; e < s.n_ends; e++) {
        const std::string fn = s.out_prefix + "_R" + std::to_string(e + 1) + ".fq";
        struct Fd { int fd = -1; ~Fd() { if (fd >= 0) ::close(fd); } } F;
        F.fd = ::open(fn.c_str(), O_WRONLY | O_CREAT, 0644);
        if (F.fd < 0) throw Error(JK_ERR_IO, "Unable to open file " + fn + ".\n");
        uint64_t at = file_offset[e];
        stream_to_host(s.d_out[e].as<uint8_t>(), s.bytes[e], BGZF_IN * 1024, [&](const uint8_t* p, size_t n) {
            while (n) {
                const ssize_t w = ::pwrite(F.fd, p, n, (off_t)at);
                if (w < 0) { if (errno == EINTR) continue; throw Error(JK_ERR_IO, "write to " + fn + " failed"); }
                p += w; n -= (size_t)w; at += (uint64_t)w;
            }
        });
        const int fd = F.fd; F.fd = -1;
        if (::close(fd) != 0) throw Error(JK_ERR_IO, "error closing " + fn);
    }
}

static void write_files(const jk_session& s) {
    if (!s.generated) throw Error(JK_ERR_ARG, "jk_session_write before jk_session_generate");
    if (s.n_shard != s.n_lanes_total)
        throw Error(JK_ERR_UNSUPPORTED, "this session holds lanes " + std::to_string(s.lane_begin) + ".." + std::to_string(s.lane_end) + " of " +
                    std::to_string(s.n_lanes_total) + ": writing it as the whole file would drop the other lanes' reads -- use "
                    "jk_session_write_shard with the byte offsets from the ranks' jk_session_sizes");
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

}  // namespace jk
