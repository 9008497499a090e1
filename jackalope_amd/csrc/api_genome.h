// api_genome.h -- C ABI of create_genome and read_fasta (device-resident genomes)
// (part of the one translation unit jk_api.hip; see the include list there)
#pragma once

extern "C" {

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
        JK_HIP(hipSetDevice(device));
        std::unique_ptr<jk_genome> G(new jk_genome);
        G->device = device;

        // ---- host: seeds, lengths, first state of every chromosome
        SeedReader sr{*seeds};
        const uint64_t T = n_threads;
        std::vector<uint32_t> lane_seed(T * 8);
        for (uint64_t t = 0; t < T; t++) sr.take8(&lane_seed[t * 8]);           // mt_seeds (src/pcg.h:63-71)
        G->seed_words_used = sr.pos;
        const jk_gamma_param gp = jk_gamma_make(len_sd > 0 ? shape : 1.0, scale);      // (host draws: pow falls back to libm itself)
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
            jk_gamma_state gs{0.0, 0, 0};
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
        EventPair ev;                  // (destroyed on every path out of here)
        JK_HIP(hipEventRecord(ev.a, nullptr));
        hipLaunchKernelGGL(create_genome_kernel, dim3((uint32_t)grid), dim3(GENOME_BLOCK), 0, nullptr, P);
        JK_HIP(hipGetLastError());
        JK_HIP(hipEventRecord(ev.b, nullptr));
        JK_HIP(hipDeviceSynchronize());
        G->ms = ev.ms();
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
    EventPair ev;
    JK_HIP(hipEventRecord(ev.a, nullptr));
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
    JK_HIP(hipEventRecord(ev.b, nullptr));
    JK_HIP(hipDeviceSynchronize());
    if (ms) *ms += ev.ms();
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

}  // extern "C"
