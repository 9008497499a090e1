// api_builder.h -- C ABI of the host-side helpers and of the mutation-table builder
// (part of the one translation unit jk_api.hip; see the include list there)
#pragma once

extern "C" {

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

// The per-lane quota planner of the sessions on its own (jk_plan.h): host only, no device needed.
int jk_plan_lane_quotas(int32_t hap, uint32_t n_ends, int32_t maker_halves, const double* hap_probs, uint64_t n_haps,
                        const double* chrom_probs, uint64_t n_chroms, uint64_t n_reads, uint64_t n_threads,
                        uint64_t lane_begin, uint64_t lane_end, jk_seed_source* seeds, int32_t offset_given, uint64_t offset_words,
                        uint32_t* lane_seeds, uint32_t* quotas, uint64_t* words3) {
    return guarded([&] {
        if (!seeds || !chrom_probs || n_ends == 0 || n_threads == 0) throw Error(JK_ERR_ARG, "bad argument");
        if (lane_end == 0) lane_end = n_threads;
        if (lane_begin > lane_end || lane_end > n_threads) throw Error(JK_ERR_ARG, "lane shard out of range");
        std::vector<uint64_t> per_lane = split_int(n_reads / n_ends, n_threads);
        for (uint64_t& v : per_lane) v *= n_ends;
        QuotaModel Q;
        Q.hap = hap != 0; Q.n_ends = n_ends; Q.maker_halves = maker_halves != 0; Q.n_haps = Q.hap ? n_haps : 1; Q.n_chroms = n_chroms;
        if (Q.hap) Q.hap_chain = GroupChain(std::vector<double>(hap_probs, hap_probs + n_haps));
        for (uint64_t h = 0; h < Q.n_haps; h++) Q.chrom_chain.emplace_back(std::vector<double>(chrom_probs + h * n_chroms, chrom_probs + (h + 1) * n_chroms));
        SeedReader r{*seeds};
        // JK_PLAN_HOOK_DEFER=1 (tests): plan the way the sessions do -- the chromosome-level splits as a task list (what
        // chrom_split_kernel runs on the device) -- and run the tasks here, on the host
        const bool defer = std::getenv("JK_PLAN_HOOK_DEFER") != nullptr;
        LanePlan lp = plan_lane_quotas(Q, per_lane, lane_begin, lane_end, r, offset_given != 0, offset_words, defer);
        if (lane_seeds) std::memcpy(lane_seeds, lp.lane_seeds.data(), lp.lane_seeds.size() * 4);
        if (quotas && !lp.deferred) std::memcpy(quotas, lp.quotas.data(), lp.quotas.size() * 4);
        if (quotas && lp.deferred) {
            const uint64_t n_shard = lane_end - lane_begin, nc = n_chroms, nt = lp.n_tasks();
            std::memset(quotas, 0, (size_t)Q.n_haps * nc * n_shard * 4);
            std::vector<uint32_t> vals(nt * nc);
            run_tasks_on_host(Q, lp, nullptr, 0, vals.data());
            for (uint64_t k = 0; k < nt; k++)
                for (uint64_t g = 0; g < nc; g++)
                    quotas[((uint64_t)(Q.hap ? lp.task_hap[k] : 0) * nc + g) * n_shard + lp.task_lane[k]] = vals[k * nc + g];
        }
        if (words3) { words3[0] = lp.words_used; words3[1] = lp.shard_begin_word; words3[2] = lp.shard_end_word; }
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

}  // extern "C"
