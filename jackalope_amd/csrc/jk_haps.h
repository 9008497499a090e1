// Host-side mutation tables, write side: what HapChrom::add_substitution / add_insertion /
// add_deletion do to one haplotype chromosome (/root/reference/src/hap_classes.cpp:298-771), kept
// in the same canonical form the sequencers read (old_pos, new_pos, nucleos; "" = deletion).
//
// A mutation's effect on chromosome length is never stored: it is derived from the positions of the
// mutation after it (src/hap_classes.h:312-333).  Every edit below therefore keeps the ORDER in which
// the reference touches positions, sizes and nucleotides, because the derived lengths are read while
// the table is half-updated.
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

namespace jk {

struct HapCell {
    const char* ref = nullptr;     // reference chromosome, ASCII
    uint64_t ref_len = 0;
    uint64_t size = 0;             // HapChrom::chrom_size
    std::vector<uint64_t> op, np;  // AllMutations::old_pos / new_pos
    std::vector<std::string> nt;   // AllMutations::nucleos ("" = nullptr = deletion)

    size_t count() const { return np.size(); }

    // HapChrom::size_modifier (src/hap_classes.h:312-333)
    int64_t delta(size_t i) const {
        const int64_t after = (i + 1 < np.size()) ? (int64_t)(np[i + 1] - op[i + 1]) : (int64_t)(size - ref_len);
        return after + (int64_t)(op[i] - np[i]);
    }

    // HapChrom::get_mut_ (src/hap_classes.cpp:725-771): last mutation with new_pos <= pos, count() if
    // none.  The reference guesses an index and walks; a binary search lands on the same record.
    // Returns false if pos is outside the chromosome (the reference stops with an error).
    bool locate(uint64_t pos, size_t* out) const {
        if (np.empty()) { *out = 0; return true; }
        if (pos >= size) return false;
        const size_t ub = std::upper_bound(np.begin(), np.end(), pos) - np.begin();
        *out = ub == 0 ? np.size() : ub - 1;
        return true;
    }

    void put(size_t i, uint64_t o, uint64_t n, std::string s) {
        op.insert(op.begin() + i, o);
        np.insert(np.begin() + i, n);
        nt.insert(nt.begin() + i, std::move(s));
    }
    void drop(size_t a, size_t b) {
        op.erase(op.begin() + a, op.begin() + b);
        np.erase(np.begin() + a, np.begin() + b);
        nt.erase(nt.begin() + a, nt.begin() + b);
    }
    // HapChrom::calc_positions (src/hap_classes.h:346-355)
    void shift_from(size_t i, int64_t d) {
        for (; i < np.size(); ++i) np[i] += d;
        size += d;
    }

    // add_substitution (src/hap_classes.cpp:475-509)
    bool substitute(char c, uint64_t pos) {
        size_t m;
        if (!locate(pos, &m)) return false;
        if (m == count()) { put(0, pos, pos, std::string(1, c)); return true; }
        const uint64_t ind = pos - np[m];
        const int64_t sm = delta(m);
        if ((int64_t)ind <= sm) {
            // inside this mutation's own bytes; a substitution back to the reference base disappears
            if (sm == 0 && ref[op[m]] == c) drop(m, m + 1);
            else nt[m][ind] = c;
        } else {
            put(m + 1, ind + (op[m] - sm), pos, std::string(1, c));
        }
        return true;
    }

    // add_insertion (src/hap_classes.cpp:416-466): `s` goes in AFTER position pos; the record keeps
    // the base it follows as its first byte.
    bool insert(const std::string& s, uint64_t pos) {
        const int64_t grow = (int64_t)s.size();
        size_t m;
        if (!locate(pos, &m)) return false;
        if (m == count()) {
            put(0, pos, pos, std::string(1, ref[pos]) + s);
            shift_from(1, grow);
            return true;
        }
        const uint64_t ind = pos - np[m];
        const int64_t sm = delta(m);
        if ((int64_t)ind <= sm) {
            nt[m].insert(ind + 1, s);
            shift_from(m + 1, grow);
        } else {
            const uint64_t o = ind + (op[m] - sm);
            put(m + 1, o, pos, std::string(1, ref[o]) + s);
            shift_from(m + 2, grow);
        }
        return true;
    }

    // deletion_old_pos_ (src/hap_classes.cpp:521-569): reference position the new deletion record
    // would carry, read before anything is edited.
    uint64_t deletion_anchor(uint64_t d0, size_t k) const {
        if (np[k] == d0) return op[k];
        if (np[k] > d0) return d0;
        const int64_t sm = delta(k);
        const uint64_t end = np[k] + sm;
        if (sm <= 0 || end < d0) return d0 - np[k] + op[k] - sm;
        return op[k] + 1;
    }

    // add_deletion (src/hap_classes.cpp:295-408) with deletion_one_mut_ (:583-707) folded into the loop
    void remove(uint64_t len, uint64_t pos) {
        if (len == 0 || pos >= size) return;
        const uint64_t d0 = pos, d1 = std::min(pos + len - 1, size - 1);
        const int64_t shrink = (int64_t)(d0 - d1 - 1);

        if (np.empty()) { put(0, d0, d0, std::string()); size += shrink; return; }

        if (np.front() > d1) {
            // everything lies after the deleted stretch; a deletion that abuts it absorbs the new one
            const bool abuts = np.front() == d1 + 1 && delta(0) < 0;
            for (uint64_t& p : np) p += shrink;
            if (abuts) op.front() += shrink; else put(0, d0, d0, std::string());
            size += shrink;
            return;
        }

        const bool head_overlap = np.front() > d0 && np.front() <= d1;

        // first record at d0, else the last one before it
        size_t k;
        if (np.back() < d0) k = count() - 1;
        else {
            k = std::lower_bound(np.begin(), np.end(), d0) - np.begin();
            if (np[k] > d0 && k > 0) --k;
        }
        const uint64_t anchor = deletion_anchor(d0, k);

        int64_t remaining = shrink;          // part of the deletion not absorbed by insertions
        size_t gone_lo = 0, gone_hi = 0, n_gone = 0;
        auto mark = [&](size_t i) { if (n_gone++ == 0) gone_lo = i; gone_hi = i; };

        for (size_t i = k; i < count(); ++i) {
            uint64_t& p = np[i];
            if (p > d1 + 1) { p += shrink; continue; }
            const int64_t sm = delta(i);
            if (sm == 0) {                                   // substitution
                if (p > d1) p += shrink;
                else if (p >= d0) mark(i);
            } else if (sm > 0) {                             // insertion
                if (p > d1) { p += shrink; continue; }
                const uint64_t end = p + sm;
                if (end < d0) continue;
                if (d0 <= p && d1 >= end) { remaining += sm; mark(i); continue; }
                int64_t lead = (int64_t)(d0 - p);
                if (lead < 0) lead = 0;
                const uint64_t e0 = (uint64_t)lead;
                const uint64_t e1 = std::min<uint64_t>(d1 - p + 1, nt[i].size());
                remaining += (int64_t)(e1 - e0);
                nt[i].erase(e0, e1 - e0);
                if (d0 < p && d1 < end) { p += e1 - e0; p += shrink; }
            } else {                                         // deletion: merged into the new one
                if (p < d0) continue;
                remaining += sm;
                mark(i);
            }
        }

        size += shrink;

        size_t at;
        if (n_gone == 1) { drop(gone_lo, gone_lo + 1); at = gone_lo; }
        else if (n_gone > 1) { drop(gone_lo, gone_hi + 1); at = gone_lo; }
        else { at = k; if (!head_overlap) ++at; }

        if (remaining >= 0) return;
        put(at, anchor, d0, std::string());
    }
};

}  // namespace jk
