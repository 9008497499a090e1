"""Inputs of the read-generation path: reference genomes (and, later, haplotype sets).

These mirror only what the sequencers *read* from jackalope's objects: ``RefGenome``
(/root/reference/src/ref_classes.h:127-180: name "REF", chromosomes {name, nucleos}).  Genome
creation/evolution are out of scope; ``synthetic_genome`` is this repo's own seeded generator for
tests and benchmarks.
"""
import ctypes as C

import numpy as np

from . import _abi


class RefGenome:
    """Named chromosomes held as ASCII bytes (one byte per base, as the reference stores them)."""

    def __init__(self, seqs, names=None, name="REF"):
        self.seqs = [np.frombuffer(s.encode() if isinstance(s, str) else bytes(s), dtype=np.uint8)
                     if not isinstance(s, np.ndarray) else np.ascontiguousarray(s, dtype=np.uint8) for s in seqs]
        # make_ref_genome names chromosomes chrom0.. (/root/reference/src/ref_hap_access.cpp:109-113)
        self.names = list(names) if names is not None else ["chrom%d" % i for i in range(len(self.seqs))]
        if len(self.names) != len(self.seqs):
            raise ValueError("names and seqs differ in length")
        self.name = name

    def n_chroms(self):
        return len(self.seqs)

    def sizes(self):
        return [int(s.size) for s in self.seqs]

    def _view(self):
        """(jk_ref_genome struct, keep-alive list)"""
        n = len(self.seqs)
        names = (C.c_char_p * n)(*[x.encode() for x in self.names])
        ptrs = (C.c_void_p * n)(*[s.ctypes.data for s in self.seqs])
        lens = (C.c_uint64 * n)(*[s.size for s in self.seqs])
        v = _abi.RefGenomeView(n, names, ptrs, lens, self.name.encode())
        return v, [names, ptrs, lens, self.seqs]


def synthetic_genome(chrom_sizes, seed, alphabet=b"TCAG"):
    """iid-uniform chromosomes over ``alphabet`` from numpy's seeded generator (not the reference's
    create_genome; the sequencers only need *a* genome)."""
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(alphabet, dtype=np.uint8)
    seqs = [lut[rng.integers(0, lut.size, size=int(n), dtype=np.uint8)] for n in chrom_sizes]
    return RefGenome(seqs)


class HapSet:
    """Variant haplotypes as the sequencers see them: a reference genome plus, per haplotype and
    chromosome, a mutation table (old_pos, new_pos, nucleos) and the haplotype chromosome's size --
    the read side of HapSet/HapGenome/HapChrom/AllMutations
    (/root/reference/src/hap_classes.h:100-258, 280-333, 500-611).

    ``cells[h][c]`` is a dict with ``chrom_size`` (int), ``old_pos``/``new_pos`` (int lists, new_pos
    ascending) and ``nucleos`` (list of str; "" = deletion, the reference's nullptr).
    """

    def __init__(self, ref, cells, names=None):
        self.ref = ref
        self.cells = cells
        # HapSet(ref, n) names haplotypes hap0.. (/root/reference/src/hap_classes.h:546-550)
        self.names = list(names) if names is not None else ["hap%d" % i for i in range(len(cells))]

    def n_haps(self):
        return len(self.cells)

    def hap_names(self):
        return list(self.names)

    def seed_budget(self, n_threads):
        """Upper bound on the 32-bit seed words an illumina()/pacbio() run consumes: per lane 8 (engine)
        + 8 (haplotype split) + 16 per haplotype; sep_files adds one 8-word draw and repeats per file."""
        return (int(n_threads) * (16 + 16 * self.n_haps()) + 8) * max(1, self.n_haps())

    def _view(self):
        nh, nc = self.n_haps(), self.ref.n_chroms()
        rv, keep = self.ref._view()
        chrom_size = np.zeros(nh * nc, dtype=np.uint64)
        n_mut = np.zeros(nh * nc, dtype=np.uint64)
        old_pos, new_pos, nuc_off, blob = [], [], [0], []
        for h in range(nh):
            for c in range(nc):
                cell = self.cells[h][c]
                k = h * nc + c
                chrom_size[k] = cell["chrom_size"]
                n_mut[k] = len(cell["new_pos"])
                old_pos += list(cell["old_pos"])
                new_pos += list(cell["new_pos"])
                for s in cell["nucleos"]:
                    blob.append(s.encode() if isinstance(s, str) else bytes(s))
                    nuc_off.append(nuc_off[-1] + len(blob[-1]))
        old_pos = np.asarray(old_pos, dtype=np.uint64)
        new_pos = np.asarray(new_pos, dtype=np.uint64)
        nuc_off = np.asarray(nuc_off, dtype=np.uint64)
        blob = np.frombuffer(b"".join(blob) + b"\0", dtype=np.uint8)
        names = (C.c_char_p * nh)(*[x.encode() for x in self.names])
        v = _abi.HapSetView()
        v.n_haps, v.n_chroms = nh, nc
        v.hap_names = names
        v.ref = rv
        v.chrom_size = chrom_size.ctypes.data_as(C.POINTER(C.c_uint64))
        v.n_mut = n_mut.ctypes.data_as(C.POINTER(C.c_uint64))
        v.old_pos = old_pos.ctypes.data_as(C.POINTER(C.c_uint64))
        v.new_pos = new_pos.ctypes.data_as(C.POINTER(C.c_uint64))
        v.nuc_off = nuc_off.ctypes.data_as(C.POINTER(C.c_uint64))
        v.nuc_blob = blob.ctypes.data
        return v, [keep, chrom_size, n_mut, old_pos, new_pos, nuc_off, blob, names]

    def materialize(self, hap, chrom):
        """The haplotype chromosome as bytes, by plain string editing of the reference (independent of
        get_chrom_full's index arithmetic; what the reference's R tests do with substr/paste0)."""
        ref = self.ref.seqs[chrom].tobytes()
        cell = self.cells[hap][chrom]
        out, r = [], 0
        n = len(cell["new_pos"])
        for m in range(n):
            op, nuc = int(cell["old_pos"][m]), cell["nucleos"][m]
            nuc = nuc.encode() if isinstance(nuc, str) else bytes(nuc)
            out.append(ref[r:op])
            if m + 1 < n:
                smod = (int(cell["new_pos"][m + 1]) - int(cell["old_pos"][m + 1])) - (int(cell["new_pos"][m]) - op)
            else:
                smod = (int(cell["chrom_size"]) - len(ref)) - (int(cell["new_pos"][m]) - op)
            if smod >= 0:                # substitution (0) or insertion (k): nucleos replace ref[op]
                out.append(nuc[:smod + 1])
                r = op + 1
            else:                        # deletion of -smod bases starting at op
                r = op - smod
        out.append(ref[r:])
        return b"".join(out)


def random_haplotypes(ref, n_haps, seed, sub_rate=1e-3, ins_rate=1e-4, del_rate=1e-4, mean_indel=3.0):
    """Synthetic mutation tables in the reference's canonical representation (stand-in for
    haps_phylo()/create_haplotypes(), which are out of scope): substitutions, insertions and deletions
    at the given per-base rates, geometric indel lengths, never overlapping or adjacent."""
    rng = np.random.default_rng(seed)
    alphabet = b"TCAG"
    cells = []
    for h in range(n_haps):
        row = []
        for seq in ref.seqs:
            n = int(seq.size)
            total = sub_rate + ins_rate + del_rate
            k = int(rng.binomial(n, min(total, 0.3))) if n > 4 else 0
            pos = np.sort(rng.choice(n, size=k, replace=False)) if k else np.zeros(0, dtype=np.int64)
            kinds = rng.choice(3, size=k, p=np.array([sub_rate, ins_rate, del_rate]) / total) if k else []
            old_pos, new_pos, nucleos = [], [], []
            shift, next_free = 0, 0
            for p, kind in zip(pos.tolist(), list(kinds)):
                if p < next_free:
                    continue
                if kind == 0:
                    b = alphabet[(alphabet.find(bytes([seq[p]])) + 1 + int(rng.integers(0, 3))) % 4] \
                        if bytes([seq[p]]) in alphabet else alphabet[0]
                    old_pos.append(p); new_pos.append(p + shift); nucleos.append(chr(b))
                    next_free = p + 2
                elif kind == 1:
                    ln = int(rng.geometric(1.0 / mean_indel))
                    ins = bytes(alphabet[i] for i in rng.integers(0, 4, size=ln))
                    old_pos.append(p); new_pos.append(p + shift); nucleos.append(chr(seq[p]) + ins.decode())
                    shift += ln
                    next_free = p + 2
                else:
                    ln = min(int(rng.geometric(1.0 / mean_indel)), n - p)
                    if p + ln >= n:      # keep at least one base after a deletion
                        continue
                    old_pos.append(p); new_pos.append(p + shift); nucleos.append("")
                    shift -= ln
                    next_free = p + ln + 1
            row.append({"chrom_size": n + shift, "old_pos": old_pos, "new_pos": new_pos, "nucleos": nucleos})
        cells.append(row)
    return HapSet(ref, cells)
