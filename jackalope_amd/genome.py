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
        v = _abi.RefGenomeView(n, names, ptrs, lens, self.name.encode(), 0)
        return v, [names, ptrs, lens, self.seqs]


class DeviceGenome(RefGenome):
    """A reference genome that was made on the GPU and lives there (``jk_genome``); what ``create_genome``
    returns.  illumina()/pacbio() read it in place; ``seqs`` (host copies) are fetched on first use."""

    def __init__(self, handle, n_chroms=None):
        self._h = handle
        L = _abi.lib()
        v = _abi.RefGenomeView()
        _abi.check(L.jk_genome_view(self._h, C.byref(v)))
        n_chroms = int(v.n_chroms) if n_chroms is None else n_chroms
        self.names = [v.chrom_names[i].decode(errors="replace") for i in range(n_chroms)]
        self._sizes = [int(v.chrom_lens[i]) for i in range(n_chroms)]
        self.name = "REF"
        self._seqs = None

    def n_chroms(self):
        return len(self._sizes)

    def sizes(self):
        return list(self._sizes)

    def seed_words_used(self):
        return int(_abi.lib().jk_genome_seed_words_used(self._h))

    def kernel_ms(self):
        return float(_abi.lib().jk_genome_ms(self._h))

    def chrom(self, i):
        out = np.empty(max(self._sizes[i], 1), dtype=np.uint8)
        _abi.check(_abi.lib().jk_genome_fetch(self._h, i, out.ctypes.data, out.size))
        return out[:self._sizes[i]]

    @property
    def seqs(self):
        if self._seqs is None:
            self._seqs = [self.chrom(i) for i in range(self.n_chroms())]
        return self._seqs

    def _view(self):
        v = _abi.RefGenomeView()
        _abi.check(_abi.lib().jk_genome_view(self._h, C.byref(v)))
        return v, [self]

    def close(self):
        if self._h:
            _abi.lib().jk_genome_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def create_genome(n_chroms, len_mean, len_sd=0, pi_tcag=(0.25, 0.25, 0.25, 0.25), n_threads=1, seed_words=None,
                  seed=None, device=0):
    """create_genome() of the reference (/root/reference/R/create_genome.R:26-56 -> create_genome_cpp,
    /root/reference/src/create_sequences.cpp:151-169), made on the GPU.  ``seed_words``: 8 32-bit words per
    thread (what the reference draws from R's RNG); or ``seed`` for this package's SplitMix64 stream."""
    from .rng import seed_words as _sw

    def err(par, what):
        raise ValueError("\nFor the `create_genome` function in jackalope, argument `%s` must be %s." % (par, what))
    if not isinstance(n_chroms, (int, np.integer)) or isinstance(n_chroms, bool) or n_chroms < 1:
        err("n_chroms", "a single integer >= 1")
    if not isinstance(len_mean, (int, float, np.integer, np.floating)) or isinstance(len_mean, bool) or not len_mean >= 1:
        err("len_mean", "a single number >= 1")
    if not isinstance(len_sd, (int, float, np.integer, np.floating)) or isinstance(len_sd, bool) or not len_sd >= 0:
        err("len_sd", "a single number >= 0")
    pi = np.asarray(pi_tcag, dtype=np.float64)
    if pi.shape != (4,) or np.any(~(pi >= 0)) or np.all(pi == 0):
        err("pi_tcag", "a numeric vector of length 4, where no number can be < 0 and at least one must be > 0")
    if not isinstance(n_threads, (int, np.integer)) or isinstance(n_threads, bool) or n_threads < 1:
        err("n_threads", "a single integer >= 1")
    if seed_words is None:
        seed_words = _sw(0 if seed is None else seed, 8 * int(n_threads))
    words = np.ascontiguousarray(seed_words, dtype=np.uint32)
    src = _abi.SeedSource()
    src.words = words.ctypes.data_as(C.POINTER(C.c_uint32))
    src.n_words = words.size
    h = C.c_void_p()
    _abi.check(_abi.lib().jk_create_genome(int(n_chroms), float(len_mean), float(len_sd), pi.ctypes.data_as(C.POINTER(C.c_double)),
                                           int(n_threads), C.byref(src), int(device), C.byref(h)))
    return DeviceGenome(h, int(n_chroms))


def read_fasta(fasta_files, fai_files=None, cut_names=False, device=0):
    """read_fasta() of the reference (/root/reference/R/read_write.R:24-52 -> read_fasta_noind /
    read_fasta_ind, /root/reference/src/io_fasta.cpp:153-169, :389-408): plain, gzip or bgzip FASTA
    file(s), optionally with index files, into a genome that is packed on and stays on the GPU.
    Soft masking is always removed, as the R function does."""
    def err(par, *what):
        raise ValueError("\nFor the `read_fasta` function in jackalope, argument `%s` must be %s." % (par, " ".join(what)))
    if isinstance(fasta_files, str):
        fasta_files = [fasta_files]
    if not isinstance(fasta_files, (list, tuple)) or len(fasta_files) == 0 or not all(isinstance(f, str) for f in fasta_files):
        err("fasta_files", "a character vector")
    if isinstance(fai_files, str):
        fai_files = [fai_files]
    if fai_files is not None and (not isinstance(fai_files, (list, tuple)) or len(fai_files) != len(fasta_files)
                                  or not all(isinstance(f, str) for f in fai_files)):
        err("fai_files", "NULL or a character vector of the same", "length as the `fasta_files` argument")
    if not isinstance(cut_names, (bool, np.bool_)):
        err("cut_names", "a single logical")
    n = len(fasta_files)
    fa = (C.c_char_p * n)(*[f.encode() for f in fasta_files])
    fai = (C.c_char_p * n)(*[f.encode() for f in fai_files]) if fai_files is not None else None
    h = C.c_void_p()
    _abi.check(_abi.lib().jk_read_fasta(fa, fai, n, int(bool(cut_names)), 1, int(device), C.byref(h)))
    return DeviceGenome(h)


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


class HapBuilder:
    """Editable haplotype set: the write side of the reference's ``haplotypes`` R6 class
    (/root/reference/R/aaa-classes.R:791-850, 886-893) over the native mutation-table builder
    (``jk_hap_builder``; HapChrom::add_substitution/add_insertion/add_deletion,
    /root/reference/src/hap_classes.cpp:295-509).  Indices and positions are 1-based like the R
    methods; ``snapshot()`` gives the ``HapSet`` the sequencers take."""

    _NTS = set("TCAGN")

    def __init__(self, ref, n_haps=None, _from=None):
        self.ref = ref
        self._h = C.c_void_p()
        L = _abi.lib()
        if _from is not None:
            v, keep = _from._view()
            _abi.check(L.jk_hap_builder_from(C.byref(v), C.byref(self._h)))
            self._n_haps = _from.n_haps()
        else:
            rv, keep = ref._view()
            _abi.check(L.jk_hap_builder_new(C.byref(rv), int(n_haps), C.byref(self._h)))
            self._n_haps = int(n_haps)
        self._keep = keep          # the chromosome bytes are borrowed by the builder

    @classmethod
    def from_hapset(cls, hs):
        return cls(hs.ref, _from=hs)

    def close(self):
        if self._h:
            _abi.lib().jk_hap_builder_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def n_haps(self):
        return self._n_haps

    def n_chroms(self):
        return self.ref.n_chroms()

    def _raw_view(self):
        v = _abi.HapSetView()
        _abi.check(_abi.lib().jk_hap_builder_view(self._h, C.byref(v)))
        return v

    def sizes(self, hap_ind):
        """Chromosome sizes of one haplotype (1-based), like haplotypes$sizes(hap_ind)."""
        self._check_hap(hap_ind, "sizes")
        v, nc = self._raw_view(), self.n_chroms()
        return [int(v.chrom_size[(hap_ind - 1) * nc + c]) for c in range(nc)]

    def _err(self, fun, arg, what):
        # err_msg() of /root/reference/R/util.R:46-49
        raise ValueError("\nFor the `%s` function in jackalope, argument `%s` must be %s." % (fun, arg, what))

    def _check_hap(self, hap_ind, fun):
        if not isinstance(hap_ind, (int, np.integer)) or not 1 <= hap_ind <= self._n_haps:
            self._err(fun, "hap_ind", "integer in range [1, <# haplotypes>]")

    def _check_pos(self, hap_ind, chrom_ind, pos, fun):
        if not isinstance(chrom_ind, (int, np.integer)) or not 1 <= chrom_ind <= self.n_chroms():
            self._err(fun, "chrom_ind", "integer in range [1, <# chromosomes>]")
        self._check_hap(hap_ind, fun)
        if not isinstance(pos, (int, np.integer)) or not 1 <= pos <= self.sizes(hap_ind)[chrom_ind - 1]:
            self._err(fun, "pos", "integer in range [1, <chromosome size>]")

    def add_sub(self, hap_ind, chrom_ind, pos, nt):
        self._check_pos(hap_ind, chrom_ind, pos, "add_sub")
        if not isinstance(nt, str) or len(nt) != 1:
            self._err("add_sub", "nt", "a single character")
        if nt not in self._NTS:
            self._err("add_sub", "nt", 'one of "T", "C", "A", "G", or "N"')
        _abi.check(_abi.lib().jk_add_substitution(self._h, hap_ind - 1, chrom_ind - 1, nt.encode(), pos - 1))
        return self

    def add_ins(self, hap_ind, chrom_ind, pos, nts):
        self._check_pos(hap_ind, chrom_ind, pos, "add_ins")
        if not isinstance(nts, str):
            self._err("add_ins", "nts", "a single string")
        if not set(nts) <= self._NTS:
            self._err("add_ins", "nts", 'string containing only "T", "C", "A", "G", or "N"')
        _abi.check(_abi.lib().jk_add_insertion(self._h, hap_ind - 1, chrom_ind - 1, nts.encode(), pos - 1))
        return self

    def add_del(self, hap_ind, chrom_ind, pos, n_nts):
        self._check_pos(hap_ind, chrom_ind, pos, "add_del")
        if not isinstance(n_nts, (int, np.integer)) or n_nts < 1:
            self._err("add_del", "n_nts", "a single integer >= 1")
        _abi.check(_abi.lib().jk_add_deletion(self._h, hap_ind - 1, chrom_ind - 1, int(n_nts), pos - 1))
        return self

    def chrom(self, hap_ind, chrom_ind):
        """Full haplotype chromosome (1-based), like haplotypes$chrom()."""
        self._check_hap(hap_ind, "chrom")
        v = self._raw_view()
        n = int(v.chrom_size[(hap_ind - 1) * self.n_chroms() + chrom_ind - 1])
        out = np.zeros(max(n, 1), dtype=np.uint8)
        _abi.check(_abi.lib().jk_hap_chrom_full(C.byref(v), hap_ind - 1, chrom_ind - 1, out.ctypes.data, n))
        return out[:n].tobytes().decode()

    def snapshot(self):
        """Copy of the current tables as a HapSet."""
        v = self._raw_view()
        nh, nc = self._n_haps, self.n_chroms()
        cells, m = [], 0
        blob_len = int(v.nuc_off[sum(int(v.n_mut[k]) for k in range(nh * nc))])
        blob = C.string_at(v.nuc_blob, blob_len)
        for h in range(nh):
            row = []
            for c in range(nc):
                k = h * nc + c
                n = int(v.n_mut[k])
                row.append({"chrom_size": int(v.chrom_size[k]),
                            "old_pos": [int(v.old_pos[m + i]) for i in range(n)],
                            "new_pos": [int(v.new_pos[m + i]) for i in range(n)],
                            "nucleos": [blob[int(v.nuc_off[m + i]):int(v.nuc_off[m + i + 1])].decode() for i in range(n)]})
                m += n
            cells.append(row)
        names = [v.hap_names[h].decode() for h in range(nh)]
        return HapSet(self.ref, cells, names)


class FlatHapSet:
    """Haplotypes held directly in the flat layout of ``jk_hap_set`` (numpy arrays): for tables with millions of
    mutations, where per-mutation Python objects (``HapSet.cells``) are too slow.  Same read-side interface as
    ``HapSet`` for illumina()/pacbio()."""

    def __init__(self, ref, n_haps, chrom_size, n_mut, old_pos, new_pos, nuc_off, blob, names=None):
        self.ref = ref
        self._n_haps = int(n_haps)
        self.chrom_size = np.ascontiguousarray(chrom_size, dtype=np.uint64)
        self.n_mut = np.ascontiguousarray(n_mut, dtype=np.uint64)
        self.old_pos = np.ascontiguousarray(old_pos, dtype=np.uint64)
        self.new_pos = np.ascontiguousarray(new_pos, dtype=np.uint64)
        self.nuc_off = np.ascontiguousarray(nuc_off, dtype=np.uint64)
        self.blob = np.ascontiguousarray(np.concatenate([np.asarray(blob, dtype=np.uint8), np.zeros(1, dtype=np.uint8)]))
        self.names = list(names) if names is not None else ["hap%d" % i for i in range(self._n_haps)]

    def n_haps(self):
        return self._n_haps

    def hap_names(self):
        return list(self.names)

    seed_budget = HapSet.seed_budget

    def _view(self):
        rv, keep = self.ref._view()
        names = (C.c_char_p * self._n_haps)(*[x.encode() for x in self.names])
        v = _abi.HapSetView()
        v.n_haps, v.n_chroms = self._n_haps, self.ref.n_chroms()
        v.hap_names = names
        v.ref = rv
        p64 = C.POINTER(C.c_uint64)
        v.chrom_size = self.chrom_size.ctypes.data_as(p64)
        v.n_mut = self.n_mut.ctypes.data_as(p64)
        v.old_pos = self.old_pos.ctypes.data_as(p64)
        v.new_pos = self.new_pos.ctypes.data_as(p64)
        v.nuc_off = self.nuc_off.ctypes.data_as(p64)
        v.nuc_blob = self.blob.ctypes.data
        return v, [keep, names, self]


def random_haplotypes_flat(ref, n_haps, seed, sub_rate=1e-3, ins_rate=1e-4, del_rate=1e-4):
    """Vectorised synthetic tables for genome-scale runs (BASELINE configs[2..3]): substitutions and 1-base
    insertions / deletions at the given per-base rates, candidate sites at least 4 bases apart (never adjacent or
    overlapping), in the reference's canonical representation."""
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(b"TCAG", dtype=np.uint8)
    total = sub_rate + ins_rate + del_rate
    sizes, counts, ops, nps, lens, blobs = [], [], [], [], [], []
    for h in range(n_haps):
        for seq in ref.seqs:
            n = int(seq.size)
            k = int(n * total)
            pos = np.unique(rng.integers(1, max(n - 2, 2), size=k) & ~np.int64(3)) if n > 8 else np.zeros(0, dtype=np.int64)
            pos = pos[pos >= 1]
            kind = rng.choice(3, size=pos.size, p=np.array([sub_rate, ins_rate, del_rate]) / total)
            delta = np.where(kind == 1, 1, np.where(kind == 2, -1, 0)).astype(np.int64)
            shift = np.concatenate([[0], np.cumsum(delta)[:-1]]) if pos.size else np.zeros(0, dtype=np.int64)
            nlen = np.where(kind == 0, 1, np.where(kind == 1, 2, 0)).astype(np.int64)
            off = np.concatenate([[0], np.cumsum(nlen)])
            blob = np.zeros(int(off[-1]), dtype=np.uint8)
            rnd = lut[rng.integers(0, 4, size=pos.size)]
            sub = kind == 0
            ins = kind == 1
            blob[off[:-1][sub]] = rnd[sub]
            blob[off[:-1][ins]] = seq[pos[ins]]
            blob[off[:-1][ins] + 1] = rnd[ins]
            sizes.append(n + int(delta.sum())); counts.append(pos.size)
            ops.append(pos.astype(np.uint64)); nps.append((pos + shift).astype(np.uint64)); lens.append(nlen); blobs.append(blob)
    nuc_off = np.concatenate([[0], np.cumsum(np.concatenate(lens))]).astype(np.uint64) if lens else np.zeros(1, dtype=np.uint64)
    return FlatHapSet(ref, n_haps, sizes, counts, np.concatenate(ops), np.concatenate(nps), nuc_off, np.concatenate(blobs))


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
