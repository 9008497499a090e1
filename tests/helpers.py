"""Shared helpers for the parity tests: one place that runs the same job through the oracle and
through the HIP path (via the C ABI), so the individual tests read like the reference's own."""
import numpy as np

DEFAULTS = dict(paired=True, matepair=False, prob_dup=0.02, read_pool_size=1000, frag_mean=400.0, frag_sd=100.0,
                frag_len_min=None, frag_len_max=None, ins_prob1=0.00009, del_prob1=0.00011, ins_prob2=0.00015,
                del_prob2=0.00023, barcode="")


def job(**kw):
    d = dict(DEFAULTS)
    d.update(kw)
    return d


def run_oracle(O, genome, prof1, prof2, words, n_reads, n_threads, j, **extra):
    L = prof1.read_length
    fmin = j["frag_len_min"] if j["frag_len_min"] is not None else L
    fmax = j["frag_len_max"] if j["frag_len_max"] is not None else 2 ** 32 - 1
    shape = (j["frag_mean"] / j["frag_sd"]) ** 2
    scale = j["frag_sd"] ** 2 / j["frag_mean"]
    paired = j["paired"] or j["matepair"]
    return O.illumina_ref(genome, paired=paired, matepair=j["matepair"], n_reads=n_reads, prob_dup=j["prob_dup"],
                          n_threads=n_threads, read_pool_size=j["read_pool_size"], shape=shape, scale=scale,
                          fmin=fmin, fmax=fmax, prof1=prof1, prof2=prof2, ins1=j["ins_prob1"], del1=j["del_prob1"],
                          ins2=j["ins_prob2"], del2=j["del_prob2"], barcode=j["barcode"], words=words, **extra)


def open_hip(ja, genome, profile_files, read_length, words, n_reads, n_threads, j, **extra):
    """illumina(..., _session=True) through the package's R-level mirror."""
    p1, p2 = profile_files
    return ja.illumina(genome, None, n_reads, read_length, j["paired"], frag_mean=j["frag_mean"], frag_sd=j["frag_sd"],
                       matepair=j["matepair"], profile1=p1, profile2=p2, ins_prob1=j["ins_prob1"],
                       del_prob1=j["del_prob1"], ins_prob2=j["ins_prob2"], del_prob2=j["del_prob2"],
                       frag_len_min=j["frag_len_min"], frag_len_max=j["frag_len_max"],
                       barcodes=j["barcode"] if j["barcode"] else None, prob_dup=j["prob_dup"], n_threads=n_threads,
                       read_pool_size=j["read_pool_size"], seed_words=words, _session=True, **extra)


def run_hip(ja, genome, profile_files, read_length, words, n_reads, n_threads, j, **extra):
    with open_hip(ja, genome, profile_files, read_length, words, n_reads, n_threads, j, **extra) as s:
        s.generate()
        sizes, reads = s.sizes()
        r1 = s.fetch(0)
        r2 = s.fetch(1) if len(sizes) > 1 else None
        return r1, r2, reads, s.seed_words_used()


def fastq_records(data):
    lines = data.split(b"\n")
    assert lines[-1] == b""
    lines = lines[:-1]
    assert len(lines) % 4 == 0
    return [(lines[i], lines[i + 1], lines[i + 2], lines[i + 3]) for i in range(0, len(lines), 4)]


def write_test_profile(path, n_pos=100, qual=255, count=1000):
    """The one-quality profile of the reference's known-answer tests (test-sequencer.R:82-87)."""
    with open(path, "w") as fh:
        for nt in "ACGT":          # the R code orders rows by nucleotide name
            for pos in range(n_pos):
                fh.write("%s\t%d\t%d\n" % (nt, pos, qual))
                fh.write("%s\t%d\t%d\n" % (nt, pos, count))
    return path


def first_diff(a, b):
    n = min(len(a), len(b))
    x = np.frombuffer(a[:n], dtype=np.uint8) != np.frombuffer(b[:n], dtype=np.uint8)
    idx = int(np.argmax(x)) if x.any() else n
    return idx, a[max(0, idx - 80):idx + 80], b[max(0, idx - 80):idx + 80]


def builder_haplotypes(ja, sizes, n_haps, n_edits, seed, max_indel=10):
    """Tables made by the native builder from random overlapping edits (test-R_classes.R:199-238):
    unlike random_haplotypes() they contain merged deletions, trimmed insertions and records that
    share a new_pos with the deletion before them."""
    from jackalope_amd.genome import HapBuilder
    ref = ja.synthetic_genome(sizes, seed=seed)
    b = HapBuilder(ref, n_haps)
    rng = np.random.default_rng(seed + 1)
    for h in range(1, n_haps + 1):
        for c in range(1, len(sizes) + 1):
            for _ in range(n_edits):
                size = b.sizes(h)[c - 1]
                pos = int(rng.random() * size) + 1
                r = rng.random()
                if r < 0.5:
                    b.add_sub(h, c, pos, "TCAG"[int(rng.integers(0, 4))])
                elif r < 0.75:
                    k = min(int(rng.exponential(2.0) + 1.0), max_indel)
                    b.add_ins(h, c, pos, "".join("TCAG"[int(i)] for i in rng.integers(0, 4, size=k)))
                else:
                    b.add_del(h, c, pos, min(int(rng.exponential(2.0) + 1.0), max_indel))
    return b.snapshot()


def write_fasta(fn, names, chroms, text_width=80, newline=b"\n"):
    """What write_ref_fasta__ writes (/root/reference/src/io_fasta.cpp:431-480): '>' + name, then lines of text_width."""
    with open(fn, "wb") as f:
        for name, c in zip(names, chroms):
            f.write(b">" + name.encode() + newline)
            for i in range(0, len(c), text_width):
                f.write(bytes(c[i:i + text_width]) + newline)


def write_fai(fn, names, chroms, text_width=80, newline_len=1):
    """The index of a file made by write_fasta, as tests/testthat/test-fasta_IO.R:218-235 builds one:
    name, length, byte offset of the first base, bases per line, bytes per line."""
    at = 0
    with open(fn, "w") as f:
        for name, c in zip(names, chroms):
            at += len(name) + 1 + newline_len
            f.write("%s\t%d\t%d\t%d\t%d\n" % (name, len(c), at, text_width, text_width + newline_len))
            n_lines = (len(c) + text_width - 1) // text_width
            at += len(c) + n_lines * newline_len
    return fn


class DeviceImage:
    """A session's FASTQ image of one read end as a torch uint8 tensor over the session's own device memory (no copy):
    whole-image properties of 100-200 GB images are computed where they lie."""

    def __init__(self, session, end):
        import torch
        sizes, _ = session.sizes()
        self.n = int(sizes[end])
        self.__cuda_array_interface__ = {"shape": (self.n,), "typestr": "|u1", "data": (session.device_ptr(end), False), "version": 2}
        self.t = torch.as_tensor(self, device="cuda") if self.n else torch.zeros(0, dtype=torch.uint8, device="cuda")

    def count(self, byte, lo=0, hi=None, chunk=1 << 30):
        hi = self.n if hi is None else hi
        total = 0
        for a in range(lo, hi, chunk):
            total += int((self.t[a:min(a + chunk, hi)] == byte).sum().item())
        return total

    def byte_sum(self, lo=0, hi=None, chunk=1 << 30):
        import torch
        hi = self.n if hi is None else hi
        total = 0
        for a in range(lo, hi, chunk):
            total += int(self.t[a:min(a + chunk, hi)].sum(dtype=torch.int64).item())
        return total

    def weighted_sum(self, lo, hi, chunk=1 << 28):
        """sum over the range of byte * (1 + (position in range) mod 65521): moves when bytes are permuted."""
        import torch
        total = 0
        for a in range(lo, hi, chunk):
            b = min(a + chunk, hi)
            w = (torch.arange(a - lo, b - lo, device="cuda", dtype=torch.int64) % 65521) + 1
            total += int((self.t[a:b].to(torch.int64) * w).sum().item())
        return total

    def at(self, offsets):
        import torch
        idx = torch.as_tensor(np.asarray(offsets, dtype=np.int64), device="cuda")
        return self.t[idx].cpu().numpy()
