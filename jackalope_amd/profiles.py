"""ART-style Illumina quality profiles: the Python mirror of ``read_profile`` / ``format_profile``
(/root/reference/R/hts_illumina.R:133-262) and of the built-in profile lookup (:17-115).

A profile is, per nucleotide T, C, A, G and per read position, a list of qualities and their
probabilities (successive differences of the file's cumulative counts divided by their sum).
"""
import gzip
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "art_profiles")

# name, read_length, read, file stem, abbreviation  (R/hts_illumina.R:17-47), ordered by read length
_BUILTIN = [
    ("Genome Analyzer I", 36, 1, "EmpR36R1", "GA1"), ("Genome Analyzer I", 36, 2, "EmpR36R2", "GA1"),
    ("Genome Analyzer I", 44, 1, "EmpR44R1", "GA1"), ("Genome Analyzer I", 44, 2, "EmpR44R2", "GA1"),
    ("Genome Analyzer II", 50, 1, "EmpR50R1", "GA2"), ("Genome Analyzer II", 50, 2, "EmpR50R2", "GA2"),
    ("MiniSeq TruSeq", 50, 1, "MiniSeqTruSeqL50", "MinS"),
    ("Genome Analyzer II", 75, 1, "EmpR75R1", "GA2"), ("Genome Analyzer II", 75, 2, "EmpR75R2", "GA2"),
    ("NextSeq 500 v2", 75, 1, "NextSeq500v2L75R1", "NS50"), ("NextSeq 500 v2", 75, 2, "NextSeq500v2L75R2", "NS50"),
    ("HiSeq 1000", 100, 1, "Emp100R1", "HS10"), ("HiSeq 1000", 100, 2, "Emp100R2", "HS10"),
    ("HiSeq 2000", 100, 1, "HiSeq2000L100R1", "HS20"), ("HiSeq 2000", 100, 2, "HiSeq2000L100R2", "HS20"),
    ("HiSeq 2500", 125, 1, "HiSeq2500L125R1", "HS25"), ("HiSeq 2500", 125, 2, "HiSeq2500L125R2", "HS25"),
    ("HiSeq 2500", 150, 1, "HiSeq2500L150R1filter", "HS25"), ("HiSeq 2500", 150, 2, "HiSeq2500L150R2filter", "HS25"),
    ("HiSeqX v2.5 PCR free", 150, 1, "HiSeqXPCRfreeL150R1", "HSXn"), ("HiSeqX v2.5 PCR free", 150, 2, "HiSeqXPCRfreeL150R2", "HSXn"),
    ("HiSeqX v2.5 TruSeq", 150, 1, "HiSeqXtruSeqL150R1", "HSXt"), ("HiSeqX v2.5 TruSeq", 150, 2, "HiSeqXtruSeqL150R2", "HSXt"),
    ("MiSeq v1", 250, 1, "EmpMiSeq250R1", "MSv1"), ("MiSeq v1", 250, 2, "EmpMiSeq250R2", "MSv1"),
    ("MiSeq v3", 250, 1, "MiSeqv3L250R1", "MSv3"), ("MiSeq v3", 250, 2, "MiSeqv3L250R2", "MSv3"),
]


class Profile:
    """Flattened [nt][pos][k] tables for one read end, in the layout the C ABI takes."""

    def __init__(self, n_quals, probs, quals):
        self.n_quals = np.ascontiguousarray(n_quals, dtype=np.uint32)      # [4, L]
        self.probs = np.ascontiguousarray(probs, dtype=np.float64)
        self.quals = np.ascontiguousarray(quals, dtype=np.uint8)
        self.read_length = int(self.n_quals.shape[1])

    @staticmethod
    def from_counts(n_quals, quals, cum_counts, read_length):
        n_quals = np.asarray(n_quals)
        if n_quals.shape[1] < read_length:
            raise ValueError("\nFor nucleotide T in the profile, it doesn't provide at least as many positions "
                             "as your desired read length.")
        out_n = n_quals[:, :read_length]
        probs, qs = [], []
        off = 0
        for nt in range(4):
            for pos in range(n_quals.shape[1]):
                k = int(n_quals[nt, pos])
                if pos < read_length:
                    c = np.asarray(cum_counts[off:off + k], dtype=np.float64)
                    p = c.copy()
                    if k > 1:
                        p[1:] = c[1:] - c[:-1]
                    # R's sum() accumulates in long double; the addends are integers < 2^53, so a
                    # sequential double sum is the same number
                    total = 0.0
                    for v in p:
                        total += float(v)
                    probs.append(p / total)
                    qs.append(np.asarray(quals[off:off + k]))
                off += k
        # uint8 conversion wraps like Rcpp's as<uint8>() of an out-of-range integer does in practice
        q = np.concatenate(qs).astype(np.int64) & 0xFF
        return Profile(out_n, np.concatenate(probs), q.astype(np.uint8))

    def c_struct(self):
        import ctypes as C
        from . import _abi
        return _abi.IlluminaProfile(self.read_length,
                                    self.n_quals.ctypes.data_as(C.POINTER(C.c_uint32)),
                                    self.probs.ctypes.data_as(C.POINTER(C.c_double)),
                                    self.quals.ctypes.data_as(C.POINTER(C.c_uint8)))


def seq_sys_by_read_length(read_length):
    """R/hts_illumina.R:53-72"""
    if read_length <= 44:
        return "GA1"
    if read_length <= 75:
        return "GA2"
    if read_length <= 100:
        return "HS20"
    if read_length <= 150:
        return "HS25"
    if read_length <= 250:
        return "MSv1"
    raise ValueError("\nNo built-in Illumina profile can generate reads of length %d." % read_length)


def find_profile_file(seq_sys, read_length, read):
    """R/hts_illumina.R:85-115: the built-in profile of `seq_sys` with the smallest read length
    >= `read_length`."""
    rows = [r for r in _BUILTIN if (r[0] == seq_sys or r[4] == seq_sys) and r[2] == read]
    if not rows:
        raise ValueError("\nThe desired Illumina platform name (%r) isn't among those with built-in profiles "
                         "for read %d." % (seq_sys, read))
    rows = [r for r in rows if r[1] >= read_length]
    if not rows:
        raise ValueError("\nThe desired Illumina platform (\"%s\") doesn't have a built-in profile of length %d "
                         "or longer." % (seq_sys, read_length))
    stem = min(rows, key=lambda r: r[1])[3]
    path = os.path.join(_DATA, stem + ".npz")
    if not os.path.exists(path):
        raise FileNotFoundError("built-in profile %s is not bundled with this package; pass the ART profile "
                                "file as profile%d=" % (stem, read))
    return path


def _parse_art_text(path):
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt") as fh:
        lines = [ln.rstrip("\n") for ln in fh]
    lines = [ln for ln in lines if ln[:1] in ("T", "C", "A", "G")]
    rows = {nt: {} for nt in "TCAG"}
    for i in range(0, len(lines), 2):
        a, b = lines[i].split("\t"), lines[i + 1].split("\t")
        if a and a[-1] == "":
            a.pop()
        if b and b[-1] == "":
            b.pop()
        if a[:2] != b[:2]:
            raise ValueError("\nInput profile file does not have proper format. The two lines specifying quality "
                             "and distances should always have the same values for nucleotide and position.")
        if len(a) != len(b):
            raise ValueError("\nInput profile file does not have proper format. The two lines specifying quality "
                             "and distances should always have the same number of tab-delimited columns.")
        rows[a[0]][int(a[1])] = ([int(x) for x in a[2:]], [float(x) for x in b[2:]])
    npos = len(rows["T"])
    n_quals = np.zeros((4, npos), dtype=np.int64)
    quals, cum = [], []
    for k, nt in enumerate("TCAG"):
        if sorted(rows[nt]) != list(range(len(rows[nt]))):
            raise ValueError("\nFor nucleotide %s in the profile, the positions aren't a vector from 0 to "
                             "length(positions) - 1." % nt)
        if len(rows[nt]) != npos:
            raise ValueError("profile rows differ in length between nucleotides")
        for pos in range(npos):
            q, c = rows[nt][pos]
            n_quals[k, pos] = len(q)
            quals += q
            cum += c
    return n_quals, np.asarray(quals, dtype=np.int64), np.asarray(cum, dtype=np.float64)


def read_profile(profile_fn, seq_sys, read_length, read):
    """R/hts_illumina.R:211-262."""
    if profile_fn is not None and seq_sys is not None:
        raise ValueError("\nFor Illumina sequencing, the user should never provide both a custom profile file "
                         "and a sequencing system.")
    if profile_fn is None and seq_sys is None:
        seq_sys = seq_sys_by_read_length(read_length)
    if profile_fn is None:
        profile_fn = find_profile_file(seq_sys, read_length, read)
    if profile_fn.endswith(".npz"):
        z = np.load(profile_fn)
        n_quals, quals, cum = z["n_quals"], z["quals"], z["cum_counts"]
    else:
        n_quals, quals, cum = _parse_art_text(profile_fn)
    return Profile.from_counts(n_quals, quals, cum, read_length)
