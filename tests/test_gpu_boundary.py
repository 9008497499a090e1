"""GPU: the drop-in boundary run from the two languages a maintainer would use it from.
(1) tests/abi_driver.c -- plain C99: structs filled by hand, seed words through the callback, the one-shot call and the
    job API; its FASTQ files must equal the oracle's.
(2) the four Rcpp shims of jackalope_amd/rcpp/ (compiled against tests/rcpp_stubs/, which stands in for Rcpp and for
    the member names of RefGenome / HapSet) called as RcppExports would call them: files equal the oracle's, R's RNG is
    consumed word for word as the reference consumes it, and only on the calling thread."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

import boundary_build as bb
from helpers import job, run_oracle

pytestmark = pytest.mark.gpu


def read(fn):
    with open(fn, "rb") as f:
        return f.read()


def write_job_file(fn, g, prof1, prof2, words, n_reads, n_threads, j, barcode=""):
    L = prof1.read_length
    paired = j["paired"]
    with open(fn, "wb") as f:
        f.write(b"JKJOB1\0\0")
        f.write(struct.pack("<Q", g.n_chroms()))
        for name, seq in zip(g.names, g.seqs):
            f.write(struct.pack("<Q", len(name)) + name.encode())
            f.write(struct.pack("<Q", seq.size) + seq.tobytes())
        shape, scale = (j["frag_mean"] / j["frag_sd"]) ** 2, j["frag_sd"] ** 2 / j["frag_mean"]
        f.write(struct.pack("<IIQdQQddQQ", int(paired), int(j["matepair"]), n_reads, j["prob_dup"], n_threads, j["read_pool_size"],
                            shape, scale, L, 2 ** 32 - 1))
        f.write(struct.pack("<I", L))
        for p, ins, dele in ((prof1, j["ins_prob1"], j["del_prob1"]), (prof2, j["ins_prob2"], j["del_prob2"]))[:2 if paired else 1]:
            f.write(np.ascontiguousarray(p.n_quals, dtype=np.uint32).tobytes())
            f.write(struct.pack("<Q", p.probs.size))
            f.write(np.ascontiguousarray(p.probs, dtype=np.float64).tobytes())
            f.write(np.ascontiguousarray(p.quals, dtype=np.uint8).tobytes())
            f.write(struct.pack("<dd", ins, dele))
        f.write(struct.pack("<Q", len(barcode)) + barcode.encode())
        w = np.ascontiguousarray(words, dtype=np.uint32)
        f.write(struct.pack("<Q", w.size) + w.tobytes())


@pytest.mark.parametrize("mode", ["oneshot", "job"])
def test_c99_driver_writes_the_oracles_files(ja, O, hs25, tmp_path, mode):
    exe = bb.abi_driver()
    g = ja.synthetic_genome([60_000, 25_000, 9_000], seed=41)
    n_reads, T = 30_000, 257
    words = ja.seed_words(5150, 16 * T)
    j = job()
    jf, pre = str(tmp_path / "job.bin"), str(tmp_path / "c")
    write_job_file(jf, g, hs25[0], hs25[1], words, n_reads, T, j, barcode="ACGT")
    r = subprocess.run([exe, jf, pre, mode], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = dict(line.split(" ", 1) for line in r.stdout.strip().splitlines())
    o1, o2, used = run_oracle(O, g, hs25[0], hs25[1], words, n_reads, T, job(barcode="ACGT"))
    assert int(out["seed_words"]) == used
    if mode == "job":
        assert out["progress"] == "%d %d" % (n_reads, n_reads)
    assert read(pre + "_R1.fq") == o1 and read(pre + "_R2.fq") == o2


# ---- the Rcpp shims through tests/rcpp_stubs/shim_driver.cpp --------------------------------------------------

class FlatProf(C.Structure):
    _fields_ = [("L", C.c_uint32), ("n_quals", C.c_void_p), ("probs", C.c_void_p), ("quals", C.c_void_p)]


class GenomeIn(C.Structure):
    _fields_ = [("n_chroms", C.c_uint64), ("names", C.POINTER(C.c_char_p)), ("seqs", C.POINTER(C.c_void_p)), ("lens", C.POINTER(C.c_uint64))]


class HapIn(C.Structure):
    _fields_ = [("n_haps", C.c_uint64), ("hap_names", C.POINTER(C.c_char_p)), ("chrom_size", C.c_void_p), ("n_mut", C.c_void_p),
                ("old_pos", C.c_void_p), ("new_pos", C.c_void_p), ("nuc_off", C.c_void_p), ("nuc_blob", C.c_void_p)]


class DrvIllumina(C.Structure):
    _fields_ = [("paired", C.c_int32), ("matepair", C.c_int32), ("out_prefix", C.c_char_p), ("sep_files", C.c_int32),
                ("compress", C.c_int32), ("comp_method", C.c_char_p), ("n_reads", C.c_uint64), ("prob_dup", C.c_double),
                ("n_threads", C.c_uint64), ("read_pool_size", C.c_uint64), ("haplotype_probs", C.c_void_p),
                ("shape", C.c_double), ("scale", C.c_double), ("fmin", C.c_uint64), ("fmax", C.c_uint64),
                ("p1", FlatProf), ("ins1", C.c_double), ("del1", C.c_double), ("p2", FlatProf), ("ins2", C.c_double), ("del2", C.c_double),
                ("barcodes", C.POINTER(C.c_char_p)), ("n_barcodes", C.c_uint64), ("words", C.c_void_p), ("n_words", C.c_uint64),
                ("abort_after", C.c_uint64)]


class DrvPacbio(C.Structure):
    _fields_ = [("out_prefix", C.c_char_p), ("sep_files", C.c_int32), ("compress", C.c_int32), ("comp_method", C.c_char_p),
                ("n_reads", C.c_uint64), ("n_threads", C.c_uint64), ("read_pool_size", C.c_uint64), ("haplotype_probs", C.c_void_p),
                ("prob_dup", C.c_double), ("scale", C.c_double), ("sigma", C.c_double), ("loc", C.c_double), ("min_read_len", C.c_double),
                ("read_probs", C.c_void_p), ("read_lens", C.c_void_p), ("n_read_lens", C.c_uint64), ("max_passes", C.c_uint64),
                ("chi2_n", C.c_void_p), ("chi2_s", C.c_void_p), ("sqrt_p", C.c_void_p), ("norm_p", C.c_void_p),
                ("prob_thresh", C.c_double), ("prob_ins", C.c_double), ("prob_del", C.c_double), ("prob_subst", C.c_double),
                ("words", C.c_void_p), ("n_words", C.c_uint64)]


@pytest.fixture(scope="module")
def drv(built):
    os.environ["JACKALOPE_HIP_DEVICES"] = "0"
    L = C.CDLL(bb.shim_driver())
    L.drv_last_error.restype = C.c_char_p
    return L


def genome_in(g):
    n = g.n_chroms()
    names = (C.c_char_p * n)(*[x.encode() for x in g.names])
    seqs = (C.c_void_p * n)(*[s.ctypes.data for s in g.seqs])
    lens = (C.c_uint64 * n)(*g.sizes())
    v = GenomeIn(n, names, seqs, lens)
    return v, [names, seqs, lens]


def hap_in(O, hs):
    v, keep = O._hap_view(hs)       # the same flat arrays the oracle is given
    names = (C.c_char_p * hs.n_haps())(*[x.encode() for x in hs.names])
    h = HapIn(hs.n_haps(), names, v.chrom_size, v.n_mut, v.old_pos, v.new_pos, v.nuc_off, v.nuc_blob)
    return h, [keep, names, v]


def flat(p):
    return FlatProf(p.read_length, p.n_quals.ctypes.data, p.probs.ctypes.data, p.quals.ctypes.data)


def ill_args(prefix, hs25, words, n_reads, T, j, hap_probs=None, sep_files=False, barcodes=("",), abort_after=0):
    a = DrvIllumina()
    keep = []
    a.paired, a.matepair, a.out_prefix = int(j["paired"]), int(j["matepair"]), prefix.encode()
    a.sep_files, a.compress, a.comp_method = int(sep_files), 0, b"bgzip"
    a.n_reads, a.prob_dup, a.n_threads, a.read_pool_size = n_reads, j["prob_dup"], T, j["read_pool_size"]
    if hap_probs is not None:
        hp = np.ascontiguousarray(hap_probs, dtype=np.float64)
        keep.append(hp)
        a.haplotype_probs = hp.ctypes.data
    a.shape, a.scale = (j["frag_mean"] / j["frag_sd"]) ** 2, j["frag_sd"] ** 2 / j["frag_mean"]
    a.fmin, a.fmax = 150, 2 ** 32 - 1
    a.p1, a.ins1, a.del1 = flat(hs25[0]), j["ins_prob1"], j["del_prob1"]
    a.p2, a.ins2, a.del2 = flat(hs25[1]), j["ins_prob2"], j["del_prob2"]
    bcs = (C.c_char_p * len(barcodes))(*[b.encode() for b in barcodes])
    keep.append(bcs)
    a.barcodes, a.n_barcodes = bcs, len(barcodes)
    w = np.ascontiguousarray(words, dtype=np.uint32)
    keep.append(w)
    a.words, a.n_words, a.abort_after = w.ctypes.data, w.size, abort_after
    return a, keep


def stats(drv):
    s = (C.c_uint64 * 5)()
    drv.drv_stats(s)
    return dict(seeds=int(s[0]), runif_off_main=int(s[1]), progress_off_main=int(s[2]), shown=int(s[3]), max=int(s[4]))


def test_shim_illumina_ref(ja, O, hs25, drv, tmp_path):
    g = ja.synthetic_genome([80_000, 30_000], seed=42)
    n_reads, T = 40_000, 300
    words = ja.seed_words(5151, 16 * T + 8)
    gi, keep = genome_in(g)
    pre = str(tmp_path / "r")
    a, keep2 = ill_args(pre, hs25, words, n_reads, T, job(), barcodes=("",))
    assert drv.drv_illumina_ref(C.byref(gi), C.byref(a)) == 0, drv.drv_last_error()
    o1, o2, used = run_oracle(O, g, hs25[0], hs25[1], words, n_reads, T, job())
    st = stats(drv)
    assert st["seeds"] == used and st["runif_off_main"] == 0 and st["progress_off_main"] == 0
    assert st["shown"] == st["max"] == n_reads
    assert read(pre + "_R1.fq") == o1 and read(pre + "_R2.fq") == o2
    # 300 lanes reproduce the reference's 300-thread files but are far below what the GPU needs: the shim says so
    # (Rcpp::warning) instead of running slowly in silence; with JACKALOPE_HIP_LANES set it has nothing to say
    drv.drv_warnings.restype = C.c_uint64
    w0 = drv.drv_warnings()
    assert w0 >= 1
    os.environ["JACKALOPE_HIP_LANES"] = str(T)
    try:
        assert drv.drv_illumina_ref(C.byref(gi), C.byref(a)) == 0, drv.drv_last_error()
    finally:
        del os.environ["JACKALOPE_HIP_LANES"]
    assert drv.drv_warnings() == w0
    assert read(pre + "_R1.fq") == o1


def hap_oracle(O, hs, hs25, words, n_reads, T, probs, barcodes=()):
    j = job()
    return O.illumina_hap(hs, hap_probs=probs, paired=True, n_reads=n_reads, prob_dup=0.02, n_threads=T, read_pool_size=1000,
                          shape=16.0, scale=25.0, fmin=150, fmax=2 ** 32 - 1, prof1=hs25[0], prof2=hs25[1],
                          ins1=j["ins_prob1"], del1=j["del_prob1"], ins2=j["ins_prob2"], del2=j["del_prob2"], barcodes=barcodes, words=words)


def test_shim_illumina_hap_and_sep_files(ja, O, hs25, drv, tmp_path):
    from jackalope_amd.genome import random_haplotypes
    ref = ja.synthetic_genome([70_000, 40_000], seed=43)
    hs = random_haplotypes(ref, 3, seed=9)
    n_reads, T = 30_000, 128
    words = ja.seed_words(5152, hs.seed_budget(T) + 64)
    gi, keep = genome_in(ref)
    hi, keep2 = hap_in(O, hs)
    probs = [2.0, 1.0, 1.0]
    pre = str(tmp_path / "h")
    a, keep3 = ill_args(pre, hs25, words, n_reads, T, job(), hap_probs=probs, barcodes=("AC", "", "GGT"))
    assert drv.drv_illumina_hap(C.byref(gi), C.byref(hi), C.byref(a)) == 0, drv.drv_last_error()
    o1, o2, used = hap_oracle(O, hs, hs25, words, n_reads, T, probs, barcodes=("AC", "", "GGT"))
    st = stats(drv)
    assert st["seeds"] == used and st["runif_off_main"] == 0
    assert read(pre + "_R1.fq") == o1 and read(pre + "_R2.fq") == o2
    # sep_files (write_reads_cpp_sep_files_, src/hts.h:512-552): one reads_per_group draw, then one run per haplotype
    pre = str(tmp_path / "s")
    a, keep4 = ill_args(pre, hs25, words, n_reads, T, job(), hap_probs=probs, sep_files=True, barcodes=("", "", ""))
    assert drv.drv_illumina_hap(C.byref(gi), C.byref(hi), C.byref(a)) == 0, drv.drv_last_error()
    w = np.ascontiguousarray(words, dtype=np.uint32)
    per = np.zeros(3, dtype=np.uint64)
    used0 = C.c_uint64()
    p = np.ascontiguousarray(probs, dtype=np.float64)
    assert O.lib().orc_reads_per_group(C.c_uint64(n_reads // 2), p.ctypes.data_as(C.c_void_p), C.c_uint64(3), w.ctypes.data_as(C.c_void_p),
                                       C.c_uint64(w.size), per.ctypes.data_as(C.c_void_p), C.byref(used0)) == 0
    at = int(used0.value)
    for h in range(3):
        one_hot = [1.0 if k == h else 0.0 for k in range(3)]
        o1, o2, used = hap_oracle(O, hs, hs25, w[at:], 2 * int(per[h]), T, one_hot)
        at += used
        assert read("%s_%s_R1.fq" % (pre, hs.names[h])) == o1 and read("%s_%s_R2.fq" % (pre, hs.names[h])) == o2
    assert stats(drv)["seeds"] == at


def pb_args(prefix, words, n_reads, T, hap_probs=None, sep_files=False):
    d = dict(O_DEFAULTS)
    a = DrvPacbio()
    keep = []
    a.out_prefix, a.sep_files, a.compress, a.comp_method = prefix.encode(), int(sep_files), 0, b"bgzip"
    a.n_reads, a.n_threads, a.read_pool_size, a.prob_dup = n_reads, T, d["read_pool_size"], d["prob_dup"]
    if hap_probs is not None:
        hp = np.ascontiguousarray(hap_probs, dtype=np.float64)
        keep.append(hp)
        a.haplotype_probs = hp.ctypes.data
    a.sigma, a.loc, a.scale = d["lognorm_read_length"]
    a.min_read_len = d["min_read_length"]
    a.n_read_lens, a.max_passes = 0, d["max_passes"]
    for name, key in (("chi2_n", "chi2_params_n"), ("chi2_s", "chi2_params_s"), ("sqrt_p", "sqrt_params"), ("norm_p", "norm_params")):
        arr = np.ascontiguousarray(d[key], dtype=np.float64)
        keep.append(arr)
        setattr(a, name, arr.ctypes.data)
    a.prob_thresh, a.prob_ins, a.prob_del, a.prob_subst = d["prob_thresh"], d["ins_prob"], d["del_prob"], d["sub_prob"]
    w = np.ascontiguousarray(words, dtype=np.uint32)
    keep.append(w)
    a.words, a.n_words = w.ctypes.data, w.size
    return a, keep


O_DEFAULTS = None


def test_shim_pacbio_ref_and_hap(ja, O, drv, tmp_path):
    global O_DEFAULTS
    O_DEFAULTS = O.PACBIO_DEFAULTS
    from jackalope_amd.genome import random_haplotypes
    ref = ja.synthetic_genome([400_000, 150_000], seed=44)
    n_reads, T = 1500, 200
    words = ja.seed_words(5153, 16 * T + 8)
    gi, keep = genome_in(ref)
    pre = str(tmp_path / "p")
    a, keep2 = pb_args(pre, words, n_reads, T)
    assert drv.drv_pacbio_ref(C.byref(gi), C.byref(a)) == 0, drv.drv_last_error()
    o, used, _ = O.pacbio_ref(ref, {}, n_reads=n_reads, n_threads=T, words=words)
    st = stats(drv)
    assert st["seeds"] == used and st["runif_off_main"] == 0 and st["shown"] == n_reads
    assert read(pre + "_R1.fq") == o
    hs = random_haplotypes(ref, 2, seed=10)
    words = ja.seed_words(5154, hs.seed_budget(T) + 64)
    hi, keep3 = hap_in(O, hs)
    pre = str(tmp_path / "q")
    a, keep4 = pb_args(pre, words, n_reads, T, hap_probs=[1.0, 3.0])
    assert drv.drv_pacbio_hap(C.byref(gi), C.byref(hi), C.byref(a)) == 0, drv.drv_last_error()
    o, used, _ = O.pacbio_hap(hs, {}, hap_probs=[1.0, 3.0], n_reads=n_reads, n_threads=T, words=words)
    assert stats(drv)["seeds"] == used
    assert read(pre + "_R1.fq") == o


def test_shim_user_interrupt_stops_the_run(ja, hs25, drv, tmp_path):
    """Progress::check_abort() turning true on R's main thread (a user interrupt) ends the run early and quietly, as the
    reference's loops do (src/hts.h:396-399)."""
    g = ja.synthetic_genome([1_000_000], seed=45)
    n_reads, T = 4_000_000, 1 << 15
    words = ja.seed_words(5155, 16 * T + 8)
    gi, keep = genome_in(g)
    os.environ["JK_BATCH_LANES"] = "2048"         # many small launches, so there is something to interrupt
    try:
        a, keep2 = ill_args(str(tmp_path / "i"), hs25, words, n_reads, T, job(), abort_after=1)
        assert drv.drv_illumina_ref(C.byref(gi), C.byref(a)) == 0, drv.drv_last_error()
    finally:
        del os.environ["JK_BATCH_LANES"]
    st = stats(drv)
    assert 0 < st["shown"] < n_reads and st["progress_off_main"] == 0
