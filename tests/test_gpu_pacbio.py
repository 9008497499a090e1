"""GPU parity for pacbio() (pacbio_ref_cpp / pacbio_hap_cpp, src/hts_pacbio.cpp:579-715) against the CPU
oracle on the same seeded inputs: FASTQ bytes must be identical.  The reference's own PacBio tests are
structural only (tests/testthat/test-sequencer.R:285-319); they are restated here too."""
import numpy as np
import pytest

from helpers import builder_haplotypes, first_diff, fastq_records
from jackalope_amd.genome import random_haplotypes

pytestmark = pytest.mark.gpu


def hip(ja, obj, n_reads, T, words, pb, **extra):
    s = ja.pacbio(obj, None, n_reads, n_threads=T, seed_words=words, _session=True, **pb, **extra)
    with s:
        s.generate()
        sizes, reads = s.sizes()
        return s.fetch(0), reads, s.seed_words_used()


def check_ref(ja, O, g, n_reads, T, pb=None, seed=1):
    pb = pb or {}
    words = ja.seed_words(seed, 16 * T + 64)
    o, used_o, _ = O.pacbio_ref(g, pb, n_reads=n_reads, n_threads=T, words=words)
    h, reads, used_h = hip(ja, g, n_reads, T, words, pb)
    assert used_o == used_h
    if h != o:
        raise AssertionError("FASTQ differs at byte %d:\nHIP    %r\noracle %r" % first_diff(h, o))
    assert reads == n_reads
    return h


def test_reference_structural_checks(ja, O):
    """test-sequencer.R:285-300: 4 lines per read, '@' ids, '+' separators, equal base/quality lengths."""
    g = ja.synthetic_genome([400_000, 90_000], seed=41)
    h = check_ref(ja, O, g, 100, 1)
    recs = fastq_records(h)
    assert len(recs) == 100
    assert all(r[0].startswith(b"@REF-chrom0-") and r[2] == b"+" and len(r[1]) in (len(r[3]), len(r[3]) + 1) for r in recs)


@pytest.mark.parametrize("T", [3, 64, 500])
def test_default_model_many_lanes(ja, O, T):
    g = ja.synthetic_genome([600_000], seed=42)
    check_ref(ja, O, g, 700, T, seed=T)


def test_custom_read_lengths_and_duplicates(ja, O):
    g = ja.synthetic_genome([300_000, 10_000], seed=43)
    check_ref(ja, O, g, 600, 16, {"custom_read_lengths": [[100, 1], [2500, 2], [57, 1], [9000, 0.5]]})
    check_ref(ja, O, g, 600, 16, {"custom_read_lengths": [300, 1200, 5000], "prob_dup": 0.4, "read_pool_size": 7})
    check_ref(ja, O, g, 300, 5, {"prob_dup": 0.9, "read_pool_size": 3})


def test_other_error_models(ja, O):
    g = ja.synthetic_genome([500_000], seed=44)
    check_ref(ja, O, g, 300, 8, {"ins_prob": 0.02, "del_prob": 0.15, "sub_prob": 0.05, "prob_thresh": 0.3})
    check_ref(ja, O, g, 300, 8, {"max_passes": 5, "sqrt_params": (0.8, 0.3), "norm_params": (0.1, 0.35)})
    check_ref(ja, O, g, 300, 8, {"lognorm_read_length": (0.35, -500.0, 3000.0), "min_read_length": 400})
    check_ref(ja, O, g, 300, 8, {"norm_params": (-3.0, 0.2)})      # far-tail branch of trunc_norm
    # chi-square with n < 2 degrees of freedom (the reference clamps n at 0.001, src/hts_pacbio.h:158-160, so it does go
    # there): gamma shape n / 2 < 1, libstdc++'s pow branch
    check_ref(ja, O, g, 400, 16, {"chi2_params_n": (0.0001, 0.5, 5500)})
    check_ref(ja, O, g, 400, 16, {"chi2_params_n": (0.0, 0.0, 5500)})      # n clamped to 0.001


def test_empty_chromosomes_get_no_reads(ja, O):
    """An empty chromosome has probability 0 in reads_per_group (src/hts.h:78); PacBio lanes take all their reads from their
    first chromosome with a quota (src/hts_pacbio.cpp:145-151), which is never an empty one."""
    rng = np.random.default_rng(47)
    seqs = [rng.choice(np.frombuffer(b"TCAG", dtype=np.uint8), size=n) for n in (0, 120_000, 0, 40_000, 0)]
    check_ref(ja, O, ja.RefGenome(seqs), 400, 12, {"custom_read_lengths": [500, 2500, 7000]})


def test_non_tcag_bases_are_copied_like_the_reference(ja, O):
    rng = np.random.default_rng(45)
    seq = np.frombuffer(b"TCAGTCAGTCAGNnRY-\x00\x03", dtype=np.uint8)[rng.integers(0, 19, size=250_000)]
    g = ja.RefGenome([seq])
    check_ref(ja, O, g, 200, 6, {"custom_read_lengths": [800, 3000]})


def test_runs_of_n_take_the_word_path(ja, O):
    """A genome whose only non-TCAG bytes are N -- single ones and blocks up to thousands, as in assemblies with
    gaps: pass 2 keeps such 32-position words in registers (N copied through on both strands, a substitution on
    an N gives N), which the scattered odd bytes of the test above never reach."""
    rng = np.random.default_rng(46)
    seq = np.frombuffer(b"TCAG", dtype=np.uint8)[rng.integers(0, 4, size=400_000)].copy()
    seq[rng.integers(0, seq.size, size=seq.size // 300)] = ord("N")
    for at, n in [(10_000, 1), (20_000, 5), (30_000, 33), (50_000, 700), (100_000, 6000), (250_000, 40_000), (399_000, 1000)]:
        seq[at:at + n] = ord("N")
    g = ja.RefGenome([seq])
    check_ref(ja, O, g, 400, 70, {"custom_read_lengths": [900, 4000, 12000]})
    check_ref(ja, O, g, 300, 5, {"custom_read_lengths": [2000, 7000], "sub_prob": 0.2, "ins_prob": 0.05, "del_prob": 0.05}, seed=2)


def test_haplotypes(ja, O):
    ref = ja.synthetic_genome([250_000, 60_000], seed=46)
    hs = random_haplotypes(ref, 3, seed=47, sub_rate=0.01, ins_rate=0.004, del_rate=0.004)
    T, n = 12, 400
    words = ja.seed_words(3, hs.seed_budget(T))
    pb = {"custom_read_lengths": [500, 2000, 7000], "prob_dup": 0.1}
    o, used_o, _ = O.pacbio_hap(hs, pb, hap_probs=[1.0, 2.0, 0.5], n_reads=n, n_threads=T, words=words)
    h, reads, used_h = hip(ja, hs, n, T, words, dict(pb, haplotype_probs=[1.0, 2.0, 0.5]))
    assert used_o == used_h
    if h != o:
        raise AssertionError("FASTQ differs at byte %d:\nHIP    %r\noracle %r" % first_diff(h, o))
    o2, _, _ = O.pacbio_hap(hs, {}, hap_probs=[1.0, 1.0, 1.0], n_reads=120, n_threads=4, words=words)
    h2, _, _ = hip(ja, hs, 120, 4, words, {})
    assert h2 == o2


def test_tables_from_the_mutation_builder(ja, O):
    """Overlapping edits made by jk_add_*: merged deletions, trimmed insertions, records sharing a new_pos."""
    hs = builder_haplotypes(ja, [60_000, 9_000], 2, 4000, seed=60)
    T, n = 9, 300
    words = ja.seed_words(4, hs.seed_budget(T))
    pb = {"custom_read_lengths": [300, 1500, 4000]}
    o, used_o, _ = O.pacbio_hap(hs, pb, hap_probs=[1.0, 1.0], n_reads=n, n_threads=T, words=words)
    h, reads, used_h = hip(ja, hs, n, T, words, pb)
    assert used_o == used_h
    if h != o:
        raise AssertionError("FASTQ differs at byte %d:\nHIP    %r\noracle %r" % first_diff(h, o))


def test_lane_shards_and_files(ja, O, tmp_path):
    g = ja.synthetic_genome([400_000], seed=48)
    T, n = 40, 500
    words = ja.seed_words(6, 16 * T)
    whole, _, _ = hip(ja, g, n, T, words, {})
    parts = [hip(ja, g, n, T, words, {}, lane_begin=lo, lane_end=hi)[0] for lo, hi in [(0, 13), (13, 40)]]
    assert b"".join(parts) == whole
    small, _, _ = hip(ja, g, n, T, words, {}, max_batch_bytes=3_000_000)
    assert small == whole
    prefix = str(tmp_path / "pb")
    ja.pacbio(g, prefix, n, n_threads=T, seed_words=words)
    assert open(prefix + "_R1.fq", "rb").read() == whole


def test_sep_files_and_compressed_sink(ja, O, tmp_path):
    """write_reads_cpp_sep_files_ (src/hts.h:512-552) on the PacBio path: one reads_per_group draw splits the reads
    over haplotypes, then one run per haplotype with one-hot probabilities -> <prefix>_<hap>_R1.fq[.gz]."""
    import ctypes as C
    import gzip
    ref = ja.synthetic_genome([120_000, 30_000], seed=61)
    hs = random_haplotypes(ref, 3, seed=62, sub_rate=0.01, ins_rate=0.002, del_rate=0.002)
    T, n = 5, 240
    words = ja.seed_words(11, hs.seed_budget(T))
    pb = {"custom_read_lengths": [400, 1500, 3000]}
    prefix = str(tmp_path / "pbsep")
    ja.pacbio(hs, prefix, n, n_threads=T, seed_words=words, sep_files=True, haplotype_probs=[2, 1, 1], compress=True, **pb)
    probs = np.array([2.0, 1.0, 1.0])
    per_file, used = np.zeros(3, dtype=np.uint64), C.c_uint64()
    assert O.lib().orc_reads_per_group(C.c_uint64(n), probs.ctypes.data_as(C.c_void_p), C.c_uint64(3),
                                       words.ctypes.data_as(C.c_void_p), C.c_uint64(words.size),
                                       per_file.ctypes.data_as(C.c_void_p), C.byref(used)) == 0
    pos, total = int(used.value), 0
    for h in range(3):
        one_hot = [1.0 if k == h else 0.0 for k in range(3)]
        o, u, _ = O.pacbio_hap(hs, pb, hap_probs=one_hot, n_reads=int(per_file[h]), n_threads=T, words=words[pos:])
        pos += u
        raw = open("%s_hap%d_R1.fq.gz" % (prefix, h), "rb").read()
        assert gzip.decompress(raw) == o
        assert raw[12:14] == b"BC"                 # BGZF blocks (made on the device)
        total += len(fastq_records(o))
    assert total == n


def test_exact_index_routine_for_every_draw(ja, O, monkeypatch):
    """The emit kernel turns a draw into an index of 3 or 4 from its high word and sends a whole buffer of 64 draws through
    the exact routine (runif_index32) only when one of them lies within 16 units of a boundary -- 2^-28 per draw, which no
    small case reaches.  JK_PB_FORCE_EXACT=1 sends every buffer that way (and every block through the general path): same
    bytes.  Jobs with many draws per block (high insertion / substitution rates: blocks whose draws span both buffers),
    duplicates, haplotypes through the tables and reads that over-read their window."""
    monkeypatch.setenv("JK_PB_FORCE_EXACT", "1")
    g = ja.synthetic_genome([300_000, 10_000], seed=48)
    check_ref(ja, O, g, 500, 16, {"custom_read_lengths": [300, 1200, 5000, 20000]})
    check_ref(ja, O, g, 300, 8, {"ins_prob": 0.3, "del_prob": 0.2, "sub_prob": 0.3, "prob_dup": 0.3, "read_pool_size": 5})
    tiny = ja.synthetic_genome([2500, 900], seed=49)
    check_ref(ja, O, tiny, 400, 3, {"custom_read_lengths": [800, 2500], "prob_dup": 0.5})


def test_blocks_whose_draws_span_both_buffers(ja, O):
    """Event rates at which most 64-position blocks hold dozens of draws (the fast groups must count them and stay within
    the 128 draws at hand), next to rates at which hardly any block holds one."""
    g = ja.synthetic_genome([400_000], seed=50)
    check_ref(ja, O, g, 400, 8, {"ins_prob": 0.45, "del_prob": 0.05, "sub_prob": 0.45, "custom_read_lengths": [3000, 9000]})
    check_ref(ja, O, g, 400, 8, {"ins_prob": 0.001, "del_prob": 0.001, "sub_prob": 0.001, "custom_read_lengths": [3000, 9000]})
    check_ref(ja, O, g, 400, 8, {"ins_prob": 1e-6, "del_prob": 1e-6, "sub_prob": 1e-6, "custom_read_lengths": [6000]})
    # probabilities of exactly 0 (pow(0, y) = 0): an error-free read, and each kind of event switched off on its own
    check_ref(ja, O, g, 200, 8, {"ins_prob": 0.0, "del_prob": 0.0, "sub_prob": 0.0, "custom_read_lengths": [6000]})
    check_ref(ja, O, g, 200, 8, {"ins_prob": 0.0, "custom_read_lengths": [3000]})
    check_ref(ja, O, g, 200, 8, {"del_prob": 0.0, "sub_prob": 0.0, "custom_read_lengths": [3000]})
